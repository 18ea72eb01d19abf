"""Step time of deep_scaling_hyper_model at the shape of scripts/timit/exp_scalehyper.py (one bidirectional layer,
64 inner / 128 outer units, layer norm + learned initial rows, batch 32): the option path (persistent launch, or the step
loop with ASR_SCAL_PERSIST=0) beside the option-free layer on the persistent HyperLSTM scan (diagnostic)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from asr_amd.gating_hyper_nets import deep_scaling_hyper_model
for opts in (dict(use_layer_norm=True, learn_init=True), dict()):
    net = deep_scaling_hyper_model(None, None, 123, 63, [64], [128], grad_clipping=1.0, device='cuda:0', seed=1, **opts)
    rng = np.random.default_rng(0)
    B, T = 32, 400
    lens = rng.integers(150, T + 1, B); lens[0] = T
    x = torch.from_numpy(rng.standard_normal((B, T, 123)).astype(np.float32)).cuda()
    m = torch.from_numpy((np.arange(T)[None, :] < lens[:, None]).astype(np.float32)).cuda()
    y = torch.from_numpy(rng.integers(0, 63, (B, T)).astype(np.int32)).cuda()
    for _ in range(2):
        net.loss(x, m, y, m); net.backward()
    torch.cuda.synchronize(); net.check_status()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        net.loss(x, m, y, m); net.backward()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print('%-40s B=%d T=%d: %.1f ms per fwd+bwd, %.0f frames/s, %.1f us per time step and pass' %
          (opts or 'no options (persistent scan)', B, T, dt * 1e3, lens.sum() / dt, dt * 1e6 / (2 * T)), flush=True)
