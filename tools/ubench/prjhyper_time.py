"""Time one training step (fwd + BPTT) of deep_projection_hyper_model at the reference script's defaults
(3 layers, 512 units, 256 factors, B=16, WSJ-shaped lengths) on cuda:0.  Prints frames/s."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from asr_amd import gating_hyper_nets as G, workloads

dev = 'cuda:0'
net = G.deep_projection_hyper_model(None, None, 123, 3436, 3, 256, 512, grad_clipping=1.0, device=dev, seed=1)[0]
b = workloads.make_batch(3, 16)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
x, m, y = t(b['x']), t(b['mask']), t(b['y'])
frames = float(b['mask'].sum())
for it in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    net.loss(x, m, y, m, None, train=True); net.backward(); net.join_deferred()
    torch.cuda.synchronize(); dt = time.time() - t0
    print('step %d: %.1f ms  T=%d  %.0f frames/s' % (it, dt * 1e3, x.shape[1], frames / dt), flush=True)
