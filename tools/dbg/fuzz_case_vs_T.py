"""One fuzz case (tools/fuzz_parity.py numbering) re-run with its time axis cut to several lengths and with several parameter
perturbations: error of one named gradient tensor, HIP and float32 oracle, against the float64 oracle.
  python tools/dbg/fuzz_case_vs_T.py <case> <tensor name> [occurrence]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import fuzz_parity as F
from common import make_batch
from test_gpu_parity import perturb_params
from oracle import asr_oracle as O

case = int(sys.argv[1]); name = sys.argv[2]; occ = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rng = np.random.default_rng(0)
for k in range(case + 1):
    p, mk = F.make_case(rng)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to('cuda:0')
for pseed in (100 + case, 7, 8):
    for T in (4, 8, 16, 32, p['T']):
        net = mk()
        perturb_params(net, pseed, 0.15 if p['N'] <= 36 else 0.03)
        lens = [min(l, T) for l in p['lengths']]
        b = make_batch(np.random.default_rng(200 + case), p['B'], T, p['D'], p['C'], lengths=lens, I=p['I'])
        c = b.get('ivec')
        vals = net.get_all_param_values()
        g64 = O.OracleNetwork(net.describe(), vals).loss_and_grads(b['x'], b['mask'], b['y'], b['mask'], c)[3]
        g32 = O.OracleNetwork(net.describe(), vals, dtype=np.float32).loss_and_grads(b['x'], b['mask'], b['y'], b['mask'], c)[3]
        net.loss(t(b['x']), t(b['mask']), t(b['y']), t(b['mask']), t(c) if c is not None else None, train=True)
        net.backward()
        gh = net.get_all_grad_values()
        idx = [i for i, (n, _) in enumerate(net.param_names_shapes()) if n == name][occ]
        gmax = max(np.abs(g).max() for g in g64)
        den = np.abs(g64[idx]).max() + 1e-3 * gmax
        allh = max(np.abs(h - r).max() / (np.abs(r).max() + 1e-3 * gmax) for h, r in zip(gh, g64))
        all32 = max(np.abs(h - r).max() / (np.abs(r).max() + 1e-3 * gmax) for h, r in zip(g32, g64))
        print('perturb seed %3d T %3d: %s hip %.2e oracle32 %.2e | worst tensor hip %.2e oracle32 %.2e | hip-o64 %s' % (
            pseed, T, name, np.abs(gh[idx] - g64[idx]).max() / den, np.abs(g32[idx] - g64[idx]).max() / den, allh, all32,
            np.array2string((gh[idx] - g64[idx]).ravel()[:4], precision=2)), flush=True)
