"""Diagnostic: per-phase cycles of the persistent ProjectionHyperLSTMLayer launch (needs the stamps build:
python -m asr_amd.build --stamps; ASR_AMD_LIB=asr_amd/libasr_amd_stamps.so python tools/prj_stamps.py).
Never used for reported numbers."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asr_amd import gating_hyper_nets as G, workloads, _lib

net = G.deep_projection_hyper_model(None, None, 123, 3436, 1, 256, 512, grad_clipping=1.0, device='cuda:0', seed=1)[0]
b = workloads.make_batch(3, 16)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
x, m, y = t(b['x']), t(b['mask']), t(b['y'])
T = x.shape[1]
L = _lib.lib()
need = L.asr_prjhyper_ws_bytes(16, 512, 256, 2)
key = [k for k in [('prjhyper', i) for i in range(8)]]
phases = {False: ['Wf product', 'inner cell', 'W3 product', 'outer cell'],
          True: ['outer cell', 'W3^T product', 'inner cell', 'Wf^T product']}
# four stamps per phase: [deferred stores of the phase before + this phase's independent loads issued | the wait for the
# exchange before the phase | dependent loads, arithmetic, handed-over stores issued | publish: drain, barrier, arrival]
parts = ['pre', 'wait', 'work', 'publish']
names = {k: ['%s/%s' % (p_, q_) for p_ in v for q_ in parts] for k, v in phases.items()}
for bw in (False, True):
    net.loss(x, m, y, m, None, train=True)
    if bw:
        net.backward(); net.join_deferred()
    net.check_status()
    ws = [v for k, v in net._ws.items() if k[0] == 'prjhyper'][0]
    raw = ws[need - 65536:need].view(torch.int64).cpu().numpy()[:256 * 16].reshape(256, 16) / T
    tot = raw.sum(1)
    print('%s: T=%d, shader cycles per step (s_memtime): median over workgroups %.0f = %.2f us at 2.1 GHz' %
          ('backward' if bw else 'forward', T, np.median(tot), np.median(tot) / 2100))
    for i, n in enumerate(names[bw]):
        c = raw[:, i]
        print('   %-14s all WGs: med %6.0f min %6.0f max %6.0f   WGs 0-31 (cells): med %6.0f' %
              (n, np.median(c), c.min(), c.max(), np.median(c[:32])))
