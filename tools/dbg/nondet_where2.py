"""Which elements of the backward scan's outputs (dA, dG, dOut) differ between repetitions of the same step?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from asr_amd import gating_hyper_nets as G, workloads
dev = 'cuda:0'
net = G.deep_projection_ln_lstm_model(None, None, 123, 3436, 4, None, 512, 1.0, 0.0, device=dev, seed=1)
b = workloads.make_batch(2, 16)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
x, m, y = t(b['x']), t(b['mask']), t(b['y'])
T, B = x.shape[1], 16
lens = b['mask'].sum(1).astype(int)
print('T', T, 'lengths', lens.tolist())
ref = None
for r in range(12):
    net.loss(x, m, y, m, None); net.backward(); torch.cuda.synchronize()
    cur = {}
    for l, p in enumerate(net.pairs):
        for d in range(2):
            c = p.ctx['d'][d]
            for k in ('dA', 'dG', 'dOut', 'dR', 'dXp', 'ahat', 'gates', 'cst', 'chat', 'rs_a', 'rs_c', 'xhat_in', 'R'):
                if k in c and c[k] is not None:
                    cur[(l, d, k)] = c[k].clone()
    if ref is None:
        ref = cur; continue
    for key in sorted(cur):
        a, c0 = cur[key], ref[key]
        if not torch.equal(a, c0):
            idx = (a != c0).nonzero()
            rows = idx[:, 0].unique().tolist(); cols = idx[:, 1].unique().tolist()
            print('rep %d %s: %d elements differ; rows %s (t, b = %s) cols %s (unit, gate = %s); values %s vs %s' % (
                r, key, idx.shape[0], rows[:6], [(q // B, q % B) for q in rows[:6]], cols[:8], [(q // 4, q % 4) for q in cols[:8]],
                a[idx[0, 0], idx[0, 1]].item(), c0[idx[0, 0], idx[0, 1]].item()))
