"""For a dA element that differs between repetitions: recompute it from the saved tensors (dG, alpha_hid, ahat, rs_a) and
find which input the deviating value corresponds to."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from asr_amd import gating_hyper_nets as G, workloads
dev = 'cuda:0'
net = G.deep_projection_ln_lstm_model(None, None, 123, 3436, 4, None, 512, 1.0, 0.0, device=dev, seed=1)
b = workloads.make_batch(2, 16)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
x, m, y = t(b['x']), t(b['mask']), t(b['y'])
T, B, N = x.shape[1], 16, 512
runs = []
for r in range(6):
    net.loss(x, m, y, m, None); net.backward(); torch.cuda.synchronize()
    cur = {}
    for l, p in enumerate(net.pairs):
        for d in range(2):
            c = p.ctx['d'][d]
            cur[(l, d)] = dict(dA=c['dA'].double().cpu().numpy(), dG=c['dG'].double().cpu().numpy(), ahat=c['ahat'].double().cpu().numpy(),
                               rs=c['rs_a'].double().cpu().numpy(), al=p.dirs[d].p('a_hid').double().cpu().numpy())
    runs.append(cur)
shown = 0
for key in runs[0]:
    A = np.stack([r[key]['dA'] for r in runs])
    dev_ = (A != np.median(A, axis=0)[None]).any(axis=0)
    idx = np.argwhere(dev_)
    for row, col in idx[:40:4]:
        vals = A[:, row, col]
        good = np.median(vals); bad = [v for v in vals if v != good][0]
        c0 = runs[0][key]
        dah = c0['dG'][row] * c0['al']; ah = c0['ahat'][row]; rs = c0['rs'][row, 0]
        m1 = dah.sum() / (4 * N); m2 = (dah * ah).sum() / (4 * N)
        want = rs * (dah[col] - m1 - ah[col] * m2)
        # which ahat value would give the bad result?
        ah_bad = (dah[col] - m1 - bad / rs) / m2 if m2 != 0 else np.nan
        # or which dah value?
        dah_bad = bad / rs + m1 + ah[col] * m2
        tt, bb = row // B, row % B
        near = []
        for dt in (-2, -1, 1, 2):
            r2 = (tt + dt) * B + bb
            if 0 <= r2 < T * B:
                near.append((dt, c0['ahat'][r2, col], c0['dG'][r2, col] * c0['al'][col]))
        print('%s t=%d b=%d col=%d (unit %d gate %d): good %.9g bad %.9g formula %.9g | ah %.9g ah_needed_for_bad %.9g | dah %.9g dah_needed %.9g | m2 %.3g\n      neighbours (dt, ahat, dah): %s'
              % (key, tt, bb, col, col // 4, col % 4, good, bad, want, ah[col], ah_bad, dah[col], dah_bad, m2,
                 ', '.join('(%d %.6g %.6g)' % n for n in near)))
        shown += 1
    if shown > 12:
        break
