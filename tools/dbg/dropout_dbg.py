import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import numpy as np, torch
from asr_amd import gating_hyper_nets as G
from oracle import asr_oracle as O
from common import make_batch
from test_gpu_parity import perturb_params
for p in (0.0, 0.2):
    net = G.deep_projection_ln_lstm_model(None, None, 5, 6, 2, 4, 8, grad_clipping=1.0, dropout=p, device='cuda:0', seed=91)
    vals = perturb_params(net, 93, 0.1)
    b = make_batch(np.random.default_rng(94), 5, 9, 5, 6)
    x, m, y = (torch.from_numpy(b[k]).cuda() for k in ('x', 'mask', 'y'))
    onet = O.OracleNetwork(net.describe(), vals)
    B, T = b['mask'].shape
    for rep in range(3):
        ce, _, _ = net.loss(x, m, y, m, None, train=True)
        net.backward(); net.check_status()
        masks = [mk.cpu().numpy().reshape(T, B, -1).transpose(1, 0, 2) for mk in net.dropout_masks()] if p else None
        _, os_, _, og = onet.loss_and_grads(b['x'], b['mask'], b['y'], b['mask'], None, drop=masks)
        o32 = O.OracleNetwork(net.describe(), vals, dtype=np.float32)
        g32 = o32.loss_and_grads(b['x'], b['mask'], b['y'], b['mask'], None, drop=masks)[3]
        gmax = max(np.abs(g).max() for g in og)
        errs = []
        for (n, _), g, o, f in zip(net.param_names_shapes(), net.get_all_grad_values(), og, g32):
            d = np.abs(o).max() + 1e-3 * gmax
            errs.append((np.abs(g - o).max() / d, np.abs(f - o).max() / d, n))
        errs.sort(reverse=True)
        print('p=%.1f rep %d ce %.3e: ' % (p, rep, abs(ce.item() - os_) / abs(os_)), ['%s %.1e (f32 %.1e)' % (n, e, f) for e, f, n in errs[:4]])
