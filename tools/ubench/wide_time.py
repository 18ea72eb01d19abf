"""Per-step time of the LSTM scans on layers wider than the resident-weight limit (weights streamed from L2 / Infinity
Cache): one unidirectional layer, B = 16, T = 200, forward + backward, median of 5.
  python tools/ubench/wide_time.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
from asr_amd.builders import build_deep_lstm
from common import make_batch

T, B = 200, 16
for N in (1024, 1280, 1536, 2048, 4096):
    net = build_deep_lstm(None, None, 40, 1, N, 0, 50, 1.0, False, False, 0, device='cuda:0', seed=1)
    b = make_batch(np.random.default_rng(1), B, T, 40, 50, lengths=[T] * B)
    x, m, y = (torch.from_numpy(b[k]).cuda() for k in ('x', 'mask', 'y'))
    ts = []
    for r in range(7):
        net.scan_events = []
        net.loss(x, m, y, m); net.backward(); torch.cuda.synchronize(); net.check_status()
        ev = net.scan_events
        ts.append([e0.elapsed_time(e1) for k, bw, rows, ndir, e0, e1 in ev])
    med = np.median(np.array(ts[2:]), axis=0)
    print('N=%5d  forward scan %.2f us/step   backward scan %.2f us/step' % (N, med[0] * 1e3 / T, med[1] * 1e3 / T), flush=True)
