"""Diagnostic: per-phase cycle shares of the persistent LSTM scan on one cfg2 layer (512-unit LN-LSTMP,
bidirectional).  Needs the stamps build: python -m asr_amd.build --stamps;
ASR_AMD_LIB=asr_amd/libasr_amd_stamps.so python tools/scan_stamps_lstm.py.  Never used for reported numbers."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asr_amd import gating_hyper_nets as G, workloads, _lib

net = G.deep_projection_ln_lstm_model(None, None, 123, 3436, 1, None, 512, 1.0, 0.0, device='cuda:0', seed=1)
b = workloads.make_batch(2, 16, 0)
x = torch.from_numpy(b['x']).cuda(); m = torch.from_numpy(b['mask']).cuda(); y = torch.from_numpy(b['y']).cuda()
T = x.shape[1]
for it in range(2):
    net.loss(x, m, y, m); net.backward()
net.check_status()
L = _lib.lib()
print([k for k in net._ws.keys()])
for bw, names in ((False, ['wait_r', 'mfma', 'xchg_gateLN', 'gates', 'xchg_cellLN', 'r_store', 'publish', '-', '-', '-', '-', '-']),
                  (True, ['wait_dr', 'gather', 'xchg_cellLN', 'gates', 'xchg_gateLN', 'dA_store', 'mfma_st', 'publish', '-', '-', '-', '-'])):
    need = L.asr_lstm_scan_ws_bytes(T, 16, 512, 2, int(bw))
    key = [k for k in net._ws.keys() if k[0] == 'lstm' and k[-1] == bw][0]
    ws = net._ws[key]
    raw = ws[need - 65536:need].view(torch.int64).cpu().numpy()[:256 * 16].reshape(256, 16)
    per_step = raw / T
    tot = per_step.sum(1)
    print('backward' if bw else 'forward', 'T=%d cycles/step median over WGs = %.0f (%.2f us @2.1GHz)' % (T, np.median(tot), np.median(tot) / 2100))
    for i, n in enumerate(names):
        if n != '-':
            print('   %-12s med %7.0f  min %7.0f  max %7.0f cycles/step' % (n, np.median(per_step[:, i]), per_step[:, i].min(), per_step[:, i].max()))
