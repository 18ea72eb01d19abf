"""Diagnostic: per-phase cycle shares of the persistent HyperLSTM scan (needs the stamps build:
python -m asr_amd.build --stamps; ASR_AMD_LIB=asr_amd/libasr_amd_stamps.so python tools/scan_stamps.py).
Never used for reported numbers (the stamped build forbids overlaps the real kernel has)."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asr_amd.builders import build_deep_hyperlstm
from asr_amd import workloads, _lib

net = build_deep_hyperlstm('hyper', None, None, 123, 1, 512, 128, 64, 3436, 1.0, True, 0, device='cuda:0', seed=1)
b = workloads.make_batch(3, 16, 0)
x = torch.from_numpy(b['x']).cuda(); m = torch.from_numpy(b['mask']).cuda(); y = torch.from_numpy(b['y']).cuda()
T = x.shape[1]
for it in range(2):
    net.loss(x, m, y, m); net.backward()
net.check_status()
L = _lib.lib()
rec = dict(workload='cfg3 layer 0 (123 -> 512/128/64 bidirectional), B=16', T=int(T), clock_note='s_memtime ticks (100 MHz domain scaled by hipcc: ~2.1 GHz shader clock)', kernels={})
for bw, names in ((False, ['wait_h', 'ph1_mfma', 'epi_hyper', 'publish1', 'wait_hh', 'ph2_mfma', 'epi_main', 'publish2', '-', '-', '-', '-', '-', '-', '-', '-']),
                  (True, ['S1_main', 'S2_mfma', 'publish1', 'wait1', 'S3_gather', 'S4_hyper', 'publish2', 'wait2', 'S5_tail', 'prefetch', 'da_store', 'bulk_st', 'da_mfma', 'S5_ld_mfma', 'S5_lds_bar', 'S3_dhx'])):
    need = L.asr_hyper_scan_ws_bytes(T, 16, 512, 128, 2, int(bw))
    ws = net._ws[('hyper', 0, bw)]
    raw = ws[need - 65536:need].view(torch.int64).cpu().numpy()[:256 * 16].reshape(256, 16)
    per_step = raw / T
    tot = per_step.sum(1)
    print('backward' if bw else 'forward', 'T=%d cycles/step median over WGs = %.0f (%.2f us @2.1GHz)' % (T, np.median(tot), np.median(tot) / 2100))
    k = rec['kernels']['hyper_scan_bwd_kernel' if bw else 'hyper_scan_fwd_kernel'] = dict(cycles_per_step_median=float(np.median(tot)), phases={})
    for i, n in enumerate(names):
        if n != '-':
            print('   %-10s med %7.0f  min %7.0f  max %7.0f cycles/step' % (n, np.median(per_step[:, i]), per_step[:, i].min(), per_step[:, i].max()))
            k['phases'][n] = dict(median=float(np.median(per_step[:, i])), min=float(per_step[:, i].min()), max=float(per_step[:, i].max()))
if '--json' in sys.argv:
    with open(sys.argv[sys.argv.index('--json') + 1], 'w') as f:
        json.dump(rec, f, indent=1)
