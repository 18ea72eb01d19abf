"""Feasibility probe: capture loss + backward of one TBPTT window of the cfg5 net in a HIP graph, replay it, compare the
flat gradient with the eager run and time both."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from asr_amd import builders, workloads
from asr_amd.utils import gen_win

dev = 'cuda:0'
net, _, tb = builders.build_deep_uni_lhuclstm_ivector_tbptt(
    'IVectorLHUCLSTMLayer', None, None, 123, 4, 512, 0, 3436, 1, 128, 0, 0, 0, 16, 1.0, False, False, 100, True,
    device=dev, seed=1)
b = workloads.make_batch(5, 16)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
full = (t(b['x']), t(b['mask']), t(b['ivec']), t(b['mask']), t(b['y']), t(b['mask']))
wins = list(gen_win(full, 100, 0))[:12]
for l in tb:
    l.reset(16)
# eager reference on window 0 (state zero)
xw, mw, iw, _, yw, tmw = wins[0]
sx, sm, si, sy, st = [torch.empty_like(a).copy_(a) for a in (xw, mw, iw, yw, tmw)]
for _ in range(2):
    for l in tb:
        l.reset(16)
    net.loss(sx, sm, sy, st, si, True, 100); net.backward()
torch.cuda.synchronize()
g_eager = net.store.g.clone()
ce_eager = float(net.scal[0])
for l in tb:
    l.reset(16)
g = torch.cuda.CUDAGraph()
net.scan_events = None
with torch.cuda.graph(g):
    net.loss(sx, sm, sy, st, si, True, 100); net.backward()
torch.cuda.synchronize()
for l in tb:
    l.reset(16)
net.store.g.zero_()
g.replay()
torch.cuda.synchronize()
print('graph vs eager: max |dg| = %.3e (max |g| %.3e), ce %.6f vs %.6f' % (
    float((net.store.g - g_eager).abs().max()), float(g_eager.abs().max()), float(net.scal[0]), ce_eager), flush=True)
net.check_status()
# timing: eager windows vs graph replays (inputs copied into the static buffers)
def run(mode, n=3):
    best = 1e9
    for _ in range(n):
        for l in tb:
            l.reset(16)
        torch.cuda.synchronize(); t0 = time.time()
        for xw, mw, iw, _, yw, tmw in wins[:10]:
            if mode == 'eager':
                net.loss(xw, mw, yw, tmw, iw, True, 100); net.backward()
            else:
                sx.copy_(xw); sm.copy_(mw); si.copy_(iw); sy.copy_(yw); st.copy_(tmw)
                g.replay()
            net.adam_update(1e-3)
        t_enq = (time.time() - t0) / 10
        torch.cuda.synchronize(); best = min(best, (time.time() - t0) / 10)
    print('%s: host enqueue %.3f ms per window' % (mode, t_enq * 1e3))
    return best
print('eager %.3f ms per window, graph %.3f ms per window' % (run('eager') * 1e3, run('graph') * 1e3), flush=True)
net.check_status()
