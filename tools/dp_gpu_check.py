"""Two ranks, ONE GPU, tiny network: the data-parallel path of asr_amd.dp on CUDA tensors (bucketed
all-reduce launched from the deferred side stream, loss scaled by 1/B_global) against the same global
batch run in one process.  Backend gloo (two ranks cannot share one device under RCCL); the networks
are tiny, so both ranks' persistent scans (2 workgroups each) are trivially co-resident.
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/dp_gpu_check.py"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from asr_amd import dp
from asr_amd.builders import build_deep_hyperlstm, build_deep_lstm
from common import make_batch

dist.init_process_group('gloo')
rank, world = dist.get_rank(), dist.get_world_size()
dev = 'cuda:0'
ok = True
for name, mk in (('hyper', lambda: build_deep_hyperlstm('h', None, None, 5, 2, 8, 4, 2, 6, 1.0, True, 0, device=dev, seed=3)),
                 ('lstm', lambda: build_deep_lstm(None, None, 5, 2, 8, 4, 6, 1.0, True, True, 0, device=dev, seed=4))):
    b = make_batch(np.random.default_rng(7), 4, 9, 5, 6)
    full = (b['x'], b['mask'], None, None, b['y'], b['mask'])
    ref = mk()                                                    # one process, global batch
    t = lambda a, dt=None: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    ref.loss(t(full[0]), t(full[1]), t(full[4]), t(full[5])); ref.backward()
    torch.cuda.synchronize()
    g_ref = [g.copy() for g in ref.get_all_grad_values()]
    net = mk()
    ar = dp.GradAllReduce(net)
    assert net.world == world
    sh = dp.shard_batch(full, rank, world)
    ce, _, _ = net.loss(t(sh[0]), t(sh[1]), t(sh[4]), t(sh[5])); net.backward(); ar.finish()
    torch.cuda.synchronize(); net.check_status()
    worst = 0.0
    gmax = max(np.max(np.abs(g)) for g in g_ref)
    for a, r in zip(net.get_all_grad_values(), g_ref):
        worst = max(worst, float(np.max(np.abs(a - r)) / (np.max(np.abs(r)) + 1e-3 * gmax)))
    print('rank %d %s: worst relative gradient difference %.2e' % (rank, name, worst), flush=True)
    ok = ok and worst < 2e-5
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
