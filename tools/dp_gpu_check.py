"""Two ranks, ONE GPU, tiny network: the data-parallel path of asr_amd.dp on CUDA tensors (bucketed
all-reduce launched from the deferred side stream, loss scaled by 1/B_global) against the same global
batch run in one process.  Backend gloo (two ranks cannot share one device under RCCL); the networks
are tiny, so both ranks' persistent scans (2 workgroups each) are trivially co-resident.
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/dp_gpu_check.py"""
import faulthandler, os, sys
import numpy as np
faulthandler.dump_traceback_later(60, exit=True)      # a stalled rank says where it stands and leaves (normal run: 3 s)
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
# the inter / intra variance costs of scripts/hypernets/exp_prj_hyper.py are statistics of the GLOBAL batch: the ranks
# gather their per-sequence statistics (engine_prjhyper.PrjHyperPair.aux_costs).  Equal lengths, so that the padded
# frames the reference's unmasked sums run over are the same in both runs.
from asr_amd.gating_hyper_nets import deep_projection_hyper_model
mkp = lambda: deep_projection_hyper_model(None, None, 5, 6, 2, 4, 8, grad_clipping=1.0, device=dev, seed=9)
b = make_batch(np.random.default_rng(11), 4, 9, 5, 6)
b['mask'][:] = 1.0
full = (b['x'], b['mask'], None, None, b['y'], b['mask'])
ref, fi, _ = mkp()
ref.loss(t(full[0]), t(full[1]), t(full[4]), t(full[5]), None, train=True)
c_ref = fi.pair.aux_costs(0.7, 1.3).tolist()
ref.backward(); torch.cuda.synchronize(); ref.check_status()
g_ref = [g.copy() for g in ref.get_all_grad_values()]
net, fi, _ = mkp()
ar = dp.GradAllReduce(net)
sh = dp.shard_batch(full, rank, world)
net.loss(t(sh[0]), t(sh[1]), t(sh[4]), t(sh[5]), None, train=True)
c_dp = fi.pair.aux_costs(0.7, 1.3).tolist()
net.backward(); ar.finish(); torch.cuda.synchronize(); net.check_status()
gmax = max(np.max(np.abs(g)) for g in g_ref)
worst = max(float(np.max(np.abs(a - r)) / (np.max(np.abs(r)) + 1e-3 * gmax)) for a, r in zip(net.get_all_grad_values(), g_ref))
dc = max(abs(a - r) / max(1.0, abs(r)) for a, r in zip(c_dp, c_ref))
print('rank %d prjhyper + variance costs: worst relative gradient difference %.2e, costs %.2e' % (rank, worst, dc), flush=True)
ok = ok and worst < 2e-5 and dc < 1e-5
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
