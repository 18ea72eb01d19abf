import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from asr_amd import gating_hyper_nets as G, workloads
from asr_amd.engine import Network, LstmPair
from asr_amd import specs
from oracle import asr_oracle as O

def run(kind, N, P, L=1, T=40, B=16, C=50, D=123, ndir=2, pert=0.02):
    net = Network('cuda:0')
    Din = D
    for l in range(L):
        if kind == 'ln_prj':
            pair = LstmPair(net, len(net.pairs), 'ln_prj', Din, N, P, True, 1.0, ndir)
            net.add_pair(pair, [specs.ln_prj_params(Din, N, P)] * ndir)
        elif kind == 'prj':
            pair = LstmPair(net, len(net.pairs), 'prj', Din, N, P, False, 1.0, ndir)
            net.add_pair(pair, [specs.prj_params(Din, N, P)] * ndir)
        Din = pair.out_dim
    net.set_output(Din, C, 'logits'); net.finalize(1)
    rng = np.random.default_rng(3)
    vals = [(v + rng.standard_normal(v.shape) * pert).astype(np.float32) for v in net.get_all_param_values()]
    net.set_all_param_values(vals)
    b = workloads.make_batch(2, B, max_len=T)
    x = torch.from_numpy(b['x']).cuda(); m = torch.from_numpy(b['mask']).cuda()
    post = net.posteriors(x, m).cpu().numpy(); net.check_status()
    on = O.OracleNetwork(net.describe(), vals)
    op = on.posteriors(b['x'], b['mask'])
    Tt = x.shape[1]
    errs = []
    for l, pair in enumerate(net.pairs):
        Y = pair.ctx['Y'].view(Tt, B, -1).transpose(0, 1).cpu().numpy()
        oy = on.acts[l + 1]
        errs.append(float(np.abs(Y - oy).max() / np.abs(oy).max()))
    on32 = O.OracleNetwork(net.describe(), vals, dtype=np.float32); on32.posteriors(b['x'], b['mask'])
    e32 = ['%.1e' % float(np.abs(on32.acts[l+1] - on.acts[l+1]).max() / np.abs(on.acts[l+1]).max()) for l in range(len(net.pairs))]
    print('   numpy-fp32 oracle vs fp64 layer errs', e32)
    print(kind, 'N', N, 'P', P, 'L', L, 'ndir', ndir, 'T', Tt, 'post err %.2e' % (np.abs(post - op).max() / np.abs(op).max()), 'layer errs', ['%.1e' % e for e in errs], flush=True)


run('ln_prj', 512, 256, L=4, T=96, C=3436)
run('ln_prj', 512, 256, L=4, T=96, C=3436, pert=0.005)
run('ln_prj', 512, 256, L=4, T=40, C=50)
run('prj', 512, 256, L=4, T=96, C=50)
