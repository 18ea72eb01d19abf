"""GPU parity of the recurrent_layers.py options (libs/lasagne_libs/recurrent_layers.py:52-57): peepholes (:112-123,
:190-192,:212-213), learn_init (:125-135,:246-248), only_return_hidden=False (:275-288) in whole networks against the
oracle's variant-table layer (pinned with the same options by torch autograd, tests/test_oracle_autograd.py
::test_recurrent_layers_options), and only_return_final (:265-266) on the layer itself with arbitrary output gradients,
which also reach the learned initial rows through the padding of the reverse direction."""
import numpy as np
import pytest
import torch

from asr_amd import specs
from common import make_batch
from oracle import asr_oracle as O
from test_gpu_parity import perturb_params, relerr, run_case

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'

NET_CASES = [
    ('LSTMLayer', dict(peepholes=True)),
    ('LSTMLayer', dict(learn_init=True)),
    ('ProjectLSTMLayer', dict(peepholes=True, learn_init=True)),
    ('ProjectLSTM_v0_1_Layer', dict(peepholes=True, only_return_hidden=False)),
    ('RegularLSTM_v0_1_Layer', dict(peepholes=True, learn_init=True, only_return_hidden=False)),
    ('LayerNormProjectLSTMLayer', dict(learn_init=True)),
    ('LayerNormProjectLSTM_v0_1_Layer', dict(learn_init=True, only_return_hidden=False)),
    ('LayerNormProjectLSTM_v0_2_Layer', dict(learn_init=True)),
    ('LayerNormLSTMLayer', dict(learn_init=True)),
    ('LayerNormRegularLSTM_v0_1_Layer', dict(learn_init=True, only_return_hidden=False)),
    ('LayerNormRegularLSTM_v0_2_Layer', dict(learn_init=True)),
    ('CondLayerNormProjectLSTMLayer', dict(only_return_hidden=False)),
    ('InputCondLayerNormLSTMLayer', dict(only_return_hidden=False)),
]


def _ids(cases):
    return ['%s-%s' % (c, '+'.join(sorted(k for k in o))) for c, o in cases]


@pytest.mark.parametrize('cls,opts', NET_CASES, ids=_ids(NET_CASES))
def test_recurrent_layer_options_small_net(cls, opts):
    from asr_amd.gating_hyper_nets import _build_rec
    v = specs.REC_CLASSES[cls]
    I = 3 if v['cond'] == 'ext' else 0
    net, _ = _build_rec(5, 6, 2, 8, 0.05, 0.0, cls, None, 0, num_factors=3 if v['cond'] else 4, num_prjs=4, cond_dim=I,
                        device=DEV, seed=3, rec_options=opts)
    if I:
        net.uses_cond = True
    perturb_params(net, 31, 0.15)
    b = make_batch(np.random.default_rng(32), 3, 7, 5, 6, I=I)
    run_case(net, b, cond_key='ivec' if I else None)


@pytest.mark.parametrize('cls,opts', [('ProjectLSTMLayer', dict(peepholes=True, learn_init=True, only_return_hidden=False)),
                                      ('LayerNormProjectLSTMLayer', dict(learn_init=True, only_return_hidden=False)),
                                      ('LSTMLayer', dict(peepholes=True, learn_init=True))])
def test_recurrent_layer_options_medium(cls, opts):
    """N = 64 (16 workgroups per direction), 18 rows (two 16-row chunks), T = 29."""
    from asr_amd.gating_hyper_nets import _build_rec
    net, _ = _build_rec(11, 13, 2, 64, 1.0, 0.0, cls, None, 0, num_factors=32, num_prjs=32, device=DEV, seed=4,
                        rec_options=opts)
    perturb_params(net, 33, 0.03)
    b = make_batch(np.random.default_rng(34), 18, 29, 11, 13)
    run_case(net, b, calibrate=True)


def test_recurrent_layer_options_two_tiles_per_workgroup():
    """1024 units in one direction: 256 workgroups of two unit tiles (UT = 2 instantiation of the option kernels)."""
    from asr_amd.gating_hyper_nets import _build_rec
    net, _ = _build_rec(9, 11, 1, 1024, 1.0, 0.0, 'LSTMLayer', None, 0, is_bidir=False, device=DEV, seed=5,
                        rec_options=dict(peepholes=True, learn_init=True))
    perturb_params(net, 35, 0.01)
    b = make_batch(np.random.default_rng(36), 5, 9, 9, 11)
    run_case(net, b, calibrate=True)


LAYER_CASES = [
    ('LSTMLayer', dict(only_return_final=True)),
    ('RegularLSTM_v0_1_Layer', dict(peepholes=True, learn_init=True, only_return_final=True)),
    ('ProjectLSTMLayer', dict(learn_init=True)),                       # hid_init through the reverse direction's padding
    ('ProjectLSTMLayer', dict(learn_init=True, only_return_final=True, only_return_hidden=False)),
    ('LayerNormProjectLSTMLayer', dict(learn_init=True, only_return_final=True)),
    ('LayerNormLSTMLayer', dict(learn_init=True, only_return_final=True, only_return_hidden=False)),
    ('LayerNormRegularLSTM_v0_2_Layer', dict(learn_init=True)),
    ('CondLayerNormProjectLSTM_v0_2_Layer', dict(only_return_final=True, only_return_hidden=False)),
]


@pytest.mark.parametrize('cls,opts', LAYER_CASES, ids=_ids(LAYER_CASES))
def test_recurrent_layer_options_on_the_layer(cls, opts):
    """RecPair.forward / backward called directly with a random gradient on EVERY output element (padding included),
    both directions, against one oracle layer per direction."""
    from asr_amd.engine import Network
    from asr_amd.engine_rec import RecPair
    B, T, D, N, C = 5, 9, 6, 8, 4
    v = specs.REC_CLASSES[cls]
    net = Network(DEV)
    pair = RecPair(net, 0, cls, D, N, 0.05, 2, num_factors=3 if v['cond'] else 4, num_prjs=4, **opts)
    net.add_pair(pair, [pair.spec_per_dir()] * 2)
    net.set_output(pair.out_dim, C, 'logits')
    net.finalize(7)
    perturb_params(net, 41, 0.2)
    rng = np.random.default_rng(42)
    b = make_batch(rng, B, T, D, C, lengths=[9, 4, 1, 7, 9])
    x, mask = b['x'].astype(np.float64), b['mask'].astype(np.float64)
    vals = net.get_all_param_values()
    desc = net.describe()['layers'][0]
    lays, k = [], 0
    for d in ('fwd', 'bwd'):
        P = {}
        for nm in desc[d]['params']:
            P[nm] = np.asarray(vals[k], np.float64); k += 1
        lays.append(O.RecLayerO(P, **desc[d]['args']))
        lays[-1].names = desc[d]['params']
    ys = [l.forward(x, mask) for l in lays]
    y_o = np.concatenate(ys, -1)
    w = rng.standard_normal(y_o.shape)
    split = ys[0].shape[-1]
    dx_o = lays[0].backward(w[..., :split]) + lays[1].backward(w[..., split:])
    g_o = [l.grads[n] for l in lays for n in l.names]

    X = torch.from_numpy(np.ascontiguousarray(b['x'].transpose(1, 0, 2).reshape(T * B, D))).to(DEV)
    m = torch.from_numpy(b['mask']).to(DEV)
    Y = pair.forward(X, T, B, m, None, True)
    net.join_deferred()
    y = Y.cpu().numpy()
    if not opts.get('only_return_final'):
        y = y.reshape(T, B, -1).transpose(1, 0, 2)
        wt = np.ascontiguousarray(w.transpose(1, 0, 2).reshape(T * B, -1))
    else:
        wt = w
    assert y.shape == y_o.shape
    assert relerr(y, y_o) < 2e-5, relerr(y, y_o)
    dX = pair.backward(torch.from_numpy(wt.astype(np.float32)).to(DEV), need_dx=True)
    net.join_deferred()
    net.check_status()
    dx = dX.cpu().numpy().reshape(T, B, D).transpose(1, 0, 2)
    assert relerr(dx, dx_o) < 1e-4, relerr(dx, dx_o)
    grads = net.get_all_grad_values()[:len(g_o)]
    names = [n for n, _ in net.param_names_shapes()]
    gmax = max(np.max(np.abs(g)) for g in g_o)
    for n, g, og in zip(names, grads, g_o):
        assert g.shape == og.shape, (n, g.shape, og.shape)
        err = float(np.max(np.abs(g - og)) / (np.max(np.abs(og)) + 1e-3 * gmax))
        assert err < 1e-4, (n, err)
