"""Batches above 16 rows in the persistent scans (the strong-scaling leg, global batch 128).

theano.scan steps any batch size at once (libs/lasagne_libs/hyper_lhuc_layers.py:493-500,
libs/lhuclstm_layers.py:205-211).  Here a workgroup carries up to four 16-row chunks per step as teams of four
waves on the same resident weights, with one publish / wait per exchange for all of them
(asr_scan_set_chunks_per_wg).  Checked:
  * per-row results do not depend on how many chunks a workgroup carries: posteriors, loss and every gradient of
    the same batch are BIT-IDENTICAL for 1, 2 and 4 chunks per workgroup (full width of cfg3 and small widths,
    batch sizes with a partly filled last chunk and with an empty team);
  * parity against the float64 oracle at B = 18 and B = 67 (tolerances of tests/test_gpu_parity.py)."""
import numpy as np
import pytest
import torch

from common import make_batch
from test_gpu_parity import perturb_params, run_case

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _set_nc(nc):
    from asr_amd import _lib
    _lib.check(_lib.lib().asr_scan_set_chunks_per_wg(nc), 'asr_scan_set_chunks_per_wg')


def _step(net, b, cond_key=None):
    cond = _t(b[cond_key]) if cond_key else None
    x, m, y = _t(b['x']), _t(b['mask']), _t(b['y'])
    post = net.posteriors(x, m, cond).clone()
    ce, ok, pred = net.loss(x, m, y, m, cond)
    net.backward()
    net.check_status()
    return post, ce.clone(), pred.clone(), net.store.g.clone()


def _same_for_every_nc(net, b, cond_key=None):
    try:
        ref = None
        for nc in (1, 2, 4):
            _set_nc(nc)
            got = _step(net, b, cond_key)
            if ref is None:
                ref = got
                continue
            for name, a, r in zip(('posteriors', 'loss', 'argmax', 'gradients'), got, ref):
                assert torch.equal(a, r), (nc, name, (a.float() - r.float()).abs().max().item())
    finally:
        _set_nc(0)


@pytest.mark.parametrize('B', [32, 40, 64, 67])
def test_hyperlstm_chunks_per_workgroup_bit_identical_small(B):
    """N = 64, H = 16: 16 workgroups per direction, so several z slices fit one launch as well."""
    from asr_amd.builders import build_deep_hyperlstm
    net = build_deep_hyperlstm('hyper', None, None, 9, 2, 64, 16, 8, 11, 1.0, True, 0, device=DEV, seed=5)
    perturb_params(net, 6, 0.05)
    b = make_batch(np.random.default_rng(B), B, 37, 9, 11)
    _same_for_every_nc(net, b)


@pytest.mark.parametrize('B', [48, 64])
def test_hyperlstm_chunks_per_workgroup_bit_identical_full_width(B):
    """cfg3's layer width (512 / 128 / 64, bidirectional = 256 workgroups, one per CU), 2 layers, T = 90."""
    from asr_amd.builders import build_deep_hyperlstm
    net = build_deep_hyperlstm('hyper', None, None, 123, 2, 512, 128, 64, 64, 1.0, True, 0, device=DEV, seed=7)
    perturb_params(net, 8, 0.02)
    b = make_batch(np.random.default_rng(B + 1), B, 90, 123, 64)
    _same_for_every_nc(net, b)


@pytest.mark.parametrize('B', [18, 67])
def test_hyperlstm_more_than_16_rows_vs_oracle(B):
    from asr_amd.builders import build_deep_hyperlstm
    net = build_deep_hyperlstm('hyper', None, None, 9, 2, 32, 8, 4, 11, 0.5, True, 0, device=DEV, seed=9)
    perturb_params(net, 10, 0.05)
    b = make_batch(np.random.default_rng(100 + B), B, 23, 9, 11)
    run_case(net, b)


@pytest.mark.parametrize('kind', ['lstm', 'lstm_uni_1024', 'ln_lstmp', 'ln_lstmp_gate_ln', 'ivec_ln'])
@pytest.mark.parametrize('B', [40, 64])
def test_lstm_family_chunks_per_workgroup_bit_identical(kind, B):
    """Full layer widths, so that a 16-row chunk fills the chip (512 units bidirectional = 256 workgroups) and further
    chunks ride as teams: plain LSTM (generic kernels), 1024 units unidirectional (two tiles per workgroup), LN-LSTMP
    (cell-LN forward kernel + 2-D backward), its per-gate-LN sibling (per-gate statistics), i-vector cond-LN."""
    from asr_amd import builders, gating_hyper_nets as G
    ck = None
    if kind == 'lstm':
        net = builders.build_deep_lstm(None, None, 9, 2, 512, 0, 11, 1.0, True, False, 0, device=DEV, seed=3)
    elif kind == 'lstm_uni_1024':
        net = builders.build_deep_lstm(None, None, 9, 1, 1024, 0, 11, 1.0, False, False, 0, device=DEV, seed=3)
    elif kind == 'ln_lstmp':
        net = G.deep_projection_ln_lstm_model(None, None, 9, 11, 2, None, 512, grad_clipping=1.0, dropout=0.0,
                                              device=DEV, seed=3)
    elif kind == 'ln_lstmp_gate_ln':
        net = G.deep_prj_lstm_ln_model_v2(None, None, 9, 11, 2, 512, 256, grad_clipping=1.0, dropout=0.0,
                                          device=DEV, seed=3)
    else:
        net, _ = G.deep_projection_ivector_ln_model(None, None, None, 9, 11, 2, 2, 8, 512, 1.0, 0.0, cond_dim=5,
                                                    device=DEV, seed=3)
        ck = 'ivec'
    perturb_params(net, 12, 0.02)
    b = make_batch(np.random.default_rng(B + 7), B, 14, 9, 11, I=5 if ck else 0)
    _same_for_every_nc(net, b, ck)


@pytest.mark.parametrize('kind', ['lstm', 'ln_lstmp'])
def test_lstm_family_40_rows_full_width_vs_oracle(kind):
    from asr_amd import builders, gating_hyper_nets as G
    if kind == 'lstm':
        net = builders.build_deep_lstm(None, None, 9, 1, 512, 0, 11, 1.0, True, False, 0, device=DEV, seed=4)
    else:
        net = G.deep_projection_ln_lstm_model(None, None, 9, 11, 1, None, 512, grad_clipping=1.0, dropout=0.0,
                                              device=DEV, seed=4)
    perturb_params(net, 13, 0.02)
    b = make_batch(np.random.default_rng(77), 40, 12, 9, 11)
    run_case(net, b, calibrate=(kind != 'lstm'))


@pytest.mark.parametrize('kind', ['hyper', 'ln_lstmp'])
def test_micro_batches_accumulate_to_the_whole_batch_gradient(kind):
    """bench.py's strong leg (--global-batch 128 on few GPUs): a rank's rows sorted by length, cut into micro-batches padded
    to their own longest row, each pass scaled by 1 / (rows of all parts) (Network.world), gradients accumulated on the
    device (Network.grad_accumulate): the sum is the gradient of the whole batch in one pass (fp32 summation order aside)
    and matches the oracle on the whole batch."""
    from asr_amd import builders, gating_hyper_nets as G
    from oracle import asr_oracle as O
    if kind == 'hyper':
        net = builders.build_deep_hyperlstm('hyper', None, None, 9, 2, 32, 8, 4, 11, 1.0, True, 0, device=DEV, seed=15)
    else:
        net = G.deep_projection_ln_lstm_model(None, None, 9, 11, 2, None, 32, grad_clipping=1.0, dropout=0.0, device=DEV, seed=15)
    perturb_params(net, 16, 0.05)
    B, T, parts = 40, 21, 2
    b = make_batch(np.random.default_rng(17), B, T, 9, 11)
    x, m, y = _t(b['x']), _t(b['mask']), _t(b['y'])
    ce_all, _, _ = net.loss(x, m, y, m)
    net.backward(); net.check_status()
    g_all = net.store.g.clone(); ce_all = ce_all.item()
    order = np.argsort(-b['lengths'], kind='stable')
    net.world = parts
    ce_sum = 0.0
    for k in range(parts):
        rows = order[k * (B // parts):(k + 1) * (B // parts)]
        Tk = int(b['lengths'][rows].max())
        xs, ms, ys = (_t(b[n][rows, :Tk]) for n in ('x', 'mask', 'y'))
        ce, _, _ = net.loss(xs, ms, ys, ms)
        net.backward()
        net.grad_accumulate(first=(k == 0))
        ce_sum += ce.item()
    net.grad_accumulated(); net.check_status()
    net.world = 1
    g_acc = net.store.g.clone()
    assert abs(ce_sum - ce_all) <= 2e-5 * abs(ce_all)            # the frame sums add up
    scale = g_all.abs().max().item()
    assert (g_acc - g_all).abs().max().item() <= 2e-5 * scale, (g_acc - g_all).abs().max().item() / scale
    onet = O.OracleNetwork(net.describe(), net.get_all_param_values())
    _, _, _, og = onet.loss_and_grads(b['x'], b['mask'], b['y'], b['mask'])
    gmax = max(np.abs(g).max() for g in og)
    for (nm, _), g, o in zip(net.param_names_shapes(), net.get_all_grad_values(), og):
        assert np.abs(g - o).max() / (np.abs(o).max() + 1e-3 * gmax) < 1e-4, nm
