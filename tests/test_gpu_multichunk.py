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
