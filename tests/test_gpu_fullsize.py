"""GPU parity at the BASELINE configs' real widths.

Two kinds of test:
  * oracle comparisons at full layer widths / class counts with the time axis truncated so the
    float64 numpy oracle finishes in seconds (cfg1 runs at its full TIMIT-shaped length);
  * size-independent properties at the full sequence lengths: row independence (a row's posteriors do
    not depend on which other rows share the batch -> exercises the 16-row chunking and padding),
    padding invariance, and data-parallel linearity (sum over ranks of the per-rank gradients, each
    normalised by the GLOBAL batch, equals the single-process gradient of the global batch:
    libs/comp_graph_utils.py:40).
Tolerances: forward 2e-5, gradients 1e-4 of the tensor max (see tests/test_gpu_parity.py)."""
import numpy as np
import pytest
import torch

from asr_amd import workloads
from oracle import asr_oracle as O
from test_gpu_parity import perturb_params, relerr, run_case

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _batch(cfg, B=16, max_len=None, seed_offset=0):
    b = workloads.make_batch(cfg, B, seed_offset=seed_offset, max_len=max_len)
    return b


def build_cfg(cfg, seed=11):
    from asr_amd import builders, gating_hyper_nets as G
    if cfg == 1:
        return builders.build_deep_lstm(None, None, 123, 3, 256, 0, 61, 1.0, True, False, 0, device=DEV, seed=seed), None
    if cfg == 2:
        return G.deep_projection_ln_lstm_model(None, None, 123, 3436, 4, None, 512, 1.0, 0.0, device=DEV,
                                               seed=seed), None
    if cfg == 3:
        return builders.build_deep_hyperlstm('hyper', None, None, 123, 4, 512, 128, 64, 3436, 1.0, True, 0,
                                             device=DEV, seed=seed), None
    if cfg == 4:
        net, _ = G.deep_projection_ivector_ln_model(None, None, None, 123, 4174, 4, 4, 64, 512, 1.0, 0.0,
                                                    cond_dim=100, device=DEV, seed=seed)
        return net, 'ivec'
    if cfg == 5:
        net, _, tb = builders.build_deep_uni_lhuclstm_ivector_tbptt(
            'IVectorLHUCLSTMLayer', None, None, 123, 4, 512, 0, 3436, 1, 128, 0, 0, 0, 16, 1.0, False, False,
            100, True, device=DEV, seed=seed)
        return net, 'ivec'
    raise KeyError(cfg)


def test_cfg1_full_size_forward_vs_oracle():
    """3x256 bidirectional LSTM, B=16, TIMIT-shaped lengths U{90..780}, C=61: full size."""
    net, _ = build_cfg(1)
    vals = perturb_params(net, 21, 0.02)
    b = _batch(1)
    post = net.posteriors(_t(b['x']), _t(b['mask'])).cpu().numpy()
    net.check_status()
    opost = O.OracleNetwork(net.describe(), vals).posteriors(b['x'], b['mask'])
    valid = b['mask'][:, :, None] > 0
    assert relerr(post * valid, opost * valid) < 2e-5
    lp = np.log(post[valid[:, :, 0]]); olp = np.log(opost[valid[:, :, 0]])
    assert np.max(np.abs(lp - olp)) < 2e-5 * np.max(np.abs(olp))


@pytest.mark.parametrize('cfg,max_len', [(2, 40), (3, 80), (4, 40)])
def test_full_width_truncated_vs_oracle(cfg, max_len):
    net, cond_key = build_cfg(cfg)
    perturb_params(net, 30 + cfg, 0.02)
    b = _batch(cfg, max_len=max_len)
    # LN-LSTMP stacks at these widths amplify fp32 rounding noise (measured: the float32 numpy run of
    # the oracle is 2e-4 off its float64 run at T=40, 7e-3 at T=96), so the bound is calibrated on
    # that noise floor; plain and Hyper LSTM stacks meet the fixed 2e-5 / 1e-4.
    e, worst = run_case(net, b, cond_key=cond_key, tol_f=2e-5, tol_g=1e-4, calibrate=(cfg != 3))
    print('cfg%d posteriors rel err %.2e, worst gradient rel err %.2e' % (cfg, e, worst))


def test_cfg5_full_width_tbptt_windows_vs_oracle():
    """LHUC on stateful uni LSTM, 4x512, window 100: 3 windows of a T=230 batch with carried state."""
    net, _ = build_cfg(5)
    vals = perturb_params(net, 35, 0.02)
    b = _batch(5, max_len=230)
    b['lengths'][3] = 150; b['mask'][3, 150:] = 0; b['x'][3, 150:] = 0      # a row that ends inside window 2
    onet = O.OracleNetwork(net.describe(), vals)
    for l in net.tbptt_layers():
        l.reset(16)
    for l in onet.tbptt_layers():
        l.reset(16)
    full = (b['x'], b['mask'], b['ivec'], b['mask'], b['y'], b['mask'])
    for w, (xw, mw, iw, _, yw, tmw) in enumerate(O.gen_win(full, 100, 0)):
        ce, ok, pred = net.loss(_t(xw), _t(mw), _t(yw), _t(tmw), _t(iw), True, 100)
        net.backward()
        net.check_status()
        _, os_, opred, ograds = onet.loss_and_grads(xw, mw, yw, tmw, iw, context=100)
        assert abs(ce.item() - os_) < 2e-5 * abs(os_)
        gmax = max(np.max(np.abs(g)) for g in ograds)
        for g, og in zip(net.get_all_grad_values(), ograds):
            assert np.max(np.abs(g - og)) / (np.max(np.abs(og)) + 1e-3 * gmax) < 1e-4, w


@pytest.mark.parametrize('cfg', [2, 3, 4])
def test_full_size_row_independence_and_padding(cfg):
    """Full sequence lengths.  Rows 4..7 run alone (B=4, padded only to their own max length) give the
    posteriors they get inside the B=16 batch."""
    net, cond_key = build_cfg(cfg)
    perturb_params(net, 40 + cfg, 0.02)
    b = _batch(cfg)
    cond = _t(b[cond_key]) if cond_key else None
    full = net.posteriors(_t(b['x']), _t(b['mask']), cond).cpu().numpy()
    net.check_status()
    rows = slice(4, 8)
    Ts = int(b['lengths'][rows].max())
    sub = net.posteriors(_t(b['x'][rows, :Ts]), _t(b['mask'][rows, :Ts]),
                         _t(b[cond_key][rows, :Ts]) if cond_key else None).cpu().numpy()
    net.check_status()
    valid = b['mask'][rows, :Ts, None] > 0
    # same kernels, same per-row arithmetic: equal up to split-K ordering of the input GEMMs
    assert relerr(sub * valid, full[rows, :Ts] * valid) < 2e-6


def test_cfg3_full_size_data_parallel_linearity():
    """One process plays two ranks: each half of the global batch (B=16 -> 2 x 8 rows, sharded by length
    as asr_amd.dp.shard_batch does) is run with world=2; the summed gradients equal the
    single-process gradient of the global batch, and the summed ce_frame_sum equals the global one."""
    from asr_amd import dp
    net, _ = build_cfg(3)
    perturb_params(net, 50, 0.02)
    b = _batch(3)
    m = b['mask']
    ce, _, _ = net.loss(_t(b['x']), _t(m), _t(b['y']), _t(m))
    net.backward()
    net.check_status()
    g_full = net.store.g.clone(); ce_full = ce.item()
    net.world = 2
    acc = torch.zeros_like(g_full); ce_sum = 0.0; frames = 0
    for r in range(2):
        x, mm, y = dp.shard_batch((b['x'], m, b['y']), r, 2)
        assert x.shape[0] == 8
        frames += int(mm.sum())
        ce, _, _ = net.loss(_t(x), _t(mm), _t(y), _t(mm))
        net.backward()
        net.check_status()
        acc += net.store.g; ce_sum += ce.item()
    net.world = 1
    assert frames == b['frames']
    assert abs(ce_sum - ce_full) < 1e-5 * abs(ce_full)
    err = (acc - g_full).abs().max().item() / g_full.abs().max().item()
    assert err < 2e-5, err


def test_projection_hyper_reference_width_vs_oracle():
    """scripts/hypernets/exp_prj_hyper.py:30-34 defaults: 3 layers, 512 units, 256 factors, clip 1; B=16,
    time axis truncated to 40 frames so the numpy oracle finishes in seconds."""
    from asr_amd import gating_hyper_nets as G
    net = G.deep_projection_hyper_model(None, None, 123, 3436, 3, 256, 512, grad_clipping=1.0, device=DEV, seed=31)[0]
    perturb_params(net, 32, 0.01)
    b = _batch(3, max_len=40)
    run_case(net, b, calibrate=True)
