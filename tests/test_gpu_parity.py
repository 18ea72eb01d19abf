"""GPU parity tests proper: the HIP path (through the C-ABI) against the float64 oracle on the same
seeded inputs and weights -- posteriors, loss, argmax, every parameter gradient, Adam.
Tolerance (BASELINE.json north_star): 1e-5 relative in fp32; we measure max|a-b| / max|b| per tensor
and allow 2e-5 (fp32 accumulation over T dependent steps) for forward quantities and 1e-4 for
gradients of the deepest nets (documented per test)."""
import numpy as np
import pytest
import torch

from common import make_batch, init_value
from oracle import asr_oracle as O

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    return 'cuda:0'


def relerr(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))


def perturb_params(net, seed, perturb=0.1, scale=1.0):
    """Reference init + N(0, perturb) on every parameter (mandatory for the adaptive
    paths, whose reference init is an exact identity)."""
    rng = np.random.default_rng(seed)
    vals = net.get_all_param_values()
    new = [(v * scale + rng.standard_normal(v.shape) * perturb).astype(np.float32) for v in vals]
    net.set_all_param_values(new)
    return new


def fp32_noise(desc, vals, batch, cond_key, context=0):
    """How far the SAME restatement run in float32 numpy lands from its float64 run: the rounding
    -noise amplification of the network itself (LN-LSTMP stacks amplify fp32 noise by ~4x per layer
    and exponentially in T), which no fp32 implementation -- the reference's included -- can beat."""
    o64 = O.OracleNetwork(desc, vals); o32 = O.OracleNetwork(desc, vals, dtype=np.float32)
    for o in (o64, o32):
        for l in o.tbptt_layers():
            l.reset(batch['x'].shape[0])
    c = batch.get(cond_key) if cond_key else None
    valid = batch['mask'][:, :, None] > 0
    ef = relerr(o32.posteriors(batch['x'], batch['mask'], c) * valid, o64.posteriors(batch['x'], batch['mask'], c) * valid)
    for o in (o64, o32):
        for l in o.tbptt_layers():
            l.reset(batch['x'].shape[0])
    g64 = o64.loss_and_grads(batch['x'], batch['mask'], batch['y'], batch['mask'], c, context=context)[3]
    g32 = o32.loss_and_grads(batch['x'], batch['mask'], batch['y'], batch['mask'], c, context=context)[3]
    gmax = max(np.max(np.abs(g)) for g in g64)
    eg = max(float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-3 * gmax)) for a, b in zip(g32, g64))
    return ef, eg


def run_case(net, batch, cond_key=None, context=0, tol_f=2e-5, tol_g=1e-4, gscale=1.0, calibrate=False):
    dev = net.device
    if calibrate:      # tolerance = the stated bound, or 3x the fp32 noise floor of this very network
        nf, ng = fp32_noise(net.describe(), net.get_all_param_values(), batch, cond_key, context)
        tol_f = max(tol_f, 3.0 * nf); tol_g = max(tol_g, 3.0 * ng)
    x = torch.from_numpy(batch['x']).to(dev); m = torch.from_numpy(batch['mask']).to(dev)
    y = torch.from_numpy(batch['y']).to(dev)
    cond = torch.from_numpy(batch[cond_key]).to(dev) if cond_key else None
    vals = net.get_all_param_values()
    onet = O.OracleNetwork(net.describe(), vals)
    for l in onet.tbptt_layers():
        l.reset(x.shape[0])
    # forward posteriors
    post = net.posteriors(x, m, cond).cpu().numpy()
    net.check_status()
    for l in net.tbptt_layers():
        l.reset(x.shape[0])
    opost = onet.posteriors(batch['x'], batch['mask'], batch.get(cond_key))
    for l in onet.tbptt_layers():
        l.reset(x.shape[0])
    valid = batch['mask'][:, :, None] > 0
    e = relerr(post * valid, opost * valid)
    assert e < tol_f, ('posteriors', e)
    # loss + grads
    ce, ok, pred = net.loss(x, m, y, m, cond, train=True, context=context)
    net.backward()
    net.check_status()
    oc, os_, opred, ograds = onet.loss_and_grads(batch['x'], batch['mask'], batch['y'], batch['mask'],
                                                 batch.get(cond_key), context=context)
    assert abs(ce.item() - os_) / abs(os_) < tol_f, (ce.item(), os_)
    tm = batch['mask'] if not context else batch['mask'][:, :context]
    bad = (pred.cpu().numpy() != opred) & (tm != 0)
    if bad.any():
        # a frame whose two best classes the float64 oracle itself holds closer than twice the forward bound may fall
        # either way; any other disagreement is an error
        op = opost if not context else opost[:, :context]
        top2 = np.sort(op, axis=2)[:, :, -2:]
        assert ((top2[..., 1] - top2[..., 0])[bad] < 2.0 * tol_f * np.max(np.abs(opost))).all(), 'opred'
    grads = net.get_all_grad_values()
    names = [n for n, _ in net.param_names_shapes()]
    worst = 0.0
    gmax = max(np.max(np.abs(g)) for g in ograds)
    for n, g, og in zip(names, grads, ograds):
        assert g.shape == og.shape, (n, g.shape, og.shape)
        err = float(np.max(np.abs(g - og)) / (np.max(np.abs(og)) + 1e-3 * gmax))
        worst = max(worst, err)
        assert err < tol_g, (n, err)
    return e, worst


SMALL = dict(B=3, T=7, D=5, C=6)


@pytest.mark.parametrize('is_bidir', [False, True])
@pytest.mark.parametrize('pj,use_ln', [(0, False), (4, False), (0, True), (4, True)])
@pytest.mark.parametrize('clip', [0.0, 0.05])
def test_build_deep_lstm_small(is_bidir, pj, use_ln, clip):
    from asr_amd.builders import build_deep_lstm
    net = build_deep_lstm(None, None, SMALL['D'], 2, 8, pj, SMALL['C'], clip, is_bidir, use_ln, 0,
                          device=_dev(), seed=1)
    perturb_params(net, 3, 0.15)
    b = make_batch(np.random.default_rng(5), SMALL['B'], SMALL['T'], SMALL['D'], SMALL['C'])
    run_case(net, b)


def test_build_deep_lstm_ivector_input_medium():
    """B=20 > 16 exercises the batch chunking; N=64 exercises multi-workgroup groups."""
    from asr_amd.builders import build_deep_lstm
    net = build_deep_lstm(None, None, 11, 2, 64, 0, 13, 1.0, True, False, 4, ivector_var=True,
                          device=_dev(), seed=2)
    perturb_params(net, 4, 0.02)
    b = make_batch(np.random.default_rng(6), 20, 33, 11, 13, I=4)
    run_case(net, b, cond_key='ivec')


def test_ln_lstmp_model():
    from asr_amd.gating_hyper_nets import deep_projection_ln_lstm_model
    net = deep_projection_ln_lstm_model(None, None, 5, 6, 2, 4, 8, grad_clipping=0.05, dropout=0.0,
                                        device=_dev(), seed=3)
    perturb_params(net, 7, 0.15)
    b = make_batch(np.random.default_rng(8), 3, 7, 5, 6)
    run_case(net, b)


def test_ln_lstmp_model_medium():
    from asr_amd.gating_hyper_nets import deep_projection_ln_lstm_model
    net = deep_projection_ln_lstm_model(None, None, 23, 17, 2, 32, 64, grad_clipping=1.0, dropout=0.0,
                                        device=_dev(), seed=3)
    perturb_params(net, 7, 0.03)
    b = make_batch(np.random.default_rng(8), 16, 50, 23, 17)
    run_case(net, b, calibrate=True)      # LN stack over T=50: bound by 3x its own fp32 noise floor


def test_ivector_cond_ln_model():
    from asr_amd.gating_hyper_nets import deep_projection_ivector_ln_model
    net, cond_layers = deep_projection_ivector_ln_model(None, None, None, 5, 6, 2, 1, 3, 8,
                                                        grad_clipping=0.05, dropout=0.0, cond_dim=3,
                                                        device=_dev(), seed=4)
    assert len(cond_layers) == 1
    perturb_params(net, 9, 0.15)
    b = make_batch(np.random.default_rng(10), 3, 7, 5, 6, I=3)
    run_case(net, b, cond_key='ivec')


def test_lhuc_ivector():
    from asr_amd.builders import build_deep_lhuclstm_ivector
    net, _ = build_deep_lhuclstm_ivector('ivector', None, None, 5, 2, 8, 0, 6, 1, 5, 0, 0, 0, 3, 0.05,
                                         True, False, False, 3, True, device=_dev(), seed=5)
    perturb_params(net, 11, 0.2)
    b = make_batch(np.random.default_rng(12), 3, 7, 5, 6, I=3)
    run_case(net, b, cond_key='ivec')


def test_tbptt_windows_carry_state():
    """3 windows incl. a last window shorter than the context (libs/lhuclstm_layers.py:313-319)."""
    from asr_amd.builders import build_deep_lstm_tbptt
    ctxt, rc = 3, 1
    net, tb = build_deep_lstm_tbptt(None, None, 5, 2, 8, 0, 6, 3, ctxt, 0.05, False, False, 0,
                                    device=_dev(), seed=6)
    perturb_params(net, 13, 0.15)
    b = make_batch(np.random.default_rng(14), 3, 8, 5, 6, lengths=[8, 1, 5])
    vals = net.get_all_param_values()
    onet = O.OracleNetwork(net.describe(), vals)
    for l in tb:
        l.reset(3)
    for l in onet.tbptt_layers():
        l.reset(3)
    dev = net.device
    full = (b['x'], b['mask'], b['x'], b['mask'], b['y'], b['mask'])
    for win in O.gen_win(full, ctxt, rc):
        xw, mw, _, _, yw, tmw = win
        ce, ok, pred = net.loss(torch.from_numpy(np.ascontiguousarray(xw)).to(dev),
                                torch.from_numpy(np.ascontiguousarray(mw)).to(dev),
                                torch.from_numpy(np.ascontiguousarray(yw)).to(dev),
                                torch.from_numpy(np.ascontiguousarray(tmw)).to(dev), None, True, ctxt)
        net.backward()
        net.check_status()
        oc, os_, opred, ograds = onet.loss_and_grads(xw, mw, yw, tmw, None, context=ctxt)
        assert abs(ce.item() - os_) <= 2e-5 * max(abs(os_), 1.0), (ce.item(), os_)
        gmax = max(np.max(np.abs(g)) for g in ograds)
        for g, og in zip(net.get_all_grad_values(), ograds):
            assert np.max(np.abs(g - og)) / (np.max(np.abs(og)) + 1e-3 * gmax) < 1e-4


@pytest.mark.parametrize('use_ln', [False, True])
def test_tbptt_projected_state_survives_an_optimiser_step(use_ln):
    """exp_deep_lstm2_tbptt.py takes one optimiser step per window, and TBPTTLSTMLayer hands the PROJECTED hidden
    state hid_out[:, last] to the next window (libs/lhuclstm_layers.py:313-319,230-235): the state was projected with
    the W_p from BEFORE the step.  Three windows of a stateful LSTMP (with and without layer norm), an Adam step with
    a large learning rate after each (the oracle then takes the updated weights): every window's loss and gradients against the oracle.  (Carrying
    r and re-projecting it with the updated W_p -- the round-2 form -- differs by r0.dW_p in every window after the
    first: with lr = 0.05 that is far above the tolerance.)"""
    from asr_amd.builders import build_deep_lstm_tbptt
    ctxt = 4
    net, tb = build_deep_lstm_tbptt(None, None, 5, 2, 8, 4, 6, 3, ctxt, 0.0, False, use_ln, 0, device=_dev(), seed=6)
    perturb_params(net, 13, 0.15)
    b = make_batch(np.random.default_rng(15), 3, 12, 5, 6, lengths=[12, 3, 9])
    vals = [v.astype(np.float64) for v in net.get_all_param_values()]
    onet = O.OracleNetwork(net.describe(), vals)
    for l in tb:
        l.reset(3)
    for l in onet.tbptt_layers():
        l.reset(3)
    dev = net.device
    t_ = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    full = (b['x'], b['mask'], b['x'], b['mask'], b['y'], b['mask'])
    for w, win in enumerate(O.gen_win(full, ctxt, 0)):
        xw, mw, _, _, yw, tmw = win
        ce, ok, pred = net.loss(t_(xw), t_(mw), t_(yw), t_(tmw), None, True, ctxt)
        net.backward()
        net.check_status()
        oc, os_, opred, ograds = onet.loss_and_grads(xw, mw, yw, tmw, None, context=ctxt)
        assert abs(ce.item() - os_) <= 2e-5 * max(abs(os_), 1.0), (w, ce.item(), os_)
        gmax = max(np.max(np.abs(g)) for g in ograds)
        for n, g, og in zip([n for n, _ in net.param_names_shapes()], net.get_all_grad_values(), ograds):
            assert np.max(np.abs(g - og)) / (np.max(np.abs(og)) + 1e-3 * gmax) < 1e-4, (w, n)
        net.adam_update(0.05)
        # the oracle takes the weights the HIP path now holds (its own Adam step would land within rounding of them,
        # but sign-like first Adam steps amplify 1e-7 gradient differences): what is under test is the carried state
        for v, nv in zip(onet.values, net.get_all_param_values()):
            v[...] = nv
    assert w == 2


@pytest.mark.parametrize('is_bidir', [False, True])
@pytest.mark.parametrize('clip', [0.0, 0.05])
def test_hyperlstm_small(is_bidir, clip):
    from asr_amd.builders import build_deep_hyperlstm
    net = build_deep_hyperlstm('hyper', None, None, 5, 2, 8, 4, 2, 6, clip, is_bidir, 0,
                               device=_dev(), seed=7)
    perturb_params(net, 15, 0.15)
    b = make_batch(np.random.default_rng(16), 3, 7, 5, 6)
    run_case(net, b)


@pytest.mark.parametrize('chunks,is_bidir,lhuc', [(4, True, False), (3, True, False), (5, False, False), (4, True, True)])
def test_hyper_time_chunked_launches_match_one_launch(chunks, is_bidir, lhuc):
    """A layer run as consecutive time chunks (asr_hyper_scan_desc.step0/step1; state continued from the saved
    rows forward, from `carry` backward; the next layer's projection, the output layer and dX pipelined piecewise
    on the side stream) gives what the single launch gives: recurrence outputs bit for bit, everything that
    passes through the re-associated GEMM sums to fp32 rounding; and both match the oracle.  B = 18: two batch
    chunks; ragged lengths, one row of length 1."""
    from asr_amd.builders import build_deep_hyperlstm, build_deep_hyperlhuclstm
    def build():
        if lhuc:
            net = build_deep_hyperlhuclstm('hyperlhuc', None, None, 11, 2, 32, 8, 4, 13, 1.0, is_bidir, 0,
                                           device=_dev(), seed=8)
        else:
            net = build_deep_hyperlstm('hyper', None, None, 11, 2, 32, 8, 4, 13, 1.0, is_bidir, 0, device=_dev(), seed=8)
        perturb_params(net, 17, 0.03)
        return net
    b = make_batch(np.random.default_rng(19), 18, 300, 11, 13)
    x, m, y = (torch.from_numpy(b[k]).cuda() for k in ('x', 'mask', 'y'))
    res = []
    for n in (1, chunks):
        net = build()
        net.scan_chunks = n
        ce, _, pred = net.loss(x, m, y, m)
        net.backward(); net.check_status()
        res.append((ce.item(), pred.cpu().numpy(), [p.ctx['Y'].clone() for p in net.pairs], net.store.g.clone(), net))
    (ce1, pr1, y1, g1, _), (ce2, pr2, y2, g2, net) = res
    assert torch.equal(y1[0], y2[0])                       # first layer: same projections, same recurrence
    assert (y1[1] - y2[1]).abs().max().item() < 1e-5 and abs(ce1 - ce2) < 1e-5 * abs(ce1)
    assert ((g1 - g2).abs().max() / g1.abs().max()).item() < 2e-5
    onet = O.OracleNetwork(net.describe(), net.get_all_param_values())
    _, os_, _, og = onet.loss_and_grads(b['x'], b['mask'], b['y'], b['mask'])
    assert abs(ce2 - os_) < 2e-5 * abs(os_)
    gmax = max(np.abs(g).max() for g in og)
    for (nm, _), g, o in zip(net.param_names_shapes(), net.get_all_grad_values(), og):
        assert np.abs(g - o).max() <= 1e-4 * (np.abs(o).max() + 1e-3 * gmax), nm


def test_hyperlstm_medium():
    """N=64, H=16, Z=8: 16 workgroups per direction, 1 hyper unit per workgroup, B=18 > 16."""
    from asr_amd.builders import build_deep_hyperlstm
    net = build_deep_hyperlstm('hyper', None, None, 11, 2, 64, 16, 8, 13, 1.0, True, 0,
                               device=_dev(), seed=8)
    perturb_params(net, 17, 0.03)
    b = make_batch(np.random.default_rng(18), 18, 31, 11, 13)
    run_case(net, b)


def test_lstm_width_that_needs_one_launch_per_direction_and_the_width_limit():
    """516 units bidirectional: 129 workgroups per direction, 258 do not fit one launch and 516 is not a multiple of 8 (no
    two-tile workgroups): Network.scan_lstm runs the directions one after the other.  4608 units: more than 256 workgroups
    of 16 units; the error says so."""
    from asr_amd.builders import build_deep_lstm
    from asr_amd._lib import AsrError
    net = build_deep_lstm(None, None, 7, 2, 516, 0, 9, 1.0, True, True, 0, device=_dev(), seed=91)
    perturb_params(net, 92, 0.01)
    b = make_batch(np.random.default_rng(93), 3, 6, 7, 9)
    run_case(net, b, calibrate=True)
    net = build_deep_lstm(None, None, 7, 1, 4608, 0, 9, 1.0, False, False, 0, device=_dev(), seed=94)
    x = torch.from_numpy(b['x']).to(net.device); m = torch.from_numpy(b['mask']).to(net.device)
    with pytest.raises(AsrError, match='more than the persistent scan holds'):
        net.posteriors(x, m, None)


@pytest.mark.parametrize('kind,N,bidir', [('lstm', 1536, True), ('lstm', 2048, False), ('lstmp', 2048, True), ('ln', 1536, False),
                                          ('lstm', 4096, False)])
def test_lstm_wider_than_resident_weights_streams_them(kind, N, bidir):
    """Layers whose recurrent weights do not fit a CU's LDS next to the staging buffers (16 UT N 4 bytes per workgroup:
    above ~1280 units) run the generic scans with the weights streamed from L2 / Infinity Cache every step
    (lstm_scan_{fwd,bwd}_kernel<UT, 1, false, true>; the reference has no width limit).  Plain, projected and
    layer-normalised layers, one launch per direction, against the oracle."""
    from asr_amd.builders import build_deep_lstm
    pj = 256 if kind == 'lstmp' else 0
    net = build_deep_lstm(None, None, 7, 1, N, pj, 9, 1.0, bidir, kind == 'ln', 0, device=_dev(), seed=95)
    perturb_params(net, 96, 0.005)
    b = make_batch(np.random.default_rng(97), 3, 5, 7, 9)
    run_case(net, b, calibrate=True)


def test_hyperlstm_wider_than_one_launch_holds():
    """N = 1024 bidirectional: 2 x 256 workgroups do not fit one persistent launch, the directions run as one launch each
    (Network.scan_hyper).  B = 18 (two batch chunks), ragged lengths; and the scaling layer on the same scan."""
    from asr_amd.builders import build_deep_hyperlstm
    from asr_amd.gating_hyper_nets import deep_scaling_hyper_model
    net = build_deep_hyperlstm('hyper', None, None, 11, 2, 1024, 32, 8, 13, 1.0, True, 0, device=_dev(), seed=81)
    perturb_params(net, 82, 0.01)
    b = make_batch(np.random.default_rng(83), 18, 9, 11, 13)
    run_case(net, b)
    net = deep_scaling_hyper_model(None, None, 11, 13, [16], [1024], grad_clipping=1.0, device=_dev(), seed=84)
    perturb_params(net, 85, 0.01)
    run_case(net, b)


@pytest.mark.parametrize('clip', [0.0, 0.05])
@pytest.mark.parametrize('H', [4, 6])
def test_scaling_hyper_small(clip, H):
    """ScalingHyperLSTMLayer stack (hyper_layers.py:525-1205); H=6 exercises the zero padding of the
    inner width to a multiple of 4."""
    from asr_amd.gating_hyper_nets import deep_scaling_hyper_model
    net = deep_scaling_hyper_model(None, None, 5, 6, [H, H], [8, 8], grad_clipping=clip, device=_dev(), seed=21)
    perturb_params(net, 22, 0.15)
    b = make_batch(np.random.default_rng(23), 3, 7, 5, 6)
    run_case(net, b)
    # padded inner units stay exactly zero in parameters and gradients
    for pair in net.pairs:
        for dp in pair.dirs:
            assert float(dp.p('Whh')[H:].abs().max()) == 0.0 if pair.Hp > H else True
            assert float(dp.g('M')[H:pair.Hp].abs().max()) == 0.0 if pair.Hp > H else True


def test_scaling_hyper_medium_inner_hid():
    from asr_amd.gating_hyper_nets import deep_scaling_hyper_model
    out = deep_scaling_hyper_model(None, None, 11, 13, [16, 14], [64, 32], grad_clipping=1.0,
                                   get_inner_hid=True, use_softmax=False, device=_dev(), seed=24)
    net = out[-1]; inner = out[:-1]
    assert len(inner) == 4
    perturb_params(net, 25, 0.03)
    b = make_batch(np.random.default_rng(26), 18, 31, 11, 13)
    run_case(net, b)
    onet = O.OracleNetwork(net.describe(), net.get_all_param_values())
    onet.posteriors(b['x'], b['mask'], None)
    x = torch.from_numpy(b['x']).to(net.device); m = torch.from_numpy(b['mask']).to(net.device)
    net.posteriors(x, m, None)
    for k, h in enumerate(inner):
        got = h.get_output().cpu().numpy().reshape(31, 18, -1).transpose(1, 0, 2)
        want = onet.layers[k // 2][k % 2].inner_hid
        assert relerr(got, want) < 2e-5, k


def test_scaling_hyper_reference_widths():
    """inner 250 / outer 500 (scripts/wsj/exp_scale_hypernet_skip.py): 125 workgroups per direction,
    inner width padded 250 -> 252."""
    from asr_amd.gating_hyper_nets import deep_scaling_hyper_model
    net = deep_scaling_hyper_model(None, None, 123, 50, [250], [500], grad_clipping=1.0, device=_dev(), seed=27)
    perturb_params(net, 28, 0.01)
    b = make_batch(np.random.default_rng(29), 4, 24, 123, 50)
    run_case(net, b, calibrate=True)


@pytest.fixture(params=['persistent', 'step_loop'])
def scal_form(request):
    """Both forms of asr_scaling_*: the persistent launch (the library's choice where the shape allows) and the step loop."""
    from asr_amd import _lib
    old = _lib.lib().asr_scaling_set_mode(1 if request.param == 'step_loop' else 0)
    yield request.param
    _lib.lib().asr_scaling_set_mode(old)


@pytest.mark.parametrize('peep,ln,li', [(True, False, False), (False, True, False), (False, False, True), (True, True, True)])
@pytest.mark.parametrize('clip', [0.0, 0.05])
def test_scaling_hyper_options_small(peep, ln, li, clip, scal_form):
    """use_peepholes / use_layer_norm / learn_init of ScalingHyperLSTMLayer (hyper_layers.py:977-1091,:1127-1131) on
    both forms of step_scaling.hip's path: two stacked bidirectional layers, ragged lengths, clip biting."""
    from asr_amd.gating_hyper_nets import deep_scaling_hyper_model
    net = deep_scaling_hyper_model(None, None, 5, 6, [8, 4], [12, 8], use_peepholes=peep, use_layer_norm=ln,
                                   learn_init=li, grad_clipping=clip, device=_dev(), seed=51)
    perturb_params(net, 52, 0.15)
    b = make_batch(np.random.default_rng(53), 3, 7, 5, 6)
    run_case(net, b)


@pytest.mark.parametrize('peep,ln', [(False, False), (True, False), (False, True), (True, True)])
@pytest.mark.parametrize('clip', [0.0, 0.05])
def test_equal_scale_hyper_small(peep, ln, clip, scal_form):
    """EqualScaleHyperLSTMLayer (hyper_layers.py:1207-1890): the scaling layer's step with an inner cell as wide as the outer
    one and its hidden state projected to the inner units; two stacked bidirectional layers, ragged lengths.  The kernels
    run the projection folded into its consumers (engine_scaling.EqualScaleStepPair), the oracle runs it as written."""
    from asr_amd.gating_hyper_nets import deep_equal_scale_hyper_model
    net = deep_equal_scale_hyper_model(None, None, 5, 6, [4, 8], [12, 8], use_peepholes=peep, use_layer_norm=ln,
                                       grad_clipping=clip, device=_dev(), seed=71)
    perturb_params(net, 72, 0.15)
    b = make_batch(np.random.default_rng(73), 3, 7, 5, 6)
    run_case(net, b)


def test_equal_scale_hyper_medium_inner_hid_and_limits(scal_form):
    """64 outer / 16 inner units, 18 rows (two batch chunks), T = 29, every option; the projected inner hidden sequences of
    get_inner_hid=True against the oracle's (with learn_init: the reverse direction's padding holds inner_hid_init)."""
    from asr_amd.gating_hyper_nets import deep_equal_scale_hyper_model
    out = deep_equal_scale_hyper_model(None, None, 11, 13, [16], [64], use_peepholes=True, use_layer_norm=True,
                                       learn_init=True, grad_clipping=1.0, get_inner_hid=True, device=_dev(), seed=74)
    net = out[-1]; inner = out[:-1]
    perturb_params(net, 75, 0.05)
    b = make_batch(np.random.default_rng(76), 18, 29, 11, 13)
    run_case(net, b, calibrate=True)
    onet = O.OracleNetwork(net.describe(), net.get_all_param_values())
    onet.posteriors(b['x'], b['mask'], None)
    x = torch.from_numpy(b['x']).to(net.device); m = torch.from_numpy(b['mask']).to(net.device)
    net.posteriors(x, m, None)
    for k, h in enumerate(inner):
        got = h.get_output().cpu().numpy().reshape(29, 18, -1).transpose(1, 0, 2)
        assert got.shape[-1] == 16
        assert relerr(got, onet.layers[0][k].inner_hid) < 2e-5, k


@pytest.mark.parametrize('peep,ln', [(False, False), (True, True)])
@pytest.mark.parametrize('clip', [0.0, 0.05])
def test_equal_scale_hyper_learn_init(peep, ln, clip, scal_form):
    """learn_init=True (hyper_layers.py:1782-1788): inner_hid_init is a row of the projected space; two stacked
    bidirectional layers, ragged lengths (the reverse direction starts on padding)."""
    from asr_amd.gating_hyper_nets import deep_equal_scale_hyper_model
    net = deep_equal_scale_hyper_model(None, None, 5, 6, [4, 8], [12, 8], use_peepholes=peep, use_layer_norm=ln,
                                       learn_init=True, grad_clipping=clip, device=_dev(), seed=77)
    perturb_params(net, 78, 0.15)
    b = make_batch(np.random.default_rng(79), 5, 9, 5, 6, lengths=[9, 1, 4, 7, 9])
    run_case(net, b)


def test_step_loop_layers_more_rows_than_one_call():
    """B = 67: the step-loop layers take 64 rows per C-ABI call (four 16-row chunks in every launch), the caller loops
    over calls: one call of 64 rows and one of 3."""
    from asr_amd.gating_hyper_nets import deep_scaling_hyper_model, deep_projection_hyper_model
    b = make_batch(np.random.default_rng(57), 67, 3, 5, 6)
    net = deep_scaling_hyper_model(None, None, 5, 6, [4], [8], use_peepholes=True, use_layer_norm=True, learn_init=True,
                                   grad_clipping=0.05, device=_dev(), seed=58)
    perturb_params(net, 59, 0.15)
    run_case(net, b)
    net = deep_projection_hyper_model(None, None, 5, 6, 2, 4, 8, grad_clipping=0.05, device=_dev(), seed=60)[0]
    perturb_params(net, 61, 0.15)
    run_case(net, b)


def test_scaling_hyper_options_timit_script_shape(scal_form):
    """The configuration scripts/timit/exp_scalehyper.py runs (:26-41,:331-338): one bidirectional layer, 64 inner /
    128 outer units, layer norm and learned initial rows, 123 inputs, 63 classes; 18 rows = two batch chunks; the
    inner hidden sequences of get_inner_hid=True against the oracle's."""
    from asr_amd.gating_hyper_nets import deep_scaling_hyper_model
    out = deep_scaling_hyper_model(None, None, 123, 63, [64], [128], use_layer_norm=True, learn_init=True,
                                   grad_clipping=1.0, get_inner_hid=True, device=_dev(), seed=54)
    net = out[-1]; inner = out[:-1]
    perturb_params(net, 55, 0.05)
    b = make_batch(np.random.default_rng(56), 18, 29, 123, 63)
    run_case(net, b)
    onet = O.OracleNetwork(net.describe(), net.get_all_param_values())
    onet.posteriors(b['x'], b['mask'], None)
    x = torch.from_numpy(b['x']).to(net.device); m = torch.from_numpy(b['mask']).to(net.device)
    net.posteriors(x, m, None)
    for k, h in enumerate(inner):
        got = h.get_output().cpu().numpy().reshape(29, 18, -1).transpose(1, 0, 2)
        assert relerr(got, onet.layers[0][k].inner_hid) < 2e-5, k


@pytest.mark.parametrize('is_bidir,clip,reparam', [(True, 0.05, 'relu'), (False, 0.0, 'relu'), (True, 0.0, 'exp'),
                                                    (True, 0.05, '2sigmoid'), (False, 0.0, 'identity')])
def test_hyperlhuc_small(is_bidir, clip, reparam):
    """HyperLHUCLSTMLayer (hyper_lhuc_layers.py:509-678) on the hyper scan's lhuc mode."""
    from asr_amd.builders import build_deep_hyperlhuclstm
    net = build_deep_hyperlhuclstm('hyperlhuc', None, None, 5, 2, 8, 4, 2, 6, clip, is_bidir, 0, reparam=reparam,
                                   device=_dev(), seed=41)
    perturb_params(net, 42, 0.15)
    b = make_batch(np.random.default_rng(43), 3, 7, 5, 6)
    run_case(net, b)


def test_hyperlhuc_medium():
    from asr_amd.builders import build_deep_hyperlhuclstm
    net = build_deep_hyperlhuclstm('hyperlhuc', None, None, 11, 2, 64, 16, 8, 13, 1.0, True, 0, device=_dev(), seed=44)
    perturb_params(net, 45, 0.03)
    b = make_batch(np.random.default_rng(46), 18, 31, 11, 13)
    run_case(net, b)


@pytest.fixture(params=['persistent', 'step_loop'])
def prj_form(request):
    """Both forms of asr_prjhyper_*: the persistent launch (the library's choice where the weights fit the chip) and the
    step loop (wider layers, or asr_prjhyper_set_mode(1))."""
    from asr_amd import _lib
    old = _lib.lib().asr_prjhyper_set_mode(1 if request.param == 'step_loop' else 0)
    yield request.param
    _lib.lib().asr_prjhyper_set_mode(old)


@pytest.mark.parametrize('clip', [0.0, 0.05])
def test_projection_hyper_small(clip, prj_form):
    """ProjectionHyperLSTMLayer (hyper_layers.py:14-523) on both forms of its HIP path + ProjectLSTMLayer on top."""
    from asr_amd.gating_hyper_nets import deep_projection_hyper_model
    out = deep_projection_hyper_model(None, None, 5, 6, 2, 4, 8, grad_clipping=clip, device=_dev(), seed=61)
    net, inner = out[0], out[1:]
    assert len(inner) == 2
    perturb_params(net, 62, 0.15)
    b = make_batch(np.random.default_rng(63), 3, 7, 5, 6)
    run_case(net, b)
    onet = O.OracleNetwork(net.describe(), net.get_all_param_values())
    onet.posteriors(b['x'], b['mask'], None)
    for d, h in enumerate(inner):
        got = h.get_output().cpu().numpy().reshape(7, 3, -1).transpose(1, 0, 2)
        assert relerr(got, onet.layers[0][d].inner_hid) < 2e-5


@pytest.mark.parametrize('kind', ['ln', 'icond', 'gate_ln', 'sample_gate_ln'])
def test_ln_backward_2d_decomposition_n128(kind):
    """N = 128 (32 workgroups per direction, one source tile per wave): the row-LN backward runs as
    lstm_scan_bwd_ln2d_kernel (asr_lstm_bwd_pack_mode == 2); B = 18 adds a second, ragged batch chunk."""
    from asr_amd import _lib, gating_hyper_nets as G
    assert _lib.lib().asr_lstm_bwd_pack_mode(128, 2, 1) == 2 and _lib.lib().asr_lstm_bwd_pack_mode(8, 2, 1) == 1
    cond_key = None
    if kind == 'ln':
        net = G.deep_projection_ln_lstm_model(None, None, 11, 13, 2, None, 128, 1.0, 0.0, device=_dev(), seed=81)
    elif kind == 'gate_ln':                 # per-gate statistics (LayerNormProjectLSTM_v0_2_Layer)
        net = G.deep_prj_lstm_ln_model_v2(None, None, 11, 13, 2, 128, 64, grad_clipping=1.0, dropout=0.0, device=_dev(), seed=85)
    elif kind == 'sample_gate_ln':          # CondLayerNormProjectLSTM_v0_2_Layer
        out = G.deep_prj_lstm_dln_model_v2(None, None, 11, 13, 128, 64, 16, 2, num_conds=1, grad_clipping=1.0, dropout=0.0,
                                           device=_dev(), seed=86)
        net = out[0] if isinstance(out, (tuple, list)) else out
    else:
        net, _ = G.deep_projection_ivector_ln_model(None, None, None, 11, 13, 2, 1, 16, 128, 1.0, 0.0, cond_dim=7,
                                                    device=_dev(), seed=82)
        cond_key = 'ivec'
    perturb_params(net, 83, 0.03)
    b = make_batch(np.random.default_rng(84), 18, 23, 11, 13, I=7 if cond_key else 0)
    run_case(net, b, cond_key=cond_key, calibrate=True)


def test_dropout_training_matches_oracle_with_the_same_masks():
    """models/gating_hyper_nets.py:276-307 at the reference default dropout=0.2: training (deterministic=False)
    drops the input of every layer and of the output layer; the oracle is fed the masks the HIP path used
    (regenerated from the Philox stream ids) and every gradient agrees at 1e-4; predict / ff ignore dropout.
    (64-unit layers: on 8-unit toy nets a dropped 5-dim input leaves near-degenerate layer-norm rows whose
    fp32 noise alone exceeds 1e-4.)"""
    from asr_amd import gating_hyper_nets as G
    D, C = 23, 17
    for which in ('ln', 'icond'):
        if which == 'ln':
            net = G.deep_projection_ln_lstm_model(None, None, D, C, 2, 32, 64, grad_clipping=1.0, device=_dev(), seed=91)
            ck = None
        else:
            net, _ = G.deep_projection_ivector_ln_model(None, None, None, D, C, 2, 1, 8, 64, 1.0, cond_dim=7,
                                                        device=_dev(), seed=92)
            ck = 'ivec'
        assert net.dropout == 0.2                                  # the reference's default argument
        vals = perturb_params(net, 93, 0.03)
        b = make_batch(np.random.default_rng(94), 6, 12, D, C, I=7 if ck else 0)
        x, m, y = (torch.from_numpy(b[k]).cuda() for k in ('x', 'mask', 'y'))
        c = torch.from_numpy(b[ck]).cuda() if ck else None
        onet = O.OracleNetwork(net.describe(), vals)
        B, T = b['mask'].shape
        for rep in range(3):                                       # a fresh mask every step
            ce, _, _ = net.loss(x, m, y, m, c, train=True)
            net.backward(); net.check_status()
            masks = [mk.cpu().numpy().reshape(T, B, -1).transpose(1, 0, 2) for mk in net.dropout_masks()]
            assert len(masks) == 3 and all(0.7 < (mk > 0).mean() < 0.9 for mk in masks)
            if rep:
                assert not np.array_equal(masks[1], first[1])
            first = masks
            for l, mk in enumerate(masks):                         # = the oracle's own Philox restatement
                sid = (net.dropout_step << 20) | l
                want = O.dropout_mask(T * B, mk.shape[2], 0.2, net.dropout_seed, sid)
                assert np.array_equal(mk, want.reshape(T, B, -1).transpose(1, 0, 2))
            _, os_, _, og = onet.loss_and_grads(b['x'], b['mask'], b['y'], b['mask'], b.get(ck), drop=masks)
            assert abs(ce.item() - os_) < 2e-5 * abs(os_)
            gmax = max(np.abs(g).max() for g in og)
            for (n, _), g, o in zip(net.param_names_shapes(), net.get_all_grad_values(), og):
                assert np.abs(g - o).max() <= 1e-4 * (np.abs(o).max() + 1e-3 * gmax), (which, rep, n)
        post = net.posteriors(x, m, c).cpu().numpy()               # deterministic=True: identity
        valid = b['mask'][:, :, None] > 0
        assert relerr(post * valid, onet.posteriors(b['x'], b['mask'], b.get(ck)) * valid) < 2e-5
        ce_eval, _, _ = net.loss(x, m, y, m, c, train=False)
        _, os_det, _, _ = onet.loss_and_grads(b['x'], b['mask'], b['y'], b['mask'], b.get(ck))
        assert abs(ce_eval.item() - os_det) < 2e-5 * abs(os_det)


def test_projection_hyper_aux_costs_and_trainer():
    """scripts/hypernets/exp_prj_hyper.py:70-146: the inter / intra variance costs on the inner hidden sequences
    and their gradient through the whole net, against the oracle (weights 0.7 / 1.3); then the trainer's outputs."""
    from asr_amd.gating_hyper_nets import deep_projection_hyper_model
    from asr_amd import script_graphs, updates
    net, fi, bi = deep_projection_hyper_model(None, None, 5, 6, 2, 4, 8, grad_clipping=1.0, device=_dev(), seed=71)
    perturb_params(net, 72, 0.15)
    b = make_batch(np.random.default_rng(73), 4, 7, 5, 6)
    dev = net.device
    x, m, y = (torch.from_numpy(b[k]).to(dev) for k in ('x', 'mask', 'y'))
    onet = O.OracleNetwork(net.describe(), net.get_all_param_values())
    oc, os_, opred, ograds = onet.loss_and_grads(b['x'], b['mask'], b['y'], b['mask'], aux=(0.7, 1.3))
    net.loss(x, m, y, m, None, train=True)
    inter, intra = fi.pair.aux_costs(0.7, 1.3).tolist()        # a device tensor: nothing of the algebra runs on the host
    net.backward()
    net.check_status()
    assert abs(inter - onet.aux_costs[0]) < 1e-5 * max(1.0, abs(onet.aux_costs[0]))
    assert abs(intra - onet.aux_costs[1]) < 1e-5 * max(1.0, abs(onet.aux_costs[1]))
    for (n, _), g, og in zip(net.param_names_shapes(), net.get_all_grad_values(), ograds):
        assert relerr(g, og) < 1e-4, n
    # without weights nothing is injected: gradients equal the plain CE ones
    _, _, _, og0 = onet.loss_and_grads(b['x'], b['mask'], b['y'], b['mask'])
    assert max(relerr(a_, b_) for a_, b_ in zip(ograds, og0)) > 1e-3          # the costs do matter in this case
    tf, up = script_graphs.build_prj_hyper_trainer(None, None, None, None, net.get_all_params(trainable=True), net,
                                                   fi, bi, updates.momentum, 0.7, 1.3, 1e-3)
    out = tf(b['x'], b['mask'], b['y'], b['mask'])
    assert len(out) == 5 and all(np.isfinite(v) for v in out)
    assert abs(out[3] - onet.aux_costs[0]) < 1e-4 and abs(out[4] - onet.aux_costs[1]) < 1e-4
    assert abs(out[0] - os_ / b['mask'].sum()) < 1e-4 * max(1.0, os_ / b['mask'].sum())


def test_projection_hyper_medium(prj_form):
    """B = 18 (two batch chunks), N = 64, F = 32."""
    from asr_amd.gating_hyper_nets import deep_projection_hyper_model
    net = deep_projection_hyper_model(None, None, 11, 13, 2, 32, 64, grad_clipping=1.0, device=_dev(), seed=64)[0]
    perturb_params(net, 65, 0.03)
    b = make_batch(np.random.default_rng(66), 18, 21, 11, 13)
    run_case(net, b, calibrate=True)


@pytest.mark.parametrize('shape', [(3, 7, 8, 4), (16, 33, 64, 32), (5, 19, 132, 20), (40, 9, 36, 12), (16, 12, 512, 256)])
def test_projection_hyper_persistent_launch_agrees_with_the_step_loop(shape):
    """The persistent launch (weights resident in registers and LDS, phases separated by in-kernel exchanges) against the
    step loop on the same network and batch: posteriors and every gradient to rounding (the two split the reduction axes
    differently, so not bit for bit), twice in a row (the launch's sync block and partial buffers are reused), and the
    persistent form reproduces itself bit for bit.  The last shape is the reference script's width."""
    from asr_amd import _lib
    from asr_amd.gating_hyper_nets import deep_projection_hyper_model
    B, T, N, F = shape
    L = _lib.lib()
    net = deep_projection_hyper_model(None, None, 9, 11, 2, F, N, grad_clipping=1.0, device=_dev(), seed=71)[0]
    perturb_params(net, 72, 0.03)
    b = make_batch(np.random.default_rng(73), B, T, 9, 11)
    x, m, y = (torch.from_numpy(np.ascontiguousarray(b[k])).to(_dev()) for k in ('x', 'mask', 'y'))

    def run():
        ce = net.loss(x, m, y, m, None, train=True)[0]
        net.backward(); net.join_deferred(); net.check_status()
        return float(ce), net.get_all_grad_values()

    old = L.asr_prjhyper_set_mode(1)
    try:
        l_loop, g_loop = run()
        L.asr_prjhyper_set_mode(0)
        l_per, g_per = run()
        l_per2, g_per2 = run()
    finally:
        L.asr_prjhyper_set_mode(old)
    assert l_per == l_per2 and all(np.array_equal(a, c) for a, c in zip(g_per, g_per2))
    assert abs(l_per - l_loop) <= 2e-5 * max(1.0, abs(l_loop))
    num = sum(float(((a.astype(np.float64) - c) ** 2).sum()) for a, c in zip(g_per, g_loop))
    den = sum(float((c.astype(np.float64) ** 2).sum()) for c in g_loop)
    assert (num / den) ** 0.5 < 2e-5, (num / den) ** 0.5


def test_adam_step_matches_oracle():
    from asr_amd.builders import build_deep_lstm
    net = build_deep_lstm(None, None, 5, 1, 8, 0, 6, 0.0, False, False, 0, device=_dev(), seed=1)
    b = make_batch(np.random.default_rng(5), 3, 7, 5, 6)
    dev = net.device
    x = torch.from_numpy(b['x']).to(dev); m = torch.from_numpy(b['mask']).to(dev)
    y = torch.from_numpy(b['y']).to(dev)
    vals = [v.astype(np.float64) for v in net.get_all_param_values()]
    mo = [np.zeros_like(v) for v in vals]; vo = [np.zeros_like(v) for v in vals]
    t = 0
    for _ in range(3):
        net.loss(x, m, y, m); net.backward()
        g = [v.astype(np.float64) for v in net.get_all_grad_values()]
        gn = net.adam_update(1e-3).item()
        t = O.adam_step(vals, g, mo, vo, t)
        np.testing.assert_allclose(gn, np.sqrt(sum((gi ** 2).sum() for gi in g)), rtol=1e-5)
    for a, o in zip(net.get_all_param_values(), vals):
        assert relerr(a, o) < 1e-5


@pytest.mark.parametrize('use_ivector_input', [False, True])
def test_lhuc_seqsum(use_ivector_input):
    """build_deep_lhuclstm_seqsum: shared sequence-summarising embedding, bidirectional, projection."""
    from asr_amd.builders import build_deep_lhuclstm_seqsum
    net, speaker = build_deep_lhuclstm_seqsum('SeqSumLHUCLSTMLayer', None, None, 5, 2, 8, 4, 6, 1, 5, 7, 1, 4, 3,
                                              0.05, True, False, use_ivector_input, 3, True, device=_dev(), seed=5)
    assert speaker is net.seqsum and net.describe()['seqsum']['num_layers'] == 1
    names = [p.name for p in net.get_all_params()]
    assert names[0].endswith('seqsum_W_0') and names[3].endswith('seqsum_b_1')     # incomings-first order
    perturb_params(net, 11, 0.2)
    b = make_batch(np.random.default_rng(12), 3, 7, 5, 6, I=3)
    run_case(net, b, cond_key='ivec' if use_ivector_input else None)


def test_lhuc_seqsum_tbptt_windows():
    from asr_amd.builders import build_deep_uni_lhuclstm_seqsum_tbptt
    ctxt = 3
    net, speaker, tb = build_deep_uni_lhuclstm_seqsum_tbptt('SeqSumLHUCLSTMLayer', None, None, 5, 2, 8, 0, 6, 1, 5,
                                                            7, 2, 4, 3, 0.05, False, False, 0, None,
                                                            device=_dev(), seed=6)
    assert len(tb) == 2
    for l in tb:                                  # the reference's tbptt builder sets no context on the layers
        l.context = ctxt
    perturb_params(net, 13, 0.2)
    b = make_batch(np.random.default_rng(14), 3, 8, 5, 6, lengths=[8, 1, 5])
    onet = O.OracleNetwork(net.describe(), net.get_all_param_values())
    for l in tb:
        l.reset(3)
    for l in onet.tbptt_layers():
        l.reset(3)
    dev = net.device
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    full = (b['x'], b['mask'], b['x'], b['mask'], b['y'], b['mask'])
    for win in O.gen_win(full, ctxt, 0):
        xw, mw, _, _, yw, tmw = win
        ce, ok, pred = net.loss(t(xw), t(mw), t(yw), t(tmw), None, True, ctxt)
        net.backward()
        net.check_status()
        oc, os_, opred, ograds = onet.loss_and_grads(xw, mw, yw, tmw, None, context=ctxt)
        assert abs(ce.item() - os_) <= 2e-5 * max(abs(os_), 1.0), (ce.item(), os_)
        gmax = max(np.max(np.abs(g)) for g in ograds)
        for g, og in zip(net.get_all_grad_values(), ograds):
            assert np.max(np.abs(g - og)) / (np.max(np.abs(og)) + 1e-3 * gmax) < 1e-4


@pytest.mark.parametrize('kind', ['hyper', 'ln_prj', 'ivec_cond', 'lhuc_tbptt'])
def test_work_issued_ahead_on_the_side_stream_changes_no_bit(kind):
    """Network.prepare_ahead (folds, packs, LHUC scale MLP, external cond-LN prologue, TBPTT state snapshot issued on the
    side stream before the first scan; carried-state copies behind the scan) against the same net with everything
    inline on one stream (overlap off): loss and every gradient bit for bit, over two consecutive steps / three
    TBPTT windows (the second and third start from carried state)."""
    from asr_amd import builders, gating_hyper_nets as G
    dev = _dev()
    I = 0
    tb = []
    if kind == 'hyper':
        net = builders.build_deep_hyperlstm('hyper', None, None, 5, 2, 8, 4, 2, 6, 0.05, True, 0, device=dev, seed=21)
    elif kind == 'ln_prj':
        net = G.deep_projection_ln_lstm_model(None, None, 5, 6, 2, None, 8, 0.05, 0.0, device=dev, seed=22)
    elif kind == 'ivec_cond':
        net, _ = G.deep_projection_ivector_ln_model(None, None, None, 5, 6, 2, 1, 3, 8, grad_clipping=0.05, dropout=0.0,
                                                    cond_dim=3, device=dev, seed=23)
        I = 3
    else:
        net, _, tb = builders.build_deep_uni_lhuclstm_ivector_tbptt(
            'IVectorLHUCLSTMLayer', None, None, 5, 2, 8, 0, 6, 1, 5, 0, 0, 0, 3, 0.05, False, False, 3, True,
            device=dev, seed=24)
        I = 3
    perturb_params(net, 25, 0.15)
    vals = net.get_all_param_values()
    b = make_batch(np.random.default_rng(26), 3, 9, 5, 6, lengths=[9, 1, 6], I=I)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    x, m, y = t(b['x']), t(b['mask']), t(b['y'])
    iv = t(b['ivec']) if I else None

    def run(overlap):
        net.overlap = overlap
        net.set_all_param_values(vals)
        net.t_prev = 0; net.m.zero_(); net.v.zero_()
        out = []
        if tb:
            for l in tb:
                l.reset(3)
            for w in range(3):
                sl = slice(3 * w, 3 * w + 3)
                ce, _, _ = net.loss(x[:, sl].contiguous(), m[:, sl].contiguous(), y[:, sl].contiguous(),
                                    m[:, sl].contiguous(), iv[:, sl].contiguous(), True, 3)
                net.backward()
                out.append((ce.item(), net.store.g.clone()))
                net.adam_update(1e-2)
        else:
            for _ in range(2):
                ce, _, _ = net.loss(x, m, y, m, iv)
                net.backward()
                out.append((ce.item(), net.store.g.clone()))
                net.adam_update(1e-2)
        net.check_status()
        return out

    a, c = run(True), run(False)
    for (ce1, g1), (ce0, g0) in zip(a, c):
        assert ce1 == ce0
        assert torch.equal(g1, g0)


def test_shape_fuzz_fixed_seed():
    """60 cases of tools/fuzz_parity.py (random layer family, widths, batch, ragged lengths, clip, direction) against the
    oracle: none may fail; refusals must name a limit (the fuzzer counts only those as refused).  The tool's header says
    what a 'noisy' case is; DESIGN.md section 8 has the 3550-case record."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    import fuzz_parity
    ok, noisy, refused, failed = fuzz_parity.run(60, 5)
    assert failed == 0
    assert ok >= 45
