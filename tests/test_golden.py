"""Golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py -- see its header: parity
unpinned, they freeze the repo's float64 restatement).
  CPU: the oracle reproduces every stored output from the stored inputs and weights.
  GPU: the HIP path, driven through the reference-shaped API (builders -> trainer / predictor / ff),
       matches the stored outputs: forward quantities <= 2e-5, gradients <= 1e-4 relative to the
       tensor's max (fp32 over T dependent steps against a float64 oracle; north_star asks 1e-5
       relative, which the large-norm tensors meet -- the looser bound covers tiny tensors whose
       error is absolute fp32 noise)."""
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
import make_golden as MG                               # noqa: E402
from oracle import asr_oracle as O                     # noqa: E402


def load(cfg):
    z = np.load(os.path.join(HERE, 'golden', cfg + '.npz'), allow_pickle=False)
    n = int(z['n_params'])
    vals = [z['p%03d' % i] for i in range(n)]
    return z, json.loads(str(z['desc'])), vals, n


def relmax(a, b, floor=0.0):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + floor + 1e-30))


PLAIN = [c for c in MG.CONFIGS if c not in MG.TBPTT]


@pytest.mark.parametrize('cfg', PLAIN)
def test_oracle_reproduces_golden(cfg):
    z, desc, vals, n = load(cfg)
    onet = O.OracleNetwork(desc, vals)
    cond = z['ivec'] if 'ivec' in z else None
    post = onet.posteriors(z['x'], z['mask'], cond)
    np.testing.assert_allclose(post, z['posteriors'], rtol=1e-12, atol=1e-14)
    _, ce_sum, pred, grads = onet.loss_and_grads(z['x'], z['mask'], z['y'], z['mask'], cond)
    assert abs(ce_sum - float(z['ce_frame_sum'])) < 1e-10
    assert np.array_equal(pred, z['pred_idx'])
    for i, g in enumerate(grads):
        np.testing.assert_allclose(g, z['g%03d' % i], rtol=1e-10, atol=1e-13)
    # the reference's loss normaliser is the batch size (libs/comp_graph_utils.py:40)
    assert abs(float(z['ce_cost']) - float(z['ce_frame_sum']) / z['x'].shape[0]) < 1e-12
    # fixtures exercise what they claim: ragged rows, an active clip, a non-identity adaptive path
    lens = z['mask'].sum(axis=1)
    assert lens.max() == z['x'].shape[1] and lens.min() == 1


def test_oracle_reproduces_golden_tbptt():
    cfg = 'cfg5_lhuc_tbptt'
    z, desc, vals, n = load(cfg)
    onet = O.OracleNetwork(desc, vals)
    for l in onet.tbptt_layers():
        l.reset(z['x'].shape[0])
    full = (z['x'], z['mask'], z['ivec'], z['mask'], z['y'], z['mask'])
    wins = list(O.gen_win(full, int(z['win']), int(z['right_context'])))
    assert len(wins) == int(z['n_windows']) == 3 and wins[-1][0].shape[1] < int(z['win'])
    for w, (xw, mw, iw, _, yw, tmw) in enumerate(wins):
        _, ce_sum, pred, grads = onet.loss_and_grads(xw, mw, yw, tmw, iw, context=int(z['win']))
        assert abs(ce_sum - float(z['w%d_ce_frame_sum' % w])) < 1e-10
        assert np.array_equal(pred, z['w%d_pred_idx' % w])
        for i, g in enumerate(grads):
            np.testing.assert_allclose(g, z['w%d_g%03d' % (w, i)], rtol=1e-10, atol=1e-13)


def test_golden_clip_is_active_somewhere():
    """grad_clipping must bite in the clipped fixtures, else they would not pin grad_clip semantics:
    the gradients must differ from the same network evaluated with clipping off."""
    z, desc, vals, n = load('cfg3_hyperlstm')
    off = json.loads(json.dumps(desc))
    for L in off['layers']:
        for d in ('fwd', 'bwd'):
            if L.get(d):
                L[d]['args']['grad_clipping'] = 0.0
    g_on = O.OracleNetwork(desc, vals).loss_and_grads(z['x'], z['mask'], z['y'], z['mask'])[3]
    g_off = O.OracleNetwork(off, vals).loss_and_grads(z['x'], z['mask'], z['y'], z['mask'])[3]
    assert max(np.max(np.abs(a - b)) for a, b in zip(g_on, g_off)) > 1e-4


# ------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize('cfg', PLAIN)
def test_hip_path_matches_golden_through_api(cfg):
    import torch
    from asr_amd import comp_graph_utils as CG
    from asr_amd.updates import adam
    assert torch.cuda.is_available()
    z, desc, vals, n = load(cfg)
    net, cond_key = MG.build(cfg, 'cuda:0')
    assert net.describe() == desc                      # the builder wires what the fixture froze
    from asr_amd.utils import set_model_param_value
    set_model_param_value(net.get_all_params(trainable=True), vals)
    x, m, y = z['x'], z['mask'], z['y']
    iv = z['ivec'] if cond_key else None
    ivs = (iv,) if cond_key else ()
    flag = True if cond_key else None
    valid = m[:, :, None] > 0
    # ff: posteriors (probabilities for build_* nets, logits for gating_hyper_nets models)
    post, = CG.ff(net, True, True, flag)(x, m, *ivs)
    assert post.shape == z['posteriors'].shape and post.dtype == np.float32
    assert relmax(post * valid, z['posteriors'] * valid) < 2e-5
    # log-posteriors as the forward scripts write them
    if desc['output'] == 'logits':
        lp, = CG.ff(net, True, True, flag, log=True)(x, m, *ivs)
    else:
        lp = np.log(post)
    assert relmax(lp * valid, z['log_posteriors'] * valid) < 2e-5
    # predictor
    ce, pred = CG.predictor(True, True, True, True, net, ivector_data=flag)(x, m, *ivs, y, m)
    assert abs(float(ce) - float(z['ce_frame_sum'])) < 2e-5 * abs(float(z['ce_frame_sum']))
    assert pred.dtype == np.int64 and ((pred == z['pred_idx']) | (m == 0)).all()
    # trainer: loss, grads_norm, every gradient, one Adam step
    training_fn, tp = CG.trainer(True, True, True, True, net, adam, 1e-3, ivector_data=flag)
    ce, gn = training_fn(x, m, *ivs, y, m)
    assert abs(float(ce) - float(z['ce_frame_sum'])) < 2e-5 * abs(float(z['ce_frame_sum']))
    assert abs(float(gn) - float(z['grads_norm'])) < 1e-4 * float(z['grads_norm'])
    gmax = max(np.max(np.abs(z['g%03d' % i])) for i in range(n))
    for i, (p, g) in enumerate(zip(net.get_all_params(), net.get_all_grad_values())):
        assert relmax(g, z['g%03d' % i], floor=1e-3 * gmax) < 1e-4, (cfg, p.name)
    for i, p in enumerate(net.get_all_params()):
        assert relmax(p.get_value(), z['a%03d' % i], floor=1e-3) < 1e-5, (cfg, p.name)
    assert float(tp['t_prev'].get_value()) == 1.0


@pytest.mark.gpu
def test_hip_path_matches_golden_tbptt():
    import torch
    from asr_amd import comp_graph_utils as CG
    from asr_amd.updates import adam
    from asr_amd.utils import gen_win, set_model_param_value
    cfg = 'cfg5_lhuc_tbptt'
    z, desc, vals, n = load(cfg)
    net, _ = MG.build(cfg, 'cuda:0')
    assert net.describe() == desc
    tb = net.tbptt_layers()
    win, rc = int(z['win']), int(z['right_context'])
    full = (z['x'], z['mask'], z['ivec'], z['mask'], z['y'], z['mask'])
    # predictor_tbptt over the windows (weights fixed): loss + argmax per window, state carried
    set_model_param_value(net.get_all_params(), vals)
    pfn = CG.predictor_tbptt(True, True, True, True, net, tb, True, 0, win, ivector_data=True)
    for l in tb:
        l.reset(z['x'].shape[0])
    for w, (xw, mw, iw, _, yw, tmw) in enumerate(gen_win(full, win, rc)):
        ce, pred = pfn(xw, mw, iw, yw, tmw, 1 if w == 0 else 0)
        want = float(z['w%d_ce_frame_sum' % w])
        assert abs(float(ce) - want) <= 2e-5 * max(abs(want), 1.0)
        assert ((pred == z['w%d_pred_idx' % w]) | (tmw[:, :win] == 0)).all()
    # trainer_tbptt with lr = 0 keeps the weights of the fixture: gradients per window, un-rooted norm
    tfn, _ = CG.trainer_tbptt(True, True, True, True, net, adam, 0.0, tb, True, 0, win, ivector_data=True)
    for l in tb:
        l.reset(z['x'].shape[0])
    for w, (xw, mw, iw, _, yw, tmw) in enumerate(gen_win(full, win, rc)):
        ce, gsq = tfn(xw, mw, iw, yw, tmw, 1 if w == 0 else 0)
        want = float(z['w%d_grads_sqsum' % w])
        assert abs(float(gsq) - want) <= 2e-4 * want
        gmax = max(np.max(np.abs(z['w%d_g%03d' % (w, i)])) for i in range(n))
        for i, g in enumerate(net.get_all_grad_values()):
            assert relmax(g, z['w%d_g%03d' % (w, i)], floor=1e-3 * gmax) < 1e-4, (w, i)


def test_philox_known_answers_and_dropout_mask():
    """The generator behind asr_dropout, pinned to the published Random123 known-answer vectors for
    philox4x32-10 (kat_vectors: zero counter/key, all-ones, and the digits-of-pi case)."""
    from oracle import asr_oracle as O
    z = O.philox4x32_10(np.zeros((1, 4), np.uint32), (0, 0))[0]
    assert [int(v) for v in z] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    f = O.philox4x32_10(np.full((1, 4), 0xffffffff, np.uint32), (0xffffffff, 0xffffffff))[0]
    assert [int(v) for v in f] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    p = O.philox4x32_10(np.array([[0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344]], np.uint32), (0xa4093822, 0x299f31d0))[0]
    assert [int(v) for v in p] == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    m = O.dropout_mask(2000, 37, 0.2, 12345, 7)
    assert m.shape == (2000, 37) and set(np.unique(m).tolist()) == {0.0, np.float32(1.25)}
    assert abs((m > 0).mean() - 0.8) < 0.01 and np.array_equal(m, O.dropout_mask(2000, 37, 0.2, 12345, 7))
    assert not np.array_equal(m, O.dropout_mask(2000, 37, 0.2, 12345, 8))
    assert (O.dropout_mask(10, 5, 0.0, 1, 1) == 1.0).all()
