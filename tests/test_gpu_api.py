"""GPU tests of the drop-in boundary: forward scripts -> Kaldi posterior archives, training loops ->
pickles -> resume, through the reference-shaped API; numbers checked against the oracle."""
import os
import pickle
import sys

import numpy as np
import pytest

from oracle import asr_oracle as O

pytestmark = pytest.mark.gpu


def _perturb(net, seed, s=0.1):
    rng = np.random.default_rng(seed)
    vals = [(v + rng.standard_normal(v.shape) * s).astype(np.float32) for v in net.get_all_param_values()]
    net.set_all_param_values(vals)
    return vals


def _write_feats(tmp_path, utts, with_ivec=False):
    from asr_amd import kaldi_io
    fp = str(tmp_path / 'feats.ark')
    with kaldi_io.BaseFloatMatrixWriter('ark:' + fp) as w:
        for u in utts:
            w.write(u.uttid, u.features)
    ip = None
    if with_ivec:
        ip = str(tmp_path / 'ivec.ark')
        with kaldi_io.BaseFloatMatrixWriter('ark:' + ip) as w:
            for u in utts:
                w.write(u.uttid, u.ivectors[:1])
    return fp, ip


def test_forward_lstm2_cli_writes_log_posteriors(tmp_path):
    """bin/forward_lstm2.py contract: per utterance a (valid_len, C) matrix of log(softmax)."""
    from asr_amd import data, forward, kaldi_io, utils
    from asr_amd.builders import build_deep_lstm
    D, C_, N = 7, 9, 16
    utts = data.synthetic_set(5, D, C_, min_len=3, max_len=21, seed=1)
    fp, _ = _write_feats(tmp_path, utts)
    net = build_deep_lstm(True, True, D, 2, N, 0, C_, 1.0, True, False, 0, device='cuda:0', seed=3)
    vals = _perturb(net, 4)
    model = str(tmp_path / 'm.pkl')
    utils.save_network(net, None, 3, model)
    out = str(tmp_path / 'post.ark')
    rc = forward.main(['lstm2', '--input-dim', str(D), '--output-dim', str(C_), '--num-layers', '2',
                       '--num-units', str(N), '--batch-size', '2', model, 'ark:' + fp, 'ark:' + out])
    assert rc == 0
    got = kaldi_io.read_mat_ark('ark:' + out)
    assert [k for k, _ in got] == [u.uttid for u in utts]
    onet = O.OracleNetwork(net.describe(), vals)
    for (k, m), u in zip(got, utts):
        T = u.features.shape[0]
        assert m.shape == (T, C_)
        want = np.log(onet.posteriors(u.features[None], np.ones((1, T)))[0])
        assert np.max(np.abs(m - want)) < 2e-5 * np.max(np.abs(want))
        assert np.allclose(np.exp(m.astype(np.float64)).sum(axis=1), 1.0, atol=1e-5)


def test_forward_lstm2_cli_skip_frames(tmp_path):
    """bin/forward_lstm2.py:86-102 with --skip 3: the net sees every third frame (libs/utils.py:208-229), its
    log-posteriors are repeated 3 times along time and trimmed to the utterance (numpy.repeat un-skipping, the
    idiom test/test_libs_utils.py:43-51 pins); an utterance shorter than the skip still gets all its frames."""
    from asr_amd import data, forward, kaldi_io, utils
    from asr_amd.builders import build_deep_lstm
    D, C_, N, SKIP = 7, 9, 16, 3
    utts = data.synthetic_set(5, D, C_, min_len=2, max_len=20, seed=11)
    fp, _ = _write_feats(tmp_path, utts)
    net = build_deep_lstm(True, True, D, 2, N, 0, C_, 1.0, True, False, 0, device='cuda:0', seed=3)
    vals = _perturb(net, 4)
    model = str(tmp_path / 'm.pkl')
    utils.save_network(net, None, 3, model)
    out = str(tmp_path / 'post.ark')
    rc = forward.main(['lstm2', '--input-dim', str(D), '--output-dim', str(C_), '--num-layers', '2', '--num-units',
                       str(N), '--batch-size', '2', '--skip', str(SKIP), model, 'ark:' + fp, 'ark:' + out])
    assert rc == 0
    got = kaldi_io.read_mat_ark('ark:' + out)
    assert [k for k, _ in got] == [u.uttid for u in utts]
    onet = O.OracleNetwork(net.describe(), vals)
    for (k, m), u in zip(got, utts):
        T = u.features.shape[0]
        sk = u.features[::SKIP]
        want = np.repeat(np.log(onet.posteriors(sk[None], np.ones((1, len(sk))))[0]), SKIP, axis=0)[:T]
        assert m.shape == (T, C_)
        assert np.max(np.abs(m - want)) < 2e-5 * np.max(np.abs(want))
        assert np.array_equal(m[0], m[min(SKIP, T) - 1])


def test_forward_ivec_ln_lstmp_cli(tmp_path):
    from asr_amd import data, forward, kaldi_io, utils
    from asr_amd.gating_hyper_nets import deep_projection_ivector_ln_model
    D, C_, N, F, I = 6, 8, 16, 3, 4
    utts = data.synthetic_set(3, D, C_, I=I, min_len=2, max_len=15, seed=2)
    fp, ip = _write_feats(tmp_path, utts, with_ivec=True)
    net, _ = deep_projection_ivector_ln_model(True, True, True, D, C_, 2, 1, F, N, 1.0, 0.0, cond_dim=I,
                                              device='cuda:0', seed=5)
    vals = _perturb(net, 6)
    model = str(tmp_path / 'm.pkl')
    utils.save_network(net, None, 1, model)
    out = str(tmp_path / 'post.ark')
    rc = forward.main(['ivec_ln_lstmp', '--input_dim', str(D), '--output_dim', str(C_), '--num_layers', '2',
                       '--num_conds', '1', '--num_factors', str(F), '--num_units', str(N), '--ivector_dim',
                       str(I), '--batch_size', '3', '--ivectors', 'ark:' + ip, model, 'ark:' + fp, 'ark:' + out])
    assert rc == 0
    got = dict(kaldi_io.read_mat_ark('ark:' + out))
    onet = O.OracleNetwork(net.describe(), vals)
    for u in utts:
        T = u.features.shape[0]
        lg = onet.posteriors(u.features[None], np.ones((1, T)), u.ivectors[None])[0]
        want = O.log_softmax(lg)
        assert got[u.uttid].shape == (T, C_)
        assert np.max(np.abs(got[u.uttid] - want)) < 2e-5 * np.max(np.abs(want))


class _Args(object):
    num_epochs = 2; log_freq = 2; batch_size = 2; use_ivector_input = False; delay = 0
    num_tbptt_steps = 5; right_context = 2; reload_model = None; no_reload = False


def test_training_loop_pickles_and_resume(tmp_path, capsys):
    """scripts/exp_deep_hyperlstm.py skeleton on a tiny HyperLSTM: the loss goes down, the per-epoch
    pickle has the reference layout and reloads into a fresh network, resume skips finished epochs."""
    from asr_amd import comp_graph_utils as CG, data, train, utils
    from asr_amd.builders import build_deep_hyperlstm
    from asr_amd.updates import adam
    D, C_ = 5, 4
    tr = data.UtteranceStream(data.synthetic_set(6, D, C_, min_len=4, max_len=12, seed=3), 2)
    va = data.UtteranceStream(data.synthetic_set(2, D, C_, min_len=4, max_len=12, seed=4), 2)
    args = _Args(); args.save_path = str(tmp_path / 'hyper')

    def make():
        net = build_deep_hyperlstm('hyper', True, True, D, 1, 8, 4, 2, C_, 1.0, True, 0, device='cuda:0', seed=9)
        tf, tp = CG.trainer(True, True, True, True, net, adam, 0.01)
        pf = CG.predictor(True, True, True, True, net)
        return net, tf, tp, pf

    net, tf, tp, pf = make()
    ce0, fer0 = CG.eval_net(pf, tr)
    hist = train.run_training(args, net, tf, tp, pf, tr, va, va)
    ce1, fer1 = CG.eval_net(pf, tr)
    assert ce1 < ce0 and len(hist) == 3
    last = args.save_path + '_last_model.pkl'
    obj = pickle.load(open(last, 'rb'))
    assert len(obj) == 3 and obj[1] is None and obj[2] == 2
    names = [p.name.split('.')[-1] for p in net.get_all_params()]
    assert len(obj[0]) == len(names) == 2 * 56 + 2 and names[0] == 'W_hhat_ig' and names[-2:] == ['W', 'b']
    assert os.path.exists(args.save_path + '_eval_history.pkl')
    # a fresh network restored from the pickle predicts the same
    net2, tf2, tp2, pf2 = make()
    utils.find_reload_model(args)
    assert args.reload_model == last
    _, done = utils.load_or_init_model(net2.get_all_params(), args)
    assert done == 2
    ce2, fer2 = CG.eval_net(pf2, tr)
    assert abs(ce2 - ce1) < 1e-6 * abs(ce1) and fer2 == fer1
    capsys.readouterr()
    train.run_training(args, net2, tf2, tp2, pf2, tr, va, va, pretrain_total_epoch_cnt=done)
    assert capsys.readouterr().out.count('Skip Epoch') == 2
    # wrong architecture -> the reference's ValueError
    net3 = build_deep_hyperlstm('hyper', True, True, D, 1, 12, 4, 2, C_, 1.0, True, 0, device='cuda:0', seed=9)
    with pytest.raises(ValueError, match='mismatch'):
        utils.set_model_param_value(net3.get_all_params(), obj[0])


def test_tbptt_training_loop_and_forward(tmp_path):
    """scripts/exp_deep_lstm2_tbptt.py skeleton + bin/forward_lstm2_tbptt.py on a stateful uni LSTM:
    windowed forward with carried state equals the un-windowed forward of the same weights."""
    from asr_amd import comp_graph_utils as CG, data, forward, kaldi_io, train, utils
    from asr_amd.builders import build_deep_lstm, build_deep_lstm_tbptt
    from asr_amd.updates import adam
    D, C_, N = 5, 4, 8
    utts = data.synthetic_set(4, D, C_, min_len=6, max_len=17, seed=5)
    tr = data.UtteranceStream(utts, 2)
    args = _Args(); args.save_path = str(tmp_path / 'tb'); args.num_epochs = 1; args.right_context = 0
    net, tb = build_deep_lstm_tbptt(True, True, D, 2, N, 0, C_, 2, args.num_tbptt_steps, 1.0, False, False, 0,
                                    device='cuda:0', seed=2)
    tf, tp = CG.trainer_tbptt(True, True, True, True, net, adam, 0.01, tb, True, 0, args.num_tbptt_steps)
    pf = CG.predictor_tbptt(True, True, True, True, net, tb, True, 0, args.num_tbptt_steps)
    ce0, _ = CG.eval_net_tbptt(pf, tr, tb, args.num_tbptt_steps, 2, 0)
    train.run_training_tbptt(args, net, tb, tf, tp, pf, tr, tr, tr)
    ce1, _ = CG.eval_net_tbptt(pf, tr, tb, args.num_tbptt_steps, 2, 0)
    assert ce1 < ce0
    model = args.save_path + '_last_model.pkl'
    fp, _ = _write_feats(tmp_path, utts)
    out = str(tmp_path / 'p_tb.ark'); out_full = str(tmp_path / 'p_full.ark')
    common = ['--input-dim', str(D), '--output-dim', str(C_), '--num-layers', '2', '--num-units', str(N),
              '--batch-size', '2', '--uni']
    assert forward.main(['lstm2_tbptt'] + common + ['--num-tbptt-steps', '5', model, 'ark:' + fp, 'ark:' + out]) == 0
    assert forward.main(['lstm2'] + common + [model, 'ark:' + fp, 'ark:' + out_full]) == 0
    a = dict(kaldi_io.read_mat_ark('ark:' + out)); b = dict(kaldi_io.read_mat_ark('ark:' + out_full))
    for u in utts:
        assert a[u.uttid].shape == b[u.uttid].shape == (u.features.shape[0], C_)
        assert np.max(np.abs(a[u.uttid] - b[u.uttid])) < 1e-5
