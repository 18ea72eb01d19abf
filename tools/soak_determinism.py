"""Soak: the same batch through loss + backward many times; every repetition must reproduce the first one's loss and flat
gradient BIT FOR BIT.  The persistent scans hand data between workgroups inside a launch while parameter-gradient GEMMs run
beside them on the side stream (uneven load): a stale or torn hand-off read would show as a differing bit here.
  python tools/soak_determinism.py [reps]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asr_amd import builders, gating_hyper_nets as G, workloads

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = 'cuda:0'
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
cases = [
    ('cfg3 4x512 HyperLSTM', 3, lambda: builders.build_deep_hyperlstm('hyper', None, None, 123, 4, 512, 128, 64, 3436, 1.0, True, 0,
                                                                       device=dev, seed=1), reps),
    ('cfg2 4x512 LN-LSTMP', 2, lambda: G.deep_projection_ln_lstm_model(None, None, 123, 3436, 4, None, 512, 1.0, 0.0, device=dev,
                                                                       seed=1), reps),
    ('cfg4 i-vector cond-LN', 4, lambda: G.deep_projection_ivector_ln_model(None, None, None, 123, 4174, 4, 4, 64, 512, 1.0, 0.0,
                                                                            cond_dim=100, device=dev, seed=1)[0], reps),
    ('cfg1 3x256 LSTM', 1, lambda: builders.build_deep_lstm(None, None, 123, 3, 256, 0, 61, 1.0, True, False, 0, device=dev,
                                                            seed=1), 4 * reps),
    # windows 1..: stateful layers reset before every repetition, first TBPTT window of the batch
    ('cfg5 LHUC TBPTT window', 5, lambda: builders.build_deep_uni_lhuclstm_ivector_tbptt(
        'IVectorLHUCLSTMLayer', None, None, 123, 4, 512, 0, 3436, 1, 128, 0, 0, 0, 16, 1.0, False, False, 100, True,
        device=dev, seed=1)[0], reps),
    # the step-loop layers (step_prjhyper.hip, step_scaling.hip), time axis cut to 200 frames
    ('ProjectionHyper 3x512', 3, lambda: G.deep_projection_hyper_model(None, None, 123, 3436, 3, 256, 512, grad_clipping=1.0,
                                                                       device=dev, seed=1)[0], max(reps // 4, 4)),
    ('ScalingHyper options', 1, lambda: G.deep_scaling_hyper_model(None, None, 123, 61, [64], [128], use_layer_norm=True,
                                                                   learn_init=True, use_peepholes=True, grad_clipping=1.0,
                                                                   device=dev, seed=1), max(reps // 4, 4)),
    ('ScalingHyper scan 250/500', 1, lambda: G.deep_scaling_hyper_model(None, None, 123, 61, [250, 250], [500, 500],
                                                                        grad_clipping=1.0, device=dev, seed=1), reps),
    # kernel variants the five configurations do not reach: per-gate LN (2-D backward, PG = 1), LN at a width the 2-D
    # backward does not take (N = 320: lstm_scan_bwd_kernel's LN path), a net wider than the register chunk (N = 1024)
    ('per-gate LN 2x512 (v0_2)', 2, lambda: G._build_rec(123, 3436, 2, 512, 1.0, 0.0, 'LayerNormProjectLSTM_v0_2_Layer', None, 0,
                                                         num_factors=None, num_prjs=256, device=dev, seed=1)[0], reps),
    ('cond per-gate LN 2x512', 2, lambda: G._build_rec(123, 3436, 2, 512, 1.0, 0.0, 'LayerNormProjectLSTM_v0_1_Layer',
                                                       'CondLayerNormProjectLSTM_v0_2_Layer', 2, num_factors=64,
                                                       num_prjs=256, device=dev, seed=1)[0], reps),
    ('LN-LSTM 2x320', 1, lambda: builders.build_deep_lstm(None, None, 123, 2, 320, 0, 61, 1.0, True, True, 0, device=dev,
                                                          seed=1), reps),
    ('LSTMP 2x1024', 1, lambda: builders.build_deep_lstm(None, None, 123, 2, 1024, 256, 61, 1.0, True, False, 0, device=dev,
                                                         seed=1), reps),
    ('HyperLSTM 2x1024', 1, lambda: builders.build_deep_hyperlstm('hyper', None, None, 123, 2, 1024, 128, 64, 61, 1.0, True, 0,
                                                                  device=dev, seed=1), max(reps // 2, 4)),
]
# round 3: 32 rows = two 16-row chunks per workgroup (teams) in the hyper scans, the generic LSTM forward scan and the 2-D
# LN backward; time axis cut to 600 frames
cases += [('cfg3 32 rows (teams)', 3, cases[0][2], max(reps // 2, 4)), ('cfg2 32 rows (teams)', 2, cases[1][2], max(reps // 2, 4)),
          ('cfg4 32 rows (teams)', 4, cases[2][2], max(reps // 2, 4)),
          ('per-gate LN 32 rows (teams)', 2, cases[8][2], max(reps // 2, 4))]
# round 3, second half: the option form of the generic scans (peepholes + learned initial rows + cell_out) and the
# streamed-weight form (2048 units in one direction)
cases += [('LSTMP options 2x512', 1, lambda: G._build_rec(123, 61, 2, 512, 1.0, 0.0, 'ProjectLSTMLayer', None, 0, num_factors=256,
                                                          device=dev, seed=1, rec_options=dict(peepholes=True, learn_init=True,
                                                                                               only_return_hidden=False))[0], reps),
          ('LN-LSTMP learn_init 2x512', 1, lambda: G._build_rec(123, 61, 2, 512, 1.0, 0.0, 'LayerNormProjectLSTMLayer', None, 0,
                                                                num_factors=256, device=dev, seed=1,
                                                                rec_options=dict(learn_init=True))[0], reps),
          ('LSTM 1x2048 streamed', 1, lambda: builders.build_deep_lstm(None, None, 123, 1, 2048, 0, 61, 1.0, False, False, 0,
                                                                       device=dev, seed=1), max(reps // 2, 4))]
bad = 0
only = os.environ.get('SOAK_ONLY', '')          # substring of a case name: only those cases
for name, cfg, mk, n in cases:
    if only and only not in name:
        continue
    net = mk()
    steploop = name.startswith(('ProjectionHyper', 'ScalingHyper options'))
    teams = '(teams)' in name
    b = workloads.make_batch(cfg, 32 if teams else 16, max_len=200 if steploop else 600 if teams else None)
    x, m, y = t(b['x']), t(b['mask']), t(b['y'])
    iv = t(b['ivec']) if 'ivec' in b else None
    tb = net.tbptt_layers()
    if tb:                                        # one window of 100 frames
        x, m, y, iv = (a[:, :100].contiguous() for a in (x, m, y, iv))
    ref_g = None
    t0 = time.time()
    diff = 0
    for r in range(n):
        if tb:
            for l in tb:
                l.reset(16)
            ce, _, _ = net.loss(x, m, y, m, iv, True, 100)
        else:
            ce, _, _ = net.loss(x, m, y, m, iv)
        net.backward()
        g = net.store.g
        if ref_g is None:
            torch.cuda.synchronize()
            ref_g = g.clone(); ref_ce = ce.clone(); ref_lg = net.logits.clone()
        elif not (torch.equal(g, ref_g) and torch.equal(ce, ref_ce)):
            diff += 1
            if diff <= 3:
                d = (g - ref_g).abs()
                i = int(d.argmax())
                print('   rep %d: ce %s (%.9g vs %.9g), max |dg| %.3e at %d of %d (|g| max %.3e), %d elements differ, logits equal: %s'
                      % (r, 'equal' if torch.equal(ce, ref_ce) else 'DIFFERS', float(ce), float(ref_ce), float(d.max()), i,
                         g.numel(), float(ref_g.abs().max()), int((d > 0).sum()), torch.equal(net.logits, ref_lg)), flush=True)
    torch.cuda.synchronize()
    net.check_status()
    bad += diff
    print('%-24s %4d repetitions, %d differ from the first (%.1f s)' % (name, n, diff, time.time() - t0), flush=True)
    del net
    torch.cuda.empty_cache()
print('SOAK', 'FAILED' if bad else 'ok')
sys.exit(1 if bad else 0)
