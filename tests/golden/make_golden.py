#!/usr/bin/env python
"""Generates tests/golden/*.npz: inputs, weights and float64 oracle outputs for the five BASELINE
configs at reduced size: B=3 ragged (one full-length, one length-1 row), T=7, D=5, N=8,
P in {0,4}, H=4, Z=2, F=3, I=3, C=6, both directions, grad_clipping active on some elements.

PARITY UNPINNED: the reference (Python-2 Theano/Lasagne) can neither be imported nor run in this
image and ships no fixtures for this path, so these vectors come from the repo's own float64
restatement (oracle/asr_oracle.py, itself pinned by torch autograd + finite differences).  They
freeze that restatement: the CPU suite checks the oracle still reproduces them bit-for-tolerance, the
GPU suite checks the HIP path against them.

Run from the repo root (CPU only):  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

from common import make_batch                      # noqa: E402
from oracle import asr_oracle as O                 # noqa: E402

B, T, D, C, N, H, Z, F, I = 3, 7, 5, 6, 8, 4, 2, 3, 3


def build(cfg, device='cpu'):
    """Builder call for each reduced config (the same call the GPU test makes on cuda:0)."""
    from asr_amd import builders, gating_hyper_nets as G
    if cfg == 'cfg1_lstm_bidir':
        return builders.build_deep_lstm(None, None, D, 3, N, 0, C, 0.0, True, False, 0, device=device, seed=1), None
    if cfg == 'cfg1_lstm_uni':
        return builders.build_deep_lstm(None, None, D, 3, N, 0, C, 0.0, False, False, 0, device=device, seed=1), None
    if cfg == 'cfg2_ln_lstmp':
        return G.deep_projection_ln_lstm_model(None, None, D, C, 2, 4, N, grad_clipping=0.05, dropout=0.0,
                                               device=device, seed=2), None
    if cfg == 'cfg3_hyperlstm':
        return builders.build_deep_hyperlstm('hyper', None, None, D, 2, N, H, Z, C, 0.05, True, 0,
                                             device=device, seed=3), None
    if cfg == 'cfg4_ivec_ln_lstmp':
        net, _ = G.deep_projection_ivector_ln_model(None, None, None, D, C, 2, 1, F, N, grad_clipping=0.05,
                                                    dropout=0.0, cond_dim=I, device=device, seed=4)
        return net, 'ivec'
    if cfg == 'cfg5_lhuc_tbptt':
        net, _, tb = builders.build_deep_uni_lhuclstm_ivector_tbptt(
            'IVectorLHUCLSTMLayer', None, None, D, 2, N, 0, C, 1, 5, 0, 0, 0, B, 0.05, False, False, I, True,
            device=device, seed=5)
        return net, 'ivec'
    if cfg == 'a5b_scaling_hyper':                 # ScalingHyperLSTMLayer, inner width 6 (padded to 8 on the GPU)
        return G.deep_scaling_hyper_model(None, None, D, C, [6, 4], [N, N], grad_clipping=0.05, device=device,
                                          seed=6), None
    if cfg == 'a5b_hyperlhuc':                     # HyperLHUCLSTMLayer, reparam relu
        return builders.build_deep_hyperlhuclstm('hyperlhuc', None, None, D, 2, N, H, Z, C, 0.05, True, 0,
                                                 device=device, seed=7), None
    if cfg == 'a3_max_cond_ln':                    # MaxCondLayerNormProjectLSTMLayer + ProjectLSTMLayer
        net, _ = G.deep_projection_max_cond_ln_model(None, None, D, C, 2, 1, F, N, grad_clipping=0.05, dropout=0.0,
                                                     device=device, seed=8)
        return net, None
    if cfg == 'a5b_projection_hyper':              # ProjectionHyperLSTMLayer (4 factors) + ProjectLSTMLayer
        return G.deep_projection_hyper_model(None, None, D, C, 2, 4, N, grad_clipping=0.05, device=device,
                                             seed=9)[0], None
    raise KeyError(cfg)


CONFIGS = ['cfg1_lstm_bidir', 'cfg1_lstm_uni', 'cfg2_ln_lstmp', 'cfg3_hyperlstm', 'cfg4_ivec_ln_lstmp',
           'cfg5_lhuc_tbptt', 'a5b_scaling_hyper', 'a5b_hyperlhuc', 'a3_max_cond_ln', 'a5b_projection_hyper']
TBPTT = dict(cfg5_lhuc_tbptt=dict(win=3, right_context=0, T=8, lengths=[8, 1, 5]))


def perturbed_values(net, seed, perturb=0.15):
    rng = np.random.default_rng(seed)
    return [(v + rng.standard_normal(v.shape) * perturb).astype(np.float32) for v in net.get_all_param_values()]


def generate(cfg):
    net, cond_key = build(cfg)
    k = CONFIGS.index(cfg)
    vals = perturbed_values(net, 100 + k)
    desc = net.describe()
    tb = TBPTT.get(cfg)
    rng = np.random.default_rng(200 + k)
    if tb:
        b = make_batch(rng, B, tb['T'], D, C, lengths=tb['lengths'], I=I)
    else:
        b = make_batch(rng, B, T, D, C, I=I if cond_key else 0)
    out = dict(desc=json.dumps(desc), n_params=len(vals), x=b['x'], mask=b['mask'], y=b['y'])
    if cond_key:
        out['ivec'] = b['ivec'].astype(np.float32)
    for i, v in enumerate(vals):
        out['p%03d' % i] = v
    onet = O.OracleNetwork(desc, vals)
    cond = out.get('ivec')
    if not tb:
        out['posteriors'] = onet.posteriors(b['x'], b['mask'], cond)
        lg = onet.logits
        out['log_posteriors'] = O.log_softmax(lg)
        ce_cost, ce_sum, pred, grads = onet.loss_and_grads(b['x'], b['mask'], b['y'], b['mask'], cond)
        out['ce_frame_sum'] = np.float64(ce_sum); out['ce_cost'] = np.float64(ce_cost)
        out['pred_idx'] = pred.astype(np.int64)
        for i, g in enumerate(grads):
            out['g%03d' % i] = g
        out['grads_norm'] = np.float64(np.sqrt(sum((g ** 2).sum() for g in grads)))
        new = [v.astype(np.float64) for v in vals]
        m = [np.zeros_like(v) for v in new]; vv = [np.zeros_like(v) for v in new]
        O.adam_step(new, grads, m, vv, 0)
        for i, v in enumerate(new):
            out['a%03d' % i] = v
    else:
        for l in onet.tbptt_layers():
            l.reset(B)
        full = (b['x'], b['mask'], out['ivec'], b['mask'], b['y'], b['mask'])
        nwin = 0
        for w, win in enumerate(O.gen_win(full, tb['win'], tb['right_context'])):
            xw, mw, iw, _, yw, tmw = win
            ce_cost, ce_sum, pred, grads = onet.loss_and_grads(xw, mw, yw, tmw, iw, context=tb['win'])
            out['w%d_ce_frame_sum' % w] = np.float64(ce_sum)
            out['w%d_pred_idx' % w] = pred.astype(np.int64)
            out['w%d_grads_sqsum' % w] = np.float64(sum((g ** 2).sum() for g in grads))
            for i, g in enumerate(grads):
                out['w%d_g%03d' % (w, i)] = g
            nwin += 1
        out['n_windows'] = nwin
        out['win'] = tb['win']; out['right_context'] = tb['right_context']
    return out


def main():
    for cfg in (sys.argv[1:] or CONFIGS):          # optional: names of the configs to (re)generate
        out = generate(cfg)
        path = os.path.join(HERE, cfg + '.npz')
        np.savez_compressed(path, **out)
        print('%-22s %6.1f KB  %d arrays' % (cfg, os.path.getsize(path) / 1024.0, len(out)))


if __name__ == '__main__':
    main()
