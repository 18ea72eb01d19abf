"""The optional bf16 mode (BASELINE.json north_star: "with a stated bound in the optional bf16 mode"): the large
GEMMs round their operands to bf16 and accumulate in fp32 (asr_gemm_bf16_ws); the recurrences stay fp32.

  * kernel: against float64 products of the bf16-ROUNDED operands the kernel is as exact as the fp32 one
    (a bf16 x bf16 product is exact in fp32), for every transpose form, ragged shapes, split-K, bias, beta;
  * stated bound, measured on MI355X against the float64 oracle at full width and then frozen with ~2x margin:
      cfg3 (4x512 HyperLSTM, T = 80):  log-posteriors  max |d log p| <= 1e-1   (measured 4.4e-2, mean 3.5e-3)
                                       CE frame sum    relative      <= 1e-3   (measured 1.7e-5)
                                       gradient, all tensors as one vector: ||g - g_ref|| / ||g_ref|| <= 0.3
                                                                               (measured 0.14; worst tensor 0.22)
      The gradient figure is large because these recurrent nets amplify any perturbation: fp32 arithmetic itself
      (eps 6e-8) loses 7e-4 of the cfg3 gradient at full length (profiles/parity_r02.json), bf16 operands (eps
      4e-3) in the GEMMs alone 0.14 at T = 80.  SURVEY section 7's proposed 2e-2 / 5e-2 is not met.
      cfg2 (4-layer LN-LSTMP): NO bound can be stated -- measured |d log p| 0.59 max, gradient error 0.93: the layer
      norm stack turns bf16 rounding into O(1) differences (it does the same to fp32 rounding at 1e-3 scale).  So
      Network.set_precision('bf16') REFUSES networks that normalise inside the recurrence (cfg2, cfg4 and their
      relatives); the test checks the refusal and records the numbers under allow_unbounded=True.
  * the two-part mode 'bf16x2' (asr_gemm_bf16x3_ws with terms = 3; round 3) has a bound on EVERY config, frozen at ~4x the
    measurement: cfg3 3e-4 / 1e-3 (log-posteriors / whole gradient; measured 7.5e-5 / 2.5e-4), cfg2 and cfg4 1e-2 / 2e-2
    (2.6e-3 / 4.9e-3, 1.8e-3 / 3.2e-3), cfg1 1e-5 / 3e-5 (2.3e-6 / 6.8e-6): SURVEY section 7's proposal (2e-2 / 5e-2) is met
    everywhere, at 5 % (cfg3) and 2 % (LN stacks) more frames/s than the default (profiles/r03_precision_modes.txt).
    measured values are printed and recorded in gpurun_out/parity_r0x.jsonl."""
import json
import os

import numpy as np
import pytest
import torch

from asr_amd import _lib, ops
from oracle import asr_oracle as O
from test_gpu_fullsize import build_cfg, _t, _batch
from test_gpu_parity import perturb_params

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BOUND_LOGPOST, BOUND_CE, BOUND_GRAD_ALL = 1e-1, 1e-3, 0.3


def bf16_round(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) & 0xFFFF0000).astype(np.uint32).view(np.float32)


@pytest.mark.parametrize('ta,tb', [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_bf16_is_exact_on_rounded_operands(ta, tb):
    L = _lib.lib()
    rng = np.random.default_rng(3 + 2 * ta + tb)
    ws = torch.empty(8 << 20, device='cuda')
    for M, N, K in ((128, 128, 64), (200, 136, 50), (37, 300, 1000), (256, 128, 4096), (5, 7, 3)):
        A = rng.standard_normal((K, M) if ta else (M, K)).astype(np.float32)
        B = rng.standard_normal((N, K) if tb else (K, N)).astype(np.float32)
        C0 = rng.standard_normal((M, N + 3)).astype(np.float32)
        bias = rng.standard_normal(N).astype(np.float32)
        dA, dB, dC, db = _t(A), _t(B), _t(C0), _t(bias)
        rc = L.asr_gemm_bf16_ws(ta, tb, M, N, K, 0.5, dA.data_ptr(), A.shape[1], dB.data_ptr(), B.shape[1], 1.0,
                                dC.data_ptr(), N + 3, db.data_ptr(), ws.data_ptr(), ws.numel() * 4,
                                torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        Ar, Br = bf16_round(A).astype(np.float64), bf16_round(B).astype(np.float64)
        ref = 0.5 * ((Ar.T if ta else Ar) @ (Br.T if tb else Br)) + C0[:, :N] + bias
        got = dC.cpu().numpy()
        assert np.abs(got[:, :N] - ref).max() / np.abs(ref).max() < 5e-6, (M, N, K)
        assert (got[:, N:] == C0[:, N:]).all()


def _measure(net, ck, b, vals, mode, **kw):
    x, m, y = _t(b['x']), _t(b['mask']), _t(b['y'])
    c = _t(b[ck]) if ck else None
    onet = O.OracleNetwork(net.describe(), vals)
    _, os_, _, og = onet.loss_and_grads(b['x'], b['mask'], b['y'], b['mask'], b.get(ck))
    olp = O.log_softmax(onet.logits)
    try:
        net.set_precision(mode, **kw)
        lp = net.posteriors(x, m, c, log=True).cpu().numpy()
        net.check_status()
        ce, _, _ = net.loss(x, m, y, m, c)
        net.backward()
        net.check_status()
        grads = net.get_all_grad_values()
    finally:
        net.set_precision(ops.DEFAULT_PRECISION)
    valid = b['mask'] > 0
    gn = [np.linalg.norm(g) for g in og]
    e_all = float(np.sqrt(sum(np.linalg.norm(g - o) ** 2 for g, o in zip(grads, og))) / np.sqrt(sum(n * n for n in gn)))
    return dict(logpost_abs=float(np.abs(lp - olp)[valid].max()), logpost_abs_mean=float(np.abs(lp - olp)[valid].mean()),
                ce_rel=abs(ce.item() - os_) / abs(os_), all_grad_norm_rel=e_all)


# the two-part mode ('bf16x2': 3 bf16 MFMA products per fp32-grade one, 16 significand bits per operand), measured on
# MI355X and frozen with ~2-3x margin.  It meets SURVEY section 7's proposal (2e-2 on log-posteriors, 5e-2 on the
# gradient) on EVERY config, the layer-norm stacks the plain bf16 mode is refused for included.
BOUNDS_X2 = {3: (3e-4, 1e-3), 2: (1e-2, 2e-2), 4: (1e-2, 2e-2), 1: (1e-5, 3e-5)}      # measured: 7.5e-5 / 2.5e-4, 2.6e-3 / 4.9e-3, 1.8e-3 / 3.2e-3, 2.3e-6 / 6.8e-6


@pytest.mark.parametrize('cfg,max_len', [(3, 80), (2, 40), (4, 40), (1, 200)])
def test_bf16x2_mode_meets_the_stated_bound(cfg, max_len):
    net, ck = build_cfg(cfg)
    vals = perturb_params(net, 30 + cfg, 0.02)
    b = _batch(cfg, max_len=max_len)
    r = _measure(net, ck, b, vals, 'bf16x2')
    print('cfg%d bf16x2 mode: %s' % (cfg, r))
    d = os.path.join(ROOT, 'gpurun_out')
    if os.path.isdir(d):
        with open(os.path.join(d, 'parity_%s.jsonl' % os.environ.get('ASR_PARITY_TAG', 'r03')), 'a') as f:
            f.write(json.dumps(dict(cfg=cfg * 100 + 1, of_cfg=cfg, T=int(b['x'].shape[1]),
                                    mode='bf16x2 GEMMs (operands as two bf16 parts, 3 products, fp32 accumulate), fp32 recurrences',
                                    bound_logpost_abs=BOUNDS_X2[cfg][0], bound_all_grad_norm_rel=BOUNDS_X2[cfg][1], **r)) + '\n')
    assert r['logpost_abs'] <= BOUNDS_X2[cfg][0] and r['all_grad_norm_rel'] <= BOUNDS_X2[cfg][1], r
    assert r['ce_rel'] <= 1e-3
    assert ops.PRECISION == ops.DEFAULT_PRECISION


@pytest.mark.parametrize('cfg,max_len', [(3, 80), (2, 40)])
def test_bf16_mode_meets_the_stated_bound(cfg, max_len):
    net, ck = build_cfg(cfg)
    vals = perturb_params(net, 30 + cfg, 0.02)
    b = _batch(cfg, max_len=max_len)
    x, m, y = _t(b['x']), _t(b['mask']), _t(b['y'])
    c = _t(b[ck]) if ck else None
    onet = O.OracleNetwork(net.describe(), vals)
    _, os_, _, og = onet.loss_and_grads(b['x'], b['mask'], b['y'], b['mask'], b.get(ck))
    olp = O.log_softmax(onet.logits)
    if cfg != 3:                  # layer norm inside the recurrence: the mode is refused unless overridden
        with pytest.raises(_lib.AsrError):
            net.set_precision('bf16')
        assert ops.PRECISION == ops.DEFAULT_PRECISION
    try:
        net.set_precision('bf16', allow_unbounded=(cfg != 3))
        lp = net.posteriors(x, m, c, log=True).cpu().numpy()
        net.check_status()
        ce, _, _ = net.loss(x, m, y, m, c)
        net.backward()
        net.check_status()
        grads = net.get_all_grad_values()
    finally:
        net.set_precision(ops.DEFAULT_PRECISION)
    valid = b['mask'] > 0
    e_lp = float(np.abs(lp - olp)[valid].max())
    gn = [np.linalg.norm(g) for g in og]
    big = max(gn)
    errs = [(float(np.linalg.norm(g - o) / n), nm) for (nm, _), g, o, n in
            zip(net.param_names_shapes(), grads, og, gn) if n > 1e-3 * big]
    e_g, worst = max(errs)
    e_all = float(np.sqrt(sum(np.linalg.norm(g - o) ** 2 for g, o in zip(grads, og))) / np.sqrt(sum(n * n for n in gn)))
    e_ce = abs(ce.item() - os_) / abs(os_)
    e_lp_mean = float(np.abs(lp - olp)[valid].mean())
    print('cfg%d bf16 mode: |d log p| max %.2e mean %.2e, CE %.2e, gradient: all tensors together %.2e, worst tensor '
          '%.2e (%s)' % (cfg, e_lp, e_lp_mean, e_ce, e_all, e_g, worst))
    print('   worst tensors:', sorted(errs, reverse=True)[:6])
    d = os.path.join(ROOT, 'gpurun_out')
    if os.path.isdir(d):
        with open(os.path.join(d, 'parity_r02.jsonl'), 'a') as f:
            f.write(json.dumps(dict(cfg=cfg * 100, mode='bf16 GEMMs (operands rounded to bf16, fp32 accumulate), fp32 recurrences',
                                    of_cfg=cfg, T=int(b['x'].shape[1]), logpost_abs=e_lp, ce_rel=e_ce, worst_grad_norm_rel=e_g, all_grad_norm_rel=e_all, logpost_abs_mean=e_lp_mean,
                                    worst_grad=worst, bound_logpost_abs=BOUND_LOGPOST if cfg == 3 else None,
                                    bound_all_grad_norm_rel=BOUND_GRAD_ALL if cfg == 3 else None)) + '\n')
    assert np.isfinite(e_lp) and np.isfinite(e_all)
    if cfg == 3:
        assert e_lp <= BOUND_LOGPOST and e_ce <= BOUND_CE and e_all <= BOUND_GRAD_ALL, (e_lp, e_ce, e_all)
    assert ops.PRECISION == ops.DEFAULT_PRECISION
