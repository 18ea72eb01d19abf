"""Kernel-level GPU checks through the C-ABI: GEMM (all transposes, ragged sizes, strides, bias,
accumulate), row layer norm, column reductions, softmax-CE, against float64 numpy."""
import numpy as np
import pytest
import torch

from oracle import asr_oracle as O

pytestmark = pytest.mark.gpu


def relerr(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - b)) / (np.max(np.abs(b)) + 1e-30))


@pytest.mark.parametrize('ta,tb', [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize('M,N,K', [(37, 123, 61), (256, 2048, 123), (1, 5, 3), (130, 129, 17), (64, 200, 9000)])
def test_gemm(ta, tb, M, N, K):
    from asr_amd import ops
    rng = np.random.default_rng(M * 7 + N)
    A = rng.standard_normal((K, M) if ta else (M, K)).astype(np.float32)
    B = rng.standard_normal((N, K) if tb else (K, N)).astype(np.float32)
    C0 = rng.standard_normal((M, N + 3)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    dA = torch.from_numpy(A).cuda(); dB = torch.from_numpy(B).cuda()
    dC = torch.from_numpy(C0).cuda(); db = torch.from_numpy(bias).cuda()
    ops.gemm(dA, dB, dC[:, :N], ta, tb, alpha=0.5, beta=1.0, bias=db)
    torch.cuda.synchronize()
    opA = A.T.astype(np.float64) if ta else A.astype(np.float64)
    opB = B.T.astype(np.float64) if tb else B.astype(np.float64)
    ref = 0.5 * (opA @ opB) + C0[:, :N] + bias
    got = dC.cpu().numpy()
    assert relerr(got[:, :N], ref) < 5e-6
    assert (got[:, N:] == C0[:, N:]).all()          # the strided view's neighbours are untouched


@pytest.mark.parametrize('terms', [6, 8])
@pytest.mark.parametrize('ta,tb', [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_split_bf16_is_fp32_grade(ta, tb, terms):
    """asr_gemm_bf16x3_ws (operands split into three bf16 parts, 6 or 8 partial products, fp32 accumulate) against
    float64, next to asr_gemm_f32_ws (fp32 MFMA) on the same inputs: the error, taken relative to |A|.|B| element by
    element, stays below 1.5e-6 and is not above the fp32 MFMA kernel's by more than rounding scatter.  Operands
    span +-6 powers of e so that every split part matters; shapes cover edge tiles, K tails, split-K, alpha / beta /
    bias and a strided output."""
    from asr_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(11 + 2 * ta + tb)
    ws = torch.empty(16 << 20, device='cuda')
    for M, N, K in ((128, 128, 64), (200, 136, 50), (37, 300, 1000), (256, 128, 8192), (5, 7, 3), (1000, 515, 123)):
        sa, sb = ((K, M) if ta else (M, K)), ((N, K) if tb else (K, N))
        A = (rng.standard_normal(sa) * np.exp(2 * rng.standard_normal(sa))).astype(np.float32)
        B = (rng.standard_normal(sb) * np.exp(2 * rng.standard_normal(sb))).astype(np.float32)
        C0 = rng.standard_normal((M, N + 3)).astype(np.float32)
        bias = rng.standard_normal(N).astype(np.float32)
        dA, dB, db = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda(), torch.from_numpy(bias).cuda()
        opA, opB = (A.T if ta else A).astype(np.float64), (B.T if tb else B).astype(np.float64)
        ref = 0.5 * (opA @ opB) + C0[:, :N] + bias
        scale = 0.5 * (np.abs(opA) @ np.abs(opB)) + np.abs(C0[:, :N]) + np.abs(bias)
        errs = []
        for split in (True, False):
            dC = torch.from_numpy(C0).cuda()
            args = (ta, tb, M, N, K, 0.5, dA.data_ptr(), A.shape[1], dB.data_ptr(), B.shape[1], 1.0, dC.data_ptr(), N + 3,
                    db.data_ptr(), ws.data_ptr(), ws.numel() * 4, torch.cuda.current_stream().cuda_stream)
            rc = L.asr_gemm_bf16x3_ws(terms, *args) if split else L.asr_gemm_f32_ws(*args)
            assert rc == 0
            got = dC.cpu().numpy()
            assert (got[:, N:] == C0[:, N:]).all()
            errs.append(float(np.max(np.abs(got[:, :N] - ref) / scale)))
        print('M=%d N=%d K=%d ta=%d tb=%d terms=%d: split %.2e, fp32 MFMA %.2e' % (M, N, K, ta, tb, terms, *errs))
        assert errs[0] < 1.5e-6, (M, N, K, errs)
        assert errs[0] < 1.5 * errs[1] + 1e-7, (M, N, K, errs)
    assert L.asr_gemm_bf16x3_ws(5, 0, 0, 8, 8, 8, 1.0, dA.data_ptr(), 8, dB.data_ptr(), 8, 0.0, dC.data_ptr(), 8, None,
                                None, 0, torch.cuda.current_stream().cuda_stream) == -1


@pytest.mark.parametrize('ta,tb', [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_two_part_mode_error_bound(ta, tb):
    """asr_gemm_bf16x3_ws(terms = 3): operands as TWO bf16 parts (16 significand bits), products hh + hm + mh.  Element by
    element against float64 the error stays below 2^-15 |A|.|B| (each operand's dropped part is below 2^-17 of it in
    round-to-nearest, the dropped m*m product below 2^-18; 2^-15 leaves the accumulation its share), 100x tighter than the
    plain bf16 mode's 2^-8 and ~250x looser than the default's."""
    from asr_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(21 + 2 * ta + tb)
    ws = torch.empty(16 << 20, device='cuda')
    for M, N, K in ((128, 128, 64), (200, 136, 50), (37, 300, 1000), (256, 128, 8192), (5, 7, 3), (1000, 515, 123)):
        sa, sb = ((K, M) if ta else (M, K)), ((N, K) if tb else (K, N))
        A = (rng.standard_normal(sa) * np.exp(2 * rng.standard_normal(sa))).astype(np.float32)
        B = (rng.standard_normal(sb) * np.exp(2 * rng.standard_normal(sb))).astype(np.float32)
        dA, dB = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()
        dC = torch.zeros(M, N, device='cuda')
        opA, opB = (A.T if ta else A).astype(np.float64), (B.T if tb else B).astype(np.float64)
        ref = opA @ opB
        scale = np.abs(opA) @ np.abs(opB)
        rc = L.asr_gemm_bf16x3_ws(3, ta, tb, M, N, K, 1.0, dA.data_ptr(), A.shape[1], dB.data_ptr(), B.shape[1], 0.0,
                                  dC.data_ptr(), N, None, ws.data_ptr(), ws.numel() * 4, torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        err = float(np.max(np.abs(dC.cpu().numpy() - ref) / scale))
        print('M=%d N=%d K=%d ta=%d tb=%d two-part: %.2e (2^-15 = %.2e)' % (M, N, K, ta, tb, err, 2.0 ** -15))
        assert err < 2.0 ** -15, (M, N, K, err)


def test_ln_rows_and_colreduce():
    from asr_amd import ops
    rng = np.random.default_rng(1)
    rows, n, B = 45, 72, 5
    x = rng.standard_normal((rows, n)).astype(np.float32) * 2 + 0.5
    al = rng.standard_normal((B, n)).astype(np.float32); pre = rng.standard_normal((B, n)).astype(np.float32)
    bias = rng.standard_normal(n).astype(np.float32)
    dx = torch.from_numpy(x).cuda(); y = torch.empty_like(dx); xh = torch.empty_like(dx)
    rs = torch.empty(rows, device='cuda')
    ops.ln_rows_fwd(dx, n, torch.from_numpy(al).cuda(), y, pre=torch.from_numpy(pre).cuda(), pmod=B, amod=B,
                    bias=torch.from_numpy(bias).cuda(), xhat=xh, rstd=rs)
    ridx = np.arange(rows) % B
    yo, cache = O.ln_fwd(x.astype(np.float64) + pre[ridx], al[ridx].astype(np.float64))
    assert relerr(y.cpu().numpy(), yo + bias) < 1e-5
    dy = rng.standard_normal((rows, n)).astype(np.float32)
    ddx = torch.empty_like(dx)
    ops.ln_rows_bwd(torch.from_numpy(dy).cuda(), n, xh, rs, torch.from_numpy(al).cuda(), ddx, amod=B)
    dxo, dalo, _ = O.ln_bwd(dy.astype(np.float64), al[ridx].astype(np.float64), cache)
    assert relerr(ddx.cpu().numpy(), dxo) < 1e-5
    out = torch.empty(B, n, device='cuda')
    ops.colreduce(torch.from_numpy(dy).cuda(), out, rows, n, y=xh, gmod=B)
    ref = np.zeros((B, n))
    np.add.at(ref, ridx, dalo)
    assert relerr(out.cpu().numpy(), ref) < 1e-5


def test_colreduce_ragged_groups():
    """rows % gmod != 0 with floor(rows/gmod) a multiple of the 32 row splits: the fullest groups hold
    one row more than the split size was once derived from (their last row was dropped)."""
    from asr_amd import ops
    rng = np.random.default_rng(5)
    for rows, gmod, n in ((32 * 3 + 1, 3, 70), (64 * 4 + 3, 4, 9), (33 * 5 + 1, 5, 130)):
        x = rng.standard_normal((rows, n)).astype(np.float32)
        out = torch.empty(gmod, n, device='cuda')
        ops.colreduce(torch.from_numpy(x).cuda(), out, rows, n, gmod=gmod)
        ref = np.zeros((gmod, n))
        np.add.at(ref, np.arange(rows) % gmod, x.astype(np.float64))
        assert relerr(out.cpu().numpy(), ref) < 1e-5, (rows, gmod)


def test_dropout_mask_is_philox_and_inverted():
    """asr_dropout against the oracle's numpy Philox4x32-10 (pinned to the Random123 known answers in
    tests/test_golden.py): bit-identical masks, vectorised and ragged widths, strided rows; keep rate and the
    1/(1-p) scaling checked statistically."""
    from asr_amd import ops
    rng = np.random.default_rng(11)
    for rows, n, ld, p, seed, sid in ((257, 64, 64, 0.2, 0x123456789ABCDEF, 5), (33, 37, 40, 0.5, 7, (3 << 20) | 2),
                                      (64, 123, 123, 0.2, 2 ** 63 + 11, 2 ** 40 + 9), (5, 8, 12, 0.0, 1, 1)):
        x = rng.standard_normal((rows, ld)).astype(np.float32)
        dx = torch.from_numpy(x).cuda()
        y = torch.full((rows, ld), -7.0, device='cuda')
        ops.dropout(dx[:, :n], y[:, :n], rows, n, p, seed, sid)
        mk = O.dropout_mask(rows, n, p, seed, sid)
        got = y.cpu().numpy()
        assert np.array_equal(got[:, :n], x[:, :n] * mk), (rows, n)
        assert (got[:, n:] == -7.0).all()
        ops.dropout(dx[:, :n], dx[:, :n], rows, n, p, seed, sid)          # in place = what the backward does
        assert np.array_equal(dx.cpu().numpy()[:, :n], got[:, :n])
    ones = torch.ones(4096, 512, device='cuda')
    m = ops.dropout(ones, torch.empty_like(ones), 4096, 512, 0.2, 99, 1).cpu().numpy()
    assert set(np.unique(m).tolist()) == {0.0, np.float32(1.25)}
    keep = (m > 0).mean()
    assert abs(keep - 0.8) < 4 * np.sqrt(0.16 / m.size) and abs(m.mean() - 1.0) < 2e-3
    assert abs((m[:, ::4] > 0).mean() - 0.8) < 2e-3 and abs((m[::2] > 0).mean() - 0.8) < 2e-3
    m2 = ops.dropout(ones, torch.empty_like(ones), 4096, 512, 0.2, 99, 2).cpu().numpy()
    assert abs(((m > 0) == (m2 > 0)).mean() - (0.64 + 0.04)) < 2e-3        # independent streams


def test_ln_rows_per_gate():
    """groups=4 (gate_ln / sample_gate_ln): the kernel works on gate-interleaved columns (col = unit*4 +
    gate), the oracle on the reference's gate-major blocks."""
    from asr_amd import ops
    rng = np.random.default_rng(3)
    rows, N, B = 37, 24, 4
    n = 4 * N
    to_il = lambda a: a.reshape(a.shape[0], 4, N).transpose(0, 2, 1).reshape(a.shape[0], n)   # gate-major -> interleaved
    x = rng.standard_normal((rows, n)) * 2 + 0.3
    al = rng.standard_normal((B, n)); bt = rng.standard_normal((B, n))
    ridx = np.arange(rows) % B
    yo, cache = O.ln_fwd(x, al[ridx], bt[ridx], groups=4)
    f = lambda a: torch.from_numpy(np.ascontiguousarray(to_il(a), dtype=np.float32)).cuda()
    dx = f(x); y = torch.empty_like(dx); xh = torch.empty_like(dx); rs = torch.empty(rows, 4, device='cuda')
    ops.ln_rows_fwd(dx, n, f(al), y, beta=f(bt), amod=B, xhat=xh, rstd=rs, groups=4)
    assert relerr(y.cpu().numpy(), to_il(yo)) < 1e-5
    assert relerr(1.0 / rs.cpu().numpy(), cache[1][:, :, 0]) < 1e-5
    dy = rng.standard_normal((rows, n))
    dxo, _, _ = O.ln_bwd(dy, al[ridx], cache)
    ddx = torch.empty_like(dx)
    ops.ln_rows_bwd(f(dy), n, xh, rs, f(al), ddx, amod=B, groups=4)
    assert relerr(ddx.cpu().numpy(), to_il(dxo)) < 1e-5


def test_seq_pool_kernels():
    from asr_amd import ops
    rng = np.random.default_rng(4)
    B, T, F = 5, 11, 70
    lens = np.array([11, 1, 7, 3, 9])
    mask = (np.arange(T)[None] < lens[:, None]).astype(np.float32)
    for mode in (0, 1):
        a = rng.standard_normal((B, T, F)).astype(np.float32)
        x = np.tanh(a) if mode == 0 else np.maximum(a, 0)
        if mode == 1:
            x[2, 3] = x[2, 5]                                   # ties: every tied frame gets the gradient
        ds = rng.standard_normal((B, F)).astype(np.float32)
        if mode == 0:
            s_ref = (x * mask[:, :, None]).sum(1) / mask.sum(1, keepdims=True)
            dx_ref = ds[:, None, :] / mask.sum(1)[:, None, None] * mask[:, :, None] * (1 - x ** 2)
        else:
            s_ref = x.max(1)
            dx_ref = ds[:, None, :] * (x == s_ref[:, None, :]) * (x > 0)
        for layout in ('batch_major', 'time_major'):
            xm = x if layout == 'batch_major' else np.ascontiguousarray(x.transpose(1, 0, 2))
            sb, st = (T * F, F) if layout == 'batch_major' else (F, B * F)
            xt = torch.from_numpy(xm).cuda(); out = torch.empty(B, F, device='cuda')
            ops.seq_pool_fwd(xt, sb, st, torch.from_numpy(mask).cuda(), B, T, F, mode, out)
            assert relerr(out.cpu().numpy(), s_ref) < 1e-6
            dxt = torch.empty_like(xt)
            ops.seq_pool_bwd(torch.from_numpy(ds).cuda(), out, xt, sb, st, torch.from_numpy(mask).cuda(), B, T, F,
                             mode, dxt)
            got = dxt.cpu().numpy() if layout == 'batch_major' else dxt.cpu().numpy().transpose(1, 0, 2)
            assert relerr(got, dx_ref) < 1e-6


@pytest.mark.parametrize('C', [3436, 61, 6, 4174, 1024, 5000, 7001, 4097])
def test_softmax_ce_kernel(C):
    """Both kernels (row in registers: C % 4 == 0 and C <= 6144; three-pass otherwise), every output, ties in
    the argmax (first occurrence, like numpy / Theano), probabilities and log-posteriors."""
    from asr_amd import ops
    rng = np.random.default_rng(2)
    B, T = 4, 9
    lg = (rng.standard_normal((T * B, C)) * 4).astype(np.float32)
    lg[3, C // 2] = lg[3, C - 1] = lg[3].max() + 1.0          # a tie: the first one wins
    y = rng.integers(0, C, (B, T)).astype(np.int32)
    m = (rng.random((B, T)) > 0.3).astype(np.float32)
    lgb = lg.reshape(T, B, C).transpose(1, 0, 2).astype(np.float64)
    cost, s, p, dlo = O.softmax_ce(lgb, y, m.astype(np.float64))
    for mode in (1, 2):
        dl = torch.empty(T * B, C, device='cuda'); rce = torch.empty(T * B, device='cuda')
        rok = torch.empty(T * B, device='cuda'); pred = torch.empty(B, T, dtype=torch.int32, device='cuda')
        out = torch.empty(B, T, C, device='cuda')
        ops.softmax_ce(torch.from_numpy(lg).cuda(), C, B, T, target=torch.from_numpy(y).cuda(),
                       tmask=torch.from_numpy(m).cuda(), scale=1.0 / B, dlogits=dl, row_ce=rce, row_ok=rok,
                       pred=pred, out_mode=mode, out=out)
        assert abs(rce.sum().item() - s) / s < 1e-5
        assert (pred.cpu().numpy() == p).all() and pred.cpu().numpy()[3 % B, 3 // B] == C // 2
        assert abs(rok.sum().item() - ((p == y) * m).sum()) < 1e-6
        assert relerr(dl.cpu().numpy().reshape(T, B, C).transpose(1, 0, 2), dlo) < 1e-5
        ref = O.log_softmax(lgb)
        if mode == 1:
            assert np.max(np.abs(out.cpu().numpy() - ref)) < 2e-5
        else:
            assert relerr(out.cpu().numpy(), np.exp(ref)) < 1e-5


def test_lstm_pack_modes_against_their_definitions():
    """asr_lstm_pack_w modes 1 and 2 (asr_amd/csrc/scan_lstm.hip header comments), N = 128."""
    from asr_amd import _lib, ops
    N = 128
    KS = N // 4; KT = N // 16; TW = KS // 32
    rng = np.random.default_rng(5)
    W = rng.standard_normal((N, 4 * N)).astype(np.float32)
    Wd = torch.from_numpy(W).cuda()
    L = _lib.lib()
    for mode in (1, 2):
        n = L.asr_lstm_pack_floats(N, mode)
        out = torch.zeros(n, device='cuda')
        ops.lstm_pack(Wd, N, mode, out)
        got = out.cpu().numpy()
        if mode == 1:
            tile, kt, l, e = np.meshgrid(np.arange(KS), np.arange(KT), np.arange(64), np.arange(4), indexing='ij')
            want = W[kt * 16 + (l & 15), tile * 16 + 4 * e + (l >> 4)]
        else:
            j, w, i, t, l, e = np.meshgrid(np.arange(KS), np.arange(4), np.arange(TW), np.arange(2), np.arange(64),
                                           np.arange(4), indexing='ij')
            ub, kr = j // 8, j % 8
            want = W[ub * 32 + t * 16 + (l & 15), 16 * (kr * (KS // 8) + w * TW + i) + 4 * (l >> 4) + e]
        assert n == want.size
        assert np.array_equal(got, want.reshape(-1))
    assert L.asr_lstm_pack_w(Wd.data_ptr(), 8, 2, out.data_ptr(), None) == -1          # mode 2 needs N % 128 == 0


@pytest.mark.parametrize('kind,name', [(0, 'sigmoid'), (1, 'tanh'), (2, 'exp')])
def test_activation_accuracy(kind, name):
    """The library's inline sigmoid / tanh / exp (asr_amd/csrc/common.h: every recurrence and the softmax use them
    for T.nnet.sigmoid / T.tanh / T.exp, libs/lhuclstm_layers.py:93-108) against float64 over every 61st float32 bit
    pattern (70 M arguments, both signs, all exponents): libm-grade = at most 3.5 ulp of the true value wherever that
    value is a normal float32, tanh with RELATIVE accuracy near 0 (the round-2 forms were absolute there and lost
    |x| * 2^-24 in the exponent's argument).  The measured maxima are written to gpurun_out/act_accuracy.jsonl."""
    import json, os
    from asr_amd import ops
    worst, worst_x = 0.0, 0.0
    step = 61
    for lo in range(0, 1 << 32, 1 << 28):
        bits = torch.arange(lo, lo + (1 << 28), step, dtype=torch.int64, device='cuda').to(torch.int32)
        x = bits.view(torch.float32)
        y = torch.empty_like(x)
        ops.act_eval(kind, x, y)
        xs, ys = x.cpu().numpy(), y.cpu().numpy()
        keep = np.isfinite(xs)
        if kind == 2:
            keep &= np.abs(xs) < 87.0
        xs, ys = xs[keep], ys[keep]
        if xs.size == 0:
            continue
        xd = xs.astype(np.float64)
        with np.errstate(over='ignore'):
            ref = 1.0 / (1.0 + np.exp(-xd)) if kind == 0 else np.tanh(xd) if kind == 1 else np.exp(xd)
        normal = np.abs(ref) >= 1.2e-38
        ulp = np.spacing(np.abs(ref).astype(np.float32)).astype(np.float64)
        err = np.abs(ys.astype(np.float64) - ref) / ulp
        err = np.where(normal, err, 0.0)
        assert np.isfinite(ys).all()
        k = int(np.argmax(err))
        if err[k] > worst:
            worst, worst_x = float(err[k]), float(xs[k])
    nanv = torch.full((4,), float('nan'), device='cuda')
    out = torch.empty_like(nanv)
    ops.act_eval(kind, nanv, out)
    assert torch.isnan(out).all()                        # a diverged run must stay visible
    inf = torch.tensor([float('inf'), -float('inf')], device='cuda')
    out = torch.empty_like(inf)
    ops.act_eval(kind, inf, out)
    want = {0: [1.0, 0.0], 1: [1.0, -1.0], 2: [float('inf'), 0.0]}[kind]
    assert out.cpu().tolist() == want
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    if os.path.isdir(d):
        with open(os.path.join(d, 'act_accuracy.jsonl'), 'a') as f:
            f.write(json.dumps(dict(fn=name, max_ulp=worst, at=worst_x, arguments=(1 << 32) // step)) + '\n')
    print('%s: max error %.2f ulp at x = %r' % (name, worst, worst_x))
    assert worst <= 3.5, (name, worst, worst_x)


@pytest.mark.parametrize('ta,tb', [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_presplit_bit_identical(ta, tb):
    """asr_gemm_presplit_ws (operands split once by asr_split_bf16x3) gives the bits of asr_gemm_bf16x3_ws with 6
    products for either or both operands pre-split: interior and edge tiles, K tails, widths that are not multiples of
    8, split-K, alpha / beta / bias, and sub-matrices taken by rows and by columns (a multiple of 8) of the planes."""
    from asr_amd import _lib, ops
    L = _lib.lib()
    rng = np.random.default_rng(31 + 2 * ta + tb)
    ws = torch.empty(16 << 20, device='cuda')
    s = torch.cuda.current_stream().cuda_stream
    for M, N, K in ((128, 128, 64), (256, 384, 160), (200, 136, 50), (37, 300, 1003), (256, 128, 8192), (5, 7, 3),
                    (1000, 515, 123)):
        sa, sb = ((K, M) if ta else (M, K)), ((N, K) if tb else (K, N))
        # operands live inside larger matrices: 3 extra rows on top, 8 extra columns on the left
        Af = torch.from_numpy((rng.standard_normal((sa[0] + 3, sa[1] + 8)) * np.exp(rng.standard_normal((sa[0] + 3, sa[1] + 8)))).astype(np.float32)).cuda()
        Bf = torch.from_numpy(rng.standard_normal((sb[0] + 3, sb[1] + 8)).astype(np.float32)).cuda()
        A, B = Af[3:, 8:], Bf[3:, 8:]
        As, Bs = ops.split3(Af)[3:, 8:], ops.split3(Bf)[3:, 8:]
        C0 = torch.from_numpy(rng.standard_normal((M, N)).astype(np.float32)).cuda()
        bias = torch.from_numpy(rng.standard_normal(N).astype(np.float32)).cuda()
        want = C0.clone()
        rc = L.asr_gemm_bf16x3_ws(6, ta, tb, M, N, K, 0.5, A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0), 1.0,
                                  want.data_ptr(), N, bias.data_ptr(), ws.data_ptr(), ws.numel() * 4, s)
        assert rc == 0
        for pa, pb in ((1, 1), (1, 0), (0, 1)):
            got = C0.clone()
            rc = L.asr_gemm_presplit_ws(pa, pb, ta, tb, M, N, K, 0.5,
                                        As.ptr() if pa else A.data_ptr(), As.ld if pa else A.stride(0), As.plane_stride if pa else 0,
                                        Bs.ptr() if pb else B.data_ptr(), Bs.ld if pb else B.stride(0), Bs.plane_stride if pb else 0,
                                        1.0, got.data_ptr(), N, bias.data_ptr(), ws.data_ptr(), ws.numel() * 4, s)
            assert rc == 0
            torch.cuda.synchronize()
            assert torch.equal(got, want), (M, N, K, pa, pb, (got - want).abs().max().item())
    # the parts add up to the value: h + m + l == x wherever x is a normal float
    X = torch.from_numpy((rng.standard_normal((50, 37)) * np.exp(3 * rng.standard_normal((50, 37)))).astype(np.float32)).cuda()
    sp = ops.split3(X)
    parts = (sp.planes.view(torch.int16).to(torch.int32) << 16).view(torch.float32)[:, :, :37].cpu().numpy().astype(np.float64)
    assert sp.ld == 40 and (sp.planes[:, :, 37:] == 0).all()
    assert np.max(np.abs(parts.sum(0) - X.cpu().numpy()) / np.abs(X.cpu().numpy())) < 2.0 ** -22
