"""Thin torch-tensor wrappers over the C-ABI (include/asr_amd.h).  torch is used for device memory
and streams only; every arithmetic op below is a HIP kernel in libasr_amd.so."""
import ctypes as C
import os
import torch

from . import _lib
from ._lib import check


def _p(t):
    return None if t is None else t.data_ptr()


def _s():
    return torch.cuda.current_stream().cuda_stream


def _chk2d(t):
    assert t.dim() == 2 and (t.numel() == 0 or t.stride(1) == 1) and t.dtype == torch.float32, \
        (t.shape, t.stride(), t.dtype)


_gemm_ws = {}
# How the large products (at least SPLIT_MIN_WORK multiply-adds: input projections, output layer, dX, weight-gradient
# contractions) are multiplied; set through engine.Network.set_precision, process-wide:
#   'f32x3' fp32-grade on the bf16 matrix cores: operands split into three bf16 parts on the way into LDS, 6 partial
#           products accumulated in fp32 (asr_gemm_bf16x3_ws; error against float64 at or below the fp32 MFMA kernel's)
#   'f32'   v_mfma_f32_32x32x2_f32 throughout (asr_gemm_f32_ws)
#   'bf16'  the optional bf16 mode: operands rounded to bf16, fp32 accumulate (keeping the weight-gradient
#           contractions in fp32 changed no measured error at cfg3 and cost 4 ms)
# Small products always take the fp32 MFMA kernel.
GEMM_TERMS = {'f32': 0, 'f32x3': 6, 'bf16x2': 3, 'bf16': 1}
DEFAULT_PRECISION = os.environ.get('ASR_PRECISION', 'f32x3')
PRECISION = DEFAULT_PRECISION
SPLIT_MIN_WORK = 1 << 24


class Split3(object):
    """An fp32 matrix held as its three bf16 parts (asr_split_bf16x3): planes[p] is (rows, ld) bf16 with the matrix in
    the first `cols` columns.  Behaves like the 2-D tensor it stands for where gemm() needs it: shape, row slices,
    column slices at multiples of 8.  Made by split3(); passed to gemm() in place of the tensor."""
    def __init__(self, planes, r0, r1, c0, c1):
        self.planes = planes                     # (3, R, ld) int16 storage
        self.r0, self.r1, self.c0, self.c1 = r0, r1, c0, c1
        self.shape = (r1 - r0, c1 - c0)

    @property
    def ld(self):
        return self.planes.shape[2]

    @property
    def plane_stride(self):
        return self.planes.stride(0)

    def ptr(self):
        return self.planes.data_ptr() + 2 * (self.r0 * self.ld + self.c0)

    def __getitem__(self, key):
        if not isinstance(key, tuple):
            key = (key, slice(None))
        rs, cs = key
        nr, ncol = self.shape
        a, b, st = rs.indices(nr)
        c, d, st2 = cs.indices(ncol)
        assert st == 1 and st2 == 1 and c % 8 == 0, 'Split3: unit-stride slices, columns from a multiple of 8'
        return Split3(self.planes, self.r0 + a, self.r0 + b, self.c0 + c, self.c0 + d)


def split3(X, out=None):
    """The three bf16 parts of the 2-D fp32 view X, once, for every product that reads it."""
    _chk2d(X)
    R, Cc = X.shape
    L = _lib.lib()
    ld = L.asr_split_ld(Cc)
    if out is None or out.numel() < 3 * R * ld:
        out = torch.empty(3 * R * ld, dtype=torch.int16, device=X.device)
    planes = out[:3 * R * ld].view(3, R, ld)
    check(L.asr_split_bf16x3(_p(X), max(X.stride(0), 1), R, Cc, _p(planes), ld, R * ld, _s()), 'asr_split_bf16x3')
    return Split3(planes, 0, R, 0, Cc)


def gemm(A, B, Cm, ta=False, tb=False, alpha=1.0, beta=0.0, bias=None):
    """Cm = alpha*op(A)*op(B) + beta*Cm (+bias).  2-D float32 views with unit inner stride; A and / or B may be a
    Split3 (pre-split operand) when the product runs in the default precision."""
    pa, pb = isinstance(A, Split3), isinstance(B, Split3)
    if not pa:
        _chk2d(A)
    if not pb:
        _chk2d(B)
    _chk2d(Cm)
    M, K = (A.shape[1], A.shape[0]) if ta else (A.shape[0], A.shape[1])
    K2, N = (B.shape[1], B.shape[0]) if tb else (B.shape[0], B.shape[1])
    assert K == K2 and tuple(Cm.shape) == (M, N), (A.shape, B.shape, Cm.shape, ta, tb)
    key = (Cm.device, _s())                       # one split-K scratch per stream: launches on
    ws = _gemm_ws.get(key)                        # different streams may overlap
    if ws is None:
        ws = torch.empty(32 * 1024 * 1024, dtype=torch.float32, device=Cm.device)     # split-K partials
        _gemm_ws[key] = ws
    L = _lib.lib()
    if pa or pb:
        assert PRECISION == 'f32x3', 'pre-split operands belong to the default precision'
        check(L.asr_gemm_presplit_ws(int(pa), int(pb), int(ta), int(tb), M, N, K, alpha,
                                     A.ptr() if pa else _p(A), A.ld if pa else max(A.stride(0), 1),
                                     A.plane_stride if pa else 0,
                                     B.ptr() if pb else _p(B), B.ld if pb else max(B.stride(0), 1),
                                     B.plane_stride if pb else 0, beta, _p(Cm), max(Cm.stride(0), 1), _p(bias),
                                     _p(ws), ws.numel() * 4, _s()), 'asr_gemm_presplit_ws')
        return Cm
    terms = GEMM_TERMS[PRECISION] if M * N * K >= SPLIT_MIN_WORK else 0
    args = (int(ta), int(tb), M, N, K, alpha, _p(A), max(A.stride(0), 1), _p(B), max(B.stride(0), 1), beta,
            _p(Cm), max(Cm.stride(0), 1), _p(bias), _p(ws), ws.numel() * 4, _s())
    if terms:
        check(L.asr_gemm_bf16x3_ws(terms, *args), 'asr_gemm_bf16x3_ws')
    else:
        check(L.asr_gemm_f32_ws(*args), 'asr_gemm_f32_ws')
    return Cm


def swap01(src, dst, A, B, D, ld_src=None, ld_dst=None):
    check(_lib.lib().asr_swap01(_p(src), _p(dst), A, B, D, ld_src or D, ld_dst or D, _s()), 'asr_swap01')
    return dst


def ln_rows_fwd(x, n, alpha, y, pre=None, pmod=1, beta=None, amod=0, bias=None, xhat=None, rstd=None,
                groups=1):
    rows = x.numel() // n
    check(_lib.lib().asr_ln_rows_fwd(_p(x), rows, n, _p(pre), pmod, _p(alpha), _p(beta), amod, _p(bias),
                                     _p(y), _p(xhat), _p(rstd), groups, _s()), 'asr_ln_rows_fwd')
    return y


def ln_rows_bwd(dy, n, xhat, rstd, alpha, dx, amod=0, groups=1):
    rows = dy.numel() // n
    check(_lib.lib().asr_ln_rows_bwd(_p(dy), rows, n, _p(xhat), _p(rstd), _p(alpha), amod, _p(dx), groups, _s()),
          'asr_ln_rows_bwd')
    return dx


_cr_ws = {}


def colreduce(x, out, rows, n, y=None, gmod=1, accumulate=False, ldx=None, ldy=None):
    """out[g,c] (+)= sum_{r % gmod == g} x[r,c] * (y[r,c] if y is given)."""
    L = _lib.lib()
    need = L.asr_colreduce_ws_bytes(n, gmod)
    key = (x.device, _s())
    ws = _cr_ws.get(key)
    if ws is None or ws.numel() * 4 < need:
        ws = torch.empty(need // 4 + 16, dtype=torch.float32, device=x.device)
        _cr_ws[key] = ws
    check(L.asr_colreduce(_p(x), ldx or n, _p(y), ldy or n, rows, n, gmod, _p(out), int(accumulate),
                          _p(ws), ws.numel() * 4, _s()), 'asr_colreduce')
    return out


def bias_act(x, bias, rows, n, act):
    check(_lib.lib().asr_bias_act(_p(x), _p(bias), rows, n, act, _s()), 'asr_bias_act')
    return x


def act_bwd(dy, y, rows, n, act):
    check(_lib.lib().asr_act_bwd(_p(dy), _p(y), rows, n, act, _s()), 'asr_act_bwd')
    return dy


def act_eval(kind, x, y):
    """test probe: the library's sigmoid (0) / tanh (1) / exp (2) applied elementwise"""
    check(_lib.lib().asr_act_eval(kind, _p(x), _p(y), x.numel(), _s()), 'asr_act_eval')
    return y


def rowscale(x, ldx, s, B, relu, y, ldy, rows, n):
    check(_lib.lib().asr_rowscale(_p(x), ldx, _p(s), B, int(relu), _p(y), ldy, rows, n, _s()),
          'asr_rowscale')
    return y


def dropout(x, y, rows, n, p, seed, stream_id):
    """y = x * keep/(1-p) with the Philox mask of (seed, stream_id); in place when y is x."""
    check(_lib.lib().asr_dropout(_p(x), x.stride(0), _p(y), y.stride(0), rows, n, float(p), int(seed),
                                 int(stream_id), _s()), 'asr_dropout')
    return y


def var_costs(st, lens, tpad, Bg, F, T, w_inter, w_intra, b0, Bloc, coef, costs, accumulate):
    check(_lib.lib().asr_var_costs(_p(st), _p(lens), _p(tpad), Bg, F, T, w_inter, w_intra, b0, Bloc, _p(coef), _p(costs),
                                   int(accumulate), _s()), 'asr_var_costs')


def affine_rows(x, ldx, a, c0, c1, mask, B, y, ldy, rows, n):
    check(_lib.lib().asr_affine_rows(_p(x), ldx, _p(a), _p(c0), _p(c1), _p(mask), mask.stride(0), mask.stride(1), B,
                                     _p(y), ldy, rows, n, _s()), 'asr_affine_rows')
    return y


def masked_mean(x, mask, B, T, F, out):
    check(_lib.lib().asr_masked_mean(_p(x), _p(mask), B, T, F, _p(out), _s()), 'asr_masked_mean')
    return out


def masked_mean_tanh_bwd(ds, mask, pre, B, T, F, dpre):
    check(_lib.lib().asr_masked_mean_tanh_bwd(_p(ds), _p(mask), _p(pre), B, T, F, _p(dpre), _s()),
          'asr_masked_mean_tanh_bwd')
    return dpre


def seq_pool_fwd(x, sb, st, mask, B, T, F, mode, out):
    check(_lib.lib().asr_seq_pool_fwd(_p(x), sb, st, _p(mask), B, T, F, mode, _p(out), _s()), 'asr_seq_pool_fwd')
    return out


def seq_pool_bwd(ds, s, x, sb, st, mask, B, T, F, mode, dx):
    check(_lib.lib().asr_seq_pool_bwd(_p(ds), _p(s), _p(x), sb, st, _p(mask), B, T, F, mode, _p(dx), _s()),
          'asr_seq_pool_bwd')
    return dx


def softmax_ce(logits, C_, B, T, target=None, tmask=None, scale=0.0, dlogits=None, row_ce=None,
               row_ok=None, pred=None, out_mode=0, out=None):
    """target/tmask are the caller's batch-major (B,T) arrays (strides taken from the tensors)."""
    tsb, tst = (target.stride(0), target.stride(1)) if target is not None else (0, 0)
    msb, mst = (tmask.stride(0), tmask.stride(1)) if tmask is not None else (0, 0)
    check(_lib.lib().asr_softmax_ce(_p(logits), C_, B, T, _p(target), tsb, tst, _p(tmask), msb, mst,
                                    scale, _p(dlogits), _p(row_ce), _p(row_ok), _p(pred), out_mode,
                                    _p(out), _s()), 'asr_softmax_ce')


def reduce_sum(x, n, out, do_sqrt=False):
    check(_lib.lib().asr_reduce_sum(_p(x), n, _p(out), int(do_sqrt), _s()), 'asr_reduce_sum')
    return out


def adam_step(p, g, m, v, a_t, b1, b2, eps, norm_out, want_sqrt, ws):
    check(_lib.lib().asr_adam_step(_p(p), _p(g), _p(m), _p(v), p.numel(), a_t, b1, b2, eps, _p(norm_out),
                                   int(want_sqrt), _p(ws), _s()), 'asr_adam_step')


def opt_step(kind, p, g, s1, s2, lr, h1, h2, eps, gscale=None):
    check(_lib.lib().asr_opt_step(kind, _p(p), _p(g), _p(s1) if s1 is not None else None,
                                  _p(s2) if s2 is not None else None, p.numel(), lr, h1, h2, eps,
                                  _p(gscale) if gscale is not None else None, _s()), 'asr_opt_step')


def norm_clip_mult(norm, max_norm, eps, out):
    check(_lib.lib().asr_norm_clip_mult(_p(norm), max_norm, eps, _p(out), _s()), 'asr_norm_clip_mult')


def sqsum(g, out, want_sqrt, ws):
    check(_lib.lib().asr_sqsum(_p(g), g.numel(), _p(out), int(want_sqrt), _p(ws), _s()), 'asr_sqsum')


def lstm_pack(W, N, mode, out):
    assert W.is_contiguous() and out.numel() >= _lib.lib().asr_lstm_pack_floats(N, mode)
    check(_lib.lib().asr_lstm_pack_w(_p(W), N, mode, _p(out), _s()), 'asr_lstm_pack_w')
    return out
