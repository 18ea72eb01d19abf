// K3: fused framewise log-softmax / cross-entropy / argmax(FER) / dlogits  (HBM-bound: one pass over
//     the logits from HBM, re-reads served by L2).
//     Restates libs/comp_graph_utils.py:29-45 and scripts/hypernets/exp_prj_ln_lstm_fix.py:84-95.
// K4: Adam (libs/lasagne_libs/updates.py:587-617) fused with the squared gradient norm
//     (libs/comp_graph_utils.py:77).
#include "common.h"

struct ArgMax { float v; int i; };
__device__ __forceinline__ ArgMax amax_(ArgMax a, ArgMax b) {
    // first-occurrence tie break like numpy / Theano argmax
    if (b.v > a.v || (b.v == a.v && b.i < a.i)) return b;
    return a;
}

// rows are time-major r = t*B + b.  targets / masks are addressed with explicit strides so that the
// caller's batch-major (B,T) arrays are used as they are.
// out_mode: 0 none, 1 log-posteriors, 2 probabilities; written batch-major out[b][t][c].
__global__ __launch_bounds__(256) void softmax_ce_kernel(
    const float* __restrict__ logits, int C, int B, int T, const int* __restrict__ target, int tgt_sb,
    int tgt_st, const float* __restrict__ tmask, int msk_sb, int msk_st, float scale,
    float* __restrict__ dlogits, float* __restrict__ row_ce, float* __restrict__ row_ok,
    int* __restrict__ pred, int out_mode, float* __restrict__ out) {
    __shared__ float smf[4];
    __shared__ int smi[4];
    const size_t r = blockIdx.x;
    const int t = (int)(r / B), b = (int)(r % B);
    const float* lr = logits + r * C;
    ArgMax am{-INFINITY, 0x7fffffff};
    for (int c = threadIdx.x; c < C; c += 256) am = amax_(am, ArgMax{lr[c], c});
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ArgMax ot{__shfl_xor(am.v, o), __shfl_xor(am.i, o)};
        am = amax_(am, ot);
    }
    if ((threadIdx.x & 63) == 0) { smf[threadIdx.x >> 6] = am.v; smi[threadIdx.x >> 6] = am.i; }
    __syncthreads();
    am = ArgMax{smf[0], smi[0]};
#pragma unroll
    for (int w = 1; w < 4; ++w) am = amax_(am, ArgMax{smf[w], smi[w]});
    __syncthreads();
    float s = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) s += expf_(lr[c] - am.v);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) smf[threadIdx.x >> 6] = s;
    __syncthreads();
    s = smf[0] + smf[1] + smf[2] + smf[3];
    const float lse = logf(s);
    const int y = target ? target[(size_t)b * tgt_sb + (size_t)t * tgt_st] : -1;
    const float m = tmask ? tmask[(size_t)b * msk_sb + (size_t)t * msk_st] : 1.0f;
    if (threadIdx.x == 0) {
        if (row_ce) row_ce[r] = (y >= 0 && y < C) ? -(lr[y] - am.v - lse) * m : 0.f;
        if (row_ok) row_ok[r] = (am.i == y) ? m : 0.f;
        if (pred) pred[(size_t)b * T + t] = am.i;
    }
    if (dlogits) {
        const float g = m * scale;
        for (int c = threadIdx.x; c < C; c += 256) {
            float p = expf_(lr[c] - am.v - lse);
            dlogits[r * C + c] = (p - (c == y ? 1.f : 0.f)) * g;
        }
    }
    if (out_mode) {
        float* orow = out + ((size_t)b * T + t) * C;
        for (int c = threadIdx.x; c < C; c += 256) {
            float lp = lr[c] - am.v - lse;
            orow[c] = out_mode == 1 ? lp : expf_(lp);
        }
    }
}

// The same with the row held in registers (C a multiple of 4, C <= 1024 * NV, 16-byte aligned rows): ONE 16-byte
// read of every logit, max / argmax and the exp sum from registers, 16-byte stores of dlogits / posteriors.  The
// three-pass kernel above re-read the row from L1/L2 twice and ran at 2.3 TB/s at C = 3436 (29 % of HBM).
template <int NV>
__global__ __launch_bounds__(256) void softmax_ce_reg_kernel(
    const float* __restrict__ logits, int C, int B, int T, const int* __restrict__ target, int tgt_sb,
    int tgt_st, const float* __restrict__ tmask, int msk_sb, int msk_st, float scale,
    float* __restrict__ dlogits, float* __restrict__ row_ce, float* __restrict__ row_ok,
    int* __restrict__ pred, int out_mode, float* __restrict__ out) {
    __shared__ float smf[4];
    __shared__ int smi[4];
    const size_t r = blockIdx.x;
    const int t = (int)(r / B), b = (int)(r % B);
    const float* lr = logits + r * C;
    const int C4 = C >> 2;
    f32x4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int g = i * 256 + threadIdx.x;
        v[i] = g < C4 ? *(const f32x4*)(lr + g * 4) : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    }
    ArgMax am{-INFINITY, 0x7fffffff};
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) am = amax_(am, ArgMax{v[i][e], (i * 256 + (int)threadIdx.x) * 4 + e});
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ArgMax ot{__shfl_xor(am.v, o), __shfl_xor(am.i, o)};
        am = amax_(am, ot);
    }
    if ((threadIdx.x & 63) == 0) { smf[threadIdx.x >> 6] = am.v; smi[threadIdx.x >> 6] = am.i; }
    __syncthreads();
    am = ArgMax{smf[0], smi[0]};
#pragma unroll
    for (int w = 1; w < 4; ++w) am = amax_(am, ArgMax{smf[w], smi[w]});
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) s += expf_(v[i][e] - am.v);        // padding lanes: exp(-inf) = 0
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) smf[threadIdx.x >> 6] = s;
    __syncthreads();
    s = smf[0] + smf[1] + smf[2] + smf[3];
    const float lse = logf(s);
    const int y = target ? target[(size_t)b * tgt_sb + (size_t)t * tgt_st] : -1;
    const float m = tmask ? tmask[(size_t)b * msk_sb + (size_t)t * msk_st] : 1.0f;
    if (threadIdx.x == 0) {
        if (row_ce) row_ce[r] = (y >= 0 && y < C) ? -(lr[y] - am.v - lse) * m : 0.f;
        if (row_ok) row_ok[r] = (am.i == y) ? m : 0.f;
        if (pred) pred[(size_t)b * T + t] = am.i;
    }
    if (dlogits) {
        const float g = m * scale;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int gi = i * 256 + threadIdx.x;
            if (gi < C4) {
                f32x4 d;
#pragma unroll
                for (int e = 0; e < 4; ++e) d[e] = (expf_(v[i][e] - am.v - lse) - (gi * 4 + e == y ? 1.f : 0.f)) * g;
                *(f32x4*)(dlogits + r * C + gi * 4) = d;
            }
        }
    }
    if (out_mode) {
        float* orow = out + ((size_t)b * T + t) * C;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int gi = i * 256 + threadIdx.x;
            if (gi < C4) {
                f32x4 d;
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float lp = v[i][e] - am.v - lse; d[e] = out_mode == 1 ? lp : expf_(lp); }
                *(f32x4*)(orow + gi * 4) = d;
            }
        }
    }
}

extern "C" int asr_softmax_ce(const float* logits, int C, int B, int T, const int* target, int tgt_sb,
                              int tgt_st, const float* tmask, int msk_sb, int msk_st, float scale,
                              float* dlogits, float* row_ce, float* row_ok, int* pred, int out_mode,
                              float* out, hipStream_t stream) {
    if (C <= 0 || B < 0 || T < 0 || !logits) return ASR_ERR_ARG;
    if (out_mode && !out) return ASR_ERR_ARG;
    if ((dlogits || row_ce || row_ok) && !target) return ASR_ERR_ARG;
    size_t rows = (size_t)B * T;
    if (!rows) return ASR_OK;
    const bool al = !(C & 3) && !(((uintptr_t)logits | (uintptr_t)dlogits | (uintptr_t)out) & 15);
#define SMX_(NV_) hipLaunchKernelGGL((softmax_ce_reg_kernel<NV_>), dim3((unsigned)rows), dim3(256), 0, stream, logits, \
                                     C, B, T, target, tgt_sb, tgt_st, tmask, msk_sb, msk_st, scale, dlogits, row_ce,     \
                                     row_ok, pred, out_mode, out)
    if (al && C <= 1024) SMX_(1);
    else if (al && C <= 2048) SMX_(2);
    else if (al && C <= 4096) SMX_(4);
    else if (al && C <= 6144) SMX_(6);
    else hipLaunchKernelGGL(softmax_ce_kernel, dim3((unsigned)rows), dim3(256), 0, stream, logits, C, B, T,
                            target, tgt_sb, tgt_st, tmask, msk_sb, msk_st, scale, dlogits, row_ce, row_ok,
                            pred, out_mode, out);
#undef SMX_
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}

// deterministic single-workgroup sum (n up to a few 1e5): out[0] = sum x
__global__ __launch_bounds__(1024) void reduce_sum_kernel(const float* __restrict__ x, size_t n,
                                                          float* __restrict__ out, int do_sqrt) {
    __shared__ double sm[16];
    double acc = 0.0;
    for (size_t i = threadIdx.x; i < n; i += 1024) acc += (double)x[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int w = 0; w < 16; ++w) tot += sm[w];
        out[0] = do_sqrt ? (float)sqrt(tot) : (float)tot;
    }
}
extern "C" int asr_reduce_sum(const float* x, size_t n, float* out, int do_sqrt, hipStream_t stream) {
    if (!x || !out) return ASR_ERR_ARG;
    hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(1024), 0, stream, x, n, out, do_sqrt);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}

// ---------------------------------------------------------------- Adam + grad-norm^2 partials
#define ADAM_BLOCKS 1024
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   size_t n, float a_t, float b1, float b2, float eps,
                                                   float* __restrict__ sq_part) {
    __shared__ float sm[4];
    float sq = 0.f;
    const size_t n4 = n / 4;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4 gv = ((const f32x4*)g)[i], mv = ((f32x4*)m)[i], vv = ((f32x4*)v)[i], pv = ((f32x4*)p)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sq += gv[e] * gv[e];
            mv[e] = b1 * mv[e] + (1.f - b1) * gv[e];
            vv[e] = b2 * vv[e] + (1.f - b2) * gv[e] * gv[e];
            pv[e] = pv[e] - a_t * mv[e] / (sqrtf(vv[e]) + eps);
        }
        ((f32x4*)m)[i] = mv; ((f32x4*)v)[i] = vv; ((f32x4*)p)[i] = pv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        size_t i = n4 * 4 + threadIdx.x;
        float gv = g[i];
        sq += gv * gv;
        float mv = b1 * m[i] + (1.f - b1) * gv;
        float vv = b2 * v[i] + (1.f - b2) * gv * gv;
        m[i] = mv; v[i] = vv;
        p[i] = p[i] - a_t * mv / (sqrtf(vv) + eps);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = sq;
    __syncthreads();
    if (threadIdx.x == 0 && sq_part) sq_part[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

// grads_norm_out[0] = sqrt(sum g^2) (or the plain sum when want_sqrt == 0, the TBPTT trainer's
// output, libs/comp_graph_utils.py:111).  ws must hold ADAM_BLOCKS floats.
extern "C" int asr_adam_step(float* p, const float* g, float* m, float* v, size_t n, float a_t, float b1,
                             float b2, float eps, float* grads_norm_out, int want_sqrt, float* ws,
                             hipStream_t stream) {
    if (!p || !g || !m || !v) return ASR_ERR_ARG;
    if (grads_norm_out && !ws) return ASR_ERR_ARG;
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return ASR_ERR_ARG;
    if (n == 0) return ASR_OK;
    hipLaunchKernelGGL(adam_kernel, dim3(ADAM_BLOCKS), dim3(256), 0, stream, p, g, m, v, n, a_t, b1, b2,
                       eps, grads_norm_out ? ws : nullptr);
    ASR_CHECK_LAUNCH();
    if (grads_norm_out) {
        hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(1024), 0, stream, ws, (size_t)ADAM_BLOCKS,
                           grads_norm_out, want_sqrt);
        ASR_CHECK_LAUNCH();
    }
    return ASR_OK;
}

// sum of squares only (used when gradients are all-reduced first and for tests)
__global__ __launch_bounds__(256) void sqsum_kernel(const float* __restrict__ g, size_t n,
                                                    float* __restrict__ part) {
    __shared__ float sm[4];
    float sq = 0.f;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        sq += g[i] * g[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = sq;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}
extern "C" int asr_sqsum(const float* g, size_t n, float* out, int want_sqrt, float* ws,
                         hipStream_t stream) {
    if (!g || !out || !ws) return ASR_ERR_ARG;
    hipLaunchKernelGGL(sqsum_kernel, dim3(ADAM_BLOCKS), dim3(256), 0, stream, g, n, ws);
    ASR_CHECK_LAUNCH();
    hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(1024), 0, stream, ws, (size_t)ADAM_BLOCKS, out,
                       want_sqrt);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}

// ---------------------------------------------------------------- the other optimisers of
// libs/lasagne_libs/updates.py (sgd :64-92, apply_momentum :95-149, apply_nesterov_momentum :191-251,
// adagrad :298-363, rmsprop :366-433, adadelta :436-531, adam :534-618, adamax :621-698) as ONE
// elementwise pass over the flat parameter buffer.  `gscale` (device scalar or null) multiplies every
// gradient first: the multiplier of total_norm_constraint (:781-855), so that clipping by the global
// norm needs no host round trip.
//   kind 0 sgd        p -= lr g
//        1 momentum   v = h1 v - lr g;            p += v
//        2 nesterov   v = h1 v - lr g;            p += h1 v - lr g
//        3 adagrad    a += g^2;                   p -= lr g / sqrt(a + eps)
//        4 rmsprop    a = h1 a + (1-h1) g^2;      p -= lr g / sqrt(a + eps)
//        5 adadelta   a = h1 a + (1-h1) g^2; u = g sqrt(d + eps)/sqrt(a + eps); p -= lr u; d = h1 d + (1-h1) u^2
//        6 adamax     m = h1 m + (1-h1) g; u = max(h2 u, |g|);  p -= lr m / (u + eps)     (lr = a_t)
//        7 adam       m = h1 m + (1-h1) g; v = h2 v + (1-h2) g^2; p -= lr m / (sqrt(v) + eps) (lr = a_t)
template <int KIND>
__device__ __forceinline__ void opt_elem(float& p, float g, float& s1, float& s2, float lr, float h1, float h2,
                                         float eps) {
    if (KIND == 0) { p = p - lr * g; }
    else if (KIND == 1) { s1 = h1 * s1 - lr * g; p = p + s1; }
    else if (KIND == 2) { s1 = h1 * s1 - lr * g; p = p + (h1 * s1 - lr * g); }
    else if (KIND == 3) { s1 = s1 + g * g; p = p - lr * g / sqrtf(s1 + eps); }
    else if (KIND == 4) { s1 = h1 * s1 + (1.f - h1) * g * g; p = p - lr * g / sqrtf(s1 + eps); }
    else if (KIND == 5) {
        s1 = h1 * s1 + (1.f - h1) * g * g;
        const float u = g * sqrtf(s2 + eps) / sqrtf(s1 + eps);
        p = p - lr * u;
        s2 = h1 * s2 + (1.f - h1) * u * u;
    }
    else if (KIND == 6) { s1 = h1 * s1 + (1.f - h1) * g; s2 = fmaxf(h2 * s2, fabsf(g)); p = p - lr * s1 / (s2 + eps); }
    else { s1 = h1 * s1 + (1.f - h1) * g; s2 = h2 * s2 + (1.f - h2) * g * g; p = p - lr * s1 / (sqrtf(s2) + eps); }
}

template <int KIND>
__global__ __launch_bounds__(256) void opt_step_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ s1, float* __restrict__ s2, size_t n,
                                                       float lr, float h1, float h2, float eps,
                                                       const float* __restrict__ gscale) {
    const float gs = gscale ? gscale[0] : 1.f;
    constexpr bool U1 = KIND >= 1, U2 = KIND >= 5;
    const size_t n4 = n / 4;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4 pv = ((f32x4*)p)[i], gv = ((const f32x4*)g)[i] * gs;
        f32x4 a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
        if (U1) a = ((f32x4*)s1)[i];
        if (U2) b = ((f32x4*)s2)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float pe = pv[e], ae = a[e], be = b[e];
            opt_elem<KIND>(pe, gv[e], ae, be, lr, h1, h2, eps);
            pv[e] = pe; a[e] = ae; b[e] = be;
        }
        ((f32x4*)p)[i] = pv;
        if (U1) ((f32x4*)s1)[i] = a;
        if (U2) ((f32x4*)s2)[i] = b;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t i = n4 * 4 + threadIdx.x;
        float pe = p[i], ae = U1 ? s1[i] : 0.f, be = U2 ? s2[i] : 0.f;
        opt_elem<KIND>(pe, g[i] * gs, ae, be, lr, h1, h2, eps);
        p[i] = pe;
        if (U1) s1[i] = ae;
        if (U2) s2[i] = be;
    }
}

extern "C" int asr_opt_step(int kind, float* p, const float* g, float* s1, float* s2, size_t n, float lr, float h1,
                            float h2, float eps, const float* gscale, hipStream_t stream) {
    if (kind < 0 || kind > 7 || !p || !g) return ASR_ERR_ARG;
    if (kind >= 1 && !s1) return ASR_ERR_ARG;
    if (kind >= 5 && !s2) return ASR_ERR_ARG;
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)s1 | (uintptr_t)s2) & 15) return ASR_ERR_ARG;
    if (n == 0) return ASR_OK;
#define OPT_(K) hipLaunchKernelGGL(opt_step_kernel<K>, dim3(ADAM_BLOCKS), dim3(256), 0, stream, p, g, s1, s2, n, \
                                   lr, h1, h2, eps, gscale)
    switch (kind) {
        case 0: OPT_(0); break; case 1: OPT_(1); break; case 2: OPT_(2); break; case 3: OPT_(3); break;
        case 4: OPT_(4); break; case 5: OPT_(5); break; case 6: OPT_(6); break; default: OPT_(7); break;
    }
#undef OPT_
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}

// total_norm_constraint's multiplier (libs/lasagne_libs/updates.py:850-853):
// mult = clip(norm, 0, max_norm) / (eps + norm), from the device scalar asr_sqsum(want_sqrt=1) wrote.
__global__ void norm_mult_kernel(const float* __restrict__ norm, float max_norm, float eps, float* __restrict__ mult) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const float nv = norm[0];
        const float tn = fminf(fmaxf(nv, 0.f), max_norm);
        mult[0] = tn / (eps + nv);
    }
}
extern "C" int asr_norm_clip_mult(const float* norm, float max_norm, float eps, float* mult_out, hipStream_t stream) {
    if (!norm || !mult_out) return ASR_ERR_ARG;
    hipLaunchKernelGGL(norm_mult_kernel, dim3(1), dim3(64), 0, stream, norm, max_norm, eps, mult_out);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}
