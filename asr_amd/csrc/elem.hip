// HBM-bound helper kernels around the scan: layout permutes, row layer-norm (input-side LN hoisted
// out of the recurrence), deterministic column reductions for the vector-parameter gradients,
// bias/activation for the small LHUC / conditioning MLPs, masked mean pooling.
#include "common.h"

// ------------------------------------------------------------------ permute (A,B,D) -> (B,A,D)
__global__ void swap01_kernel(const float* __restrict__ src, float* __restrict__ dst, int A, int B,
                              int D, int ld_src, int ld_dst) {
    size_t n = (size_t)A * B * D;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x) {
        const int d = (int)(i % D);
        const size_t r = i / D;                            // dst row = b * A + a
        const int a = (int)(r % A), b = (int)(r / A);
        dst[r * ld_dst + d] = src[((size_t)a * B + b) * ld_src + d];
    }
}

extern "C" int asr_swap01(const float* src, float* dst, int A, int B, int D, int ld_src, int ld_dst,
                          hipStream_t stream) {
    if (A < 0 || B < 0 || D < 0) return ASR_ERR_ARG;
    size_t n = (size_t)A * B * D;
    if (n == 0) return ASR_OK;
    int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(swap01_kernel, dim3(grid), dim3(256), 0, stream, src, dst, A, B, D, ld_src, ld_dst);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}

// ------------------------------------------------------------------ block reduce helper
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float block_sum(float v, float* sm) {   // 256 threads, sm[4]
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    return sm[0] + sm[1] + sm[2] + sm[3];
}

// sum over the threads of the block whose tid has the same low two bits (= the same gate in the gate-
// interleaved layout, since the column stride of a thread is 256): sm[16]
__device__ __forceinline__ float block_sum_gate(float v, float* sm) {
#pragma unroll
    for (int o = 32; o >= 4; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) < 4) sm[(threadIdx.x >> 6) * 4 + (threadIdx.x & 3)] = v;
    __syncthreads();
    const int g = threadIdx.x & 3;
    return sm[g] + sm[4 + g] + sm[8 + g] + sm[12 + g];
}
__device__ __forceinline__ float block_sum_any(float v, float* sm, int groups) {
    return groups == 4 ? block_sum_gate(v, sm) : block_sum(v, sm);
}

// ------------------------------------------------------------------ row layer norm, forward
// groups = 4: gate_ln / sample_gate_ln (recurrent_layers.py:17-25,34-42): the columns of each gate
// (col & 3 in the interleaved layout) are normalised separately; rstd then holds 4 values per row.
// y[r,:] = alpha[ra,:] * xhat + beta[ra,:] + bias,  xhat = (x+pre[rp,:] - mean)/sqrt(var+eps)
// ra = amod ? r % amod : 0 (per-sample alpha/beta when amod = B), rp = r % pmod.
// x is overwritten in place when y == x.  Saves xhat and rstd for the backward pass.
__global__ __launch_bounds__(256) void ln_rows_fwd_kernel(const float* __restrict__ x, int n,
                                                          const float* __restrict__ pre, int pmod,
                                                          const float* __restrict__ alpha,
                                                          const float* __restrict__ beta, int amod,
                                                          const float* __restrict__ bias,
                                                          float* __restrict__ y, float* __restrict__ xhat,
                                                          float* __restrict__ rstd, int groups) {
    __shared__ float sm[16];
    const size_t r = blockIdx.x;
    const float* xr = x + r * n;
    const float* pr = pre ? pre + (size_t)(r % pmod) * n : nullptr;
    const size_t ra = amod ? r % amod : 0;
    const float ng = (float)(n / groups);
    float s = 0.f;
    for (int c = threadIdx.x; c < n; c += 256) s += xr[c] + (pr ? pr[c] : 0.f);
    const float mean = block_sum_any(s, sm, groups) / ng;
    float q = 0.f;
    for (int c = threadIdx.x; c < n; c += 256) {
        float d = xr[c] + (pr ? pr[c] : 0.f) - mean;
        q += d * d;
    }
    const float var = block_sum_any(q, sm, groups) / ng;
    const float rs = 1.0f / sqrtf(var + ASR_EPS);
    if (rstd && threadIdx.x < groups) rstd[r * groups + threadIdx.x] = rs;
    for (int c = threadIdx.x; c < n; c += 256) {
        float xh = (xr[c] + (pr ? pr[c] : 0.f) - mean) * rs;
        if (xhat) xhat[r * n + c] = xh;
        float v = alpha[ra * n + c] * xh;
        if (beta) v += beta[ra * n + c];
        if (bias) v += bias[c];
        y[r * n + c] = v;
    }
}

extern "C" int asr_ln_rows_fwd(const float* x, int rows, int n, const float* pre, int pmod,
                               const float* alpha, const float* beta, int amod, const float* bias,
                               float* y, float* xhat, float* rstd, int groups, hipStream_t stream) {
    if (rows < 0 || n <= 0 || !x || !y || !alpha || (groups != 1 && groups != 4) || n % groups) return ASR_ERR_ARG;
    if (pre && pmod <= 0) return ASR_ERR_ARG;
    if (rows == 0) return ASR_OK;
    hipLaunchKernelGGL(ln_rows_fwd_kernel, dim3(rows), dim3(256), 0, stream, x, n, pre, pmod, alpha,
                       beta, amod, bias, y, xhat, rstd, groups);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}

// dx = rstd * (dxh - mean(dxh) - xhat * mean(dxh*xhat)), dxh = dy * alpha
__global__ __launch_bounds__(256) void ln_rows_bwd_kernel(const float* __restrict__ dy, int n,
                                                          const float* __restrict__ xhat,
                                                          const float* __restrict__ rstd,
                                                          const float* __restrict__ alpha, int amod,
                                                          float* __restrict__ dx, int groups) {
    __shared__ float sm[16];
    const size_t r = blockIdx.x;
    const size_t ra = amod ? r % amod : 0;
    const float ng = (float)(n / groups);
    float s1 = 0.f, s2 = 0.f;
    for (int c = threadIdx.x; c < n; c += 256) {
        float d = dy[r * n + c] * alpha[ra * n + c];
        s1 += d;
        s2 += d * xhat[r * n + c];
    }
    const float m1 = block_sum_any(s1, sm, groups) / ng;
    const float m2 = block_sum_any(s2, sm, groups) / ng;
    const float rs = rstd[r * groups + (groups == 4 ? (threadIdx.x & 3) : 0)];
    for (int c = threadIdx.x; c < n; c += 256) {
        float d = dy[r * n + c] * alpha[ra * n + c];
        dx[r * n + c] = rs * (d - m1 - xhat[r * n + c] * m2);
    }
}

extern "C" int asr_ln_rows_bwd(const float* dy, int rows, int n, const float* xhat, const float* rstd,
                               const float* alpha, int amod, float* dx, int groups, hipStream_t stream) {
    if (rows < 0 || n <= 0 || !dy || !xhat || !rstd || !alpha || !dx || (groups != 1 && groups != 4) || n % groups)
        return ASR_ERR_ARG;
    if (rows == 0) return ASR_OK;
    hipLaunchKernelGGL(ln_rows_bwd_kernel, dim3(rows), dim3(256), 0, stream, dy, n, xhat, rstd, alpha,
                       amod, dx, groups);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}

// ------------------------------------------------------------------ deterministic column reductions
// out[g, c] (+)= sum over rows r with r % gmod == g of x[r,c] * (y ? y[r,c] : 1)     (gmod >= 1)
// Stage 1: grid (col tiles of 256, S row splits, gmod) -> partial[s][g][c]; stage 2 sums the S splits
// in fixed order.  x / y may have a leading dimension ldx / ldy >= n.
#define CR_SPLIT 32
// One wave per (64 columns, row split, group).  The row loop keeps 8 rows' loads in flight (the one-load-at-a-
// time loop this replaces was latency-bound: 351 us for 19k x 2048); the sums are still taken in row order.
__global__ __launch_bounds__(64) void colreduce1_kernel(const float* __restrict__ x, int ldx,
                                                        const float* __restrict__ y, int ldy, int rows,
                                                        int n, int gmod, float* __restrict__ part) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    const int s = blockIdx.y, g = blockIdx.z;
    if (c >= n) return;
    const int per = (rows / gmod + CR_SPLIT - 1) / CR_SPLIT;     // group-rows per split
    const int q0 = s * per;
    int q1 = q0 + per;
    const int nq = (rows - g + gmod - 1) / gmod;                 // rows in this group
    if (q1 > nq) q1 = nq;
    float acc = 0.f;
    const float* px = x + ((size_t)q0 * gmod + g) * ldx + c;
    const size_t sx = (size_t)gmod * ldx;
    if (y) {
        const float* py = y + ((size_t)q0 * gmod + g) * ldy + c;
        const size_t sy = (size_t)gmod * ldy;
        int q = q0;
        for (; q + 8 <= q1; q += 8) {
            float v[8], w[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { v[i] = px[i * sx]; w[i] = py[i * sy]; }
#pragma unroll
            for (int i = 0; i < 8; ++i) acc += v[i] * w[i];
            px += 8 * sx; py += 8 * sy;
        }
        for (; q < q1; ++q) { acc += (*px) * (*py); px += sx; py += sy; }
    } else {
        int q = q0;
        for (; q + 8 <= q1; q += 8) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = px[i * sx];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc += v[i];
            px += 8 * sx;
        }
        for (; q < q1; ++q) { acc += *px; px += sx; }
    }
    part[((size_t)s * gmod + g) * n + c] = acc;
}
__global__ __launch_bounds__(256) void colreduce2_kernel(const float* __restrict__ part, int n, int gmod,
                                                         float* __restrict__ out, int accumulate) {
    const size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
    if (i >= (size_t)gmod * n) return;
    float acc = 0.f;
    for (int s = 0; s < CR_SPLIT; ++s) acc += part[(size_t)s * gmod * n + i];
    out[i] = accumulate ? out[i] + acc : acc;
}

extern "C" size_t asr_colreduce_ws_bytes(int n, int gmod) {
    return (size_t)CR_SPLIT * (size_t)(gmod > 0 ? gmod : 1) * (size_t)n * sizeof(float);
}

extern "C" int asr_colreduce(const float* x, int ldx, const float* y, int ldy, int rows, int n, int gmod,
                             float* out, int accumulate, float* ws, size_t ws_bytes,
                             hipStream_t stream) {
    if (rows < 0 || n <= 0 || gmod <= 0 || !x || !out || !ws) return ASR_ERR_ARG;
    if (ws_bytes < asr_colreduce_ws_bytes(n, gmod)) return ASR_ERR_WS;
    dim3 g1((n + 63) / 64, CR_SPLIT, gmod);
    hipLaunchKernelGGL(colreduce1_kernel, g1, dim3(64), 0, stream, x, ldx, y, ldy, rows, n, gmod, ws);
    ASR_CHECK_LAUNCH();
    size_t tot = (size_t)gmod * n;
    hipLaunchKernelGGL(colreduce2_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream, ws,
                       n, gmod, out, accumulate);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}

// ------------------------------------------------------------------ small elementwise ops
// act: 0 identity, 1 relu, 2 tanh
__global__ void bias_act_kernel(float* x, const float* bias, size_t rows, int n, int act) {
    size_t tot = rows * n;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < tot;
         i += (size_t)gridDim.x * blockDim.x) {
        float v = x[i] + (bias ? bias[i % n] : 0.f);
        if (act == 1) v = fmaxf(v, 0.f);
        else if (act == 2) v = tanhf(v);
        x[i] = v;
    }
}
extern "C" int asr_bias_act(float* x, const float* bias, int rows, int n, int act, hipStream_t stream) {
    if (rows < 0 || n <= 0 || !x) return ASR_ERR_ARG;
    size_t tot = (size_t)rows * n;
    if (!tot) return ASR_OK;
    int grid = (int)((tot + 255) / 256 < 4096 ? (tot + 255) / 256 : 4096);
    hipLaunchKernelGGL(bias_act_kernel, dim3(grid), dim3(256), 0, stream, x, bias, (size_t)rows, n, act);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}
// dy *= act'(y) with y the activation OUTPUT
__global__ void act_bwd_kernel(float* dy, const float* y, size_t tot, int act) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < tot;
         i += (size_t)gridDim.x * blockDim.x) {
        float v = y[i];
        if (act == 1) dy[i] = v > 0.f ? dy[i] : 0.f;
        else if (act == 2) dy[i] *= (1.f - v * v);
    }
}
extern "C" int asr_act_bwd(float* dy, const float* y, int rows, int n, int act, hipStream_t stream) {
    if (rows < 0 || n <= 0 || !dy || !y) return ASR_ERR_ARG;
    size_t tot = (size_t)rows * n;
    if (!tot) return ASR_OK;
    int grid = (int)((tot + 255) / 256 < 4096 ? (tot + 255) / 256 : 4096);
    hipLaunchKernelGGL(act_bwd_kernel, dim3(grid), dim3(256), 0, stream, dy, y, tot, act);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}

// rows are (t,b) time-major: y[r, :] = a[b, :] * x[r, :] + c0[b, :] + mask[b, t] * c1[b, :]  (b = r % B, t = r / B).
// Gradient of the sequence-variance costs on the inner hidden sequences of scripts/hypernets/exp_prj_hyper.py:108-124
// with respect to those sequences: the three (B, n) coefficient matrices are formed on the host from the per-row
// sums (asr_seq_pool_fwd mode 2, asr_colreduce with gmod = B).
__global__ void affine_rows_kernel(const float* x, int ldx, const float* a, const float* c0, const float* c1,
                                   const float* mask, int msb, int mst, int B, float* y, int ldy, size_t rows, int n) {
    size_t tot = rows * n;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < tot;
         i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / n;
        const int c = (int)(i % n);
        const size_t b = r % B, t = r / B;
        const size_t k = b * n + c;
        y[r * ldy + c] = a[k] * x[r * ldx + c] + c0[k] + mask[b * msb + t * mst] * c1[k];
    }
}
extern "C" int asr_affine_rows(const float* x, int ldx, const float* a, const float* c0, const float* c1,
                               const float* mask, int mask_stride_b, int mask_stride_t, int B, float* y, int ldy,
                               int rows, int n, hipStream_t stream) {
    if (rows < 0 || n <= 0 || B <= 0 || !x || !a || !c0 || !c1 || !mask || !y || rows % B) return ASR_ERR_ARG;
    size_t tot = (size_t)rows * n;
    if (!tot) return ASR_OK;
    int grid = (int)((tot + 255) / 256 < 4096 ? (tot + 255) / 256 : 4096);
    hipLaunchKernelGGL(affine_rows_kernel, dim3(grid), dim3(256), 0, stream, x, ldx, a, c0, c1, mask, mask_stride_b,
                       mask_stride_t, B, y, ldy, (size_t)rows, n);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}

// rows are (t,b) time-major: y[r, :] = x[r, :] * s[r % B, :] (s optionally passed through relu)
// LHUC epilogue (libs/lhuclstm_layers.py:461-462) and its gradient wrt the hidden sequence.
__global__ void rowscale_kernel(const float* x, int ldx, const float* s, int B, int relu, float* y,
                                int ldy, size_t rows, int n) {
    size_t tot = rows * n;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < tot;
         i += (size_t)gridDim.x * blockDim.x) {
        size_t r = i / n;
        int c = (int)(i % n);
        float sv = s[(r % B) * n + c];
        if (relu) sv = fmaxf(sv, 0.f);
        y[r * ldy + c] = x[r * ldx + c] * sv;
    }
}
extern "C" int asr_rowscale(const float* x, int ldx, const float* s, int B, int relu, float* y, int ldy,
                            int rows, int n, hipStream_t stream) {
    if (rows < 0 || n <= 0 || B <= 0 || !x || !s || !y) return ASR_ERR_ARG;
    size_t tot = (size_t)rows * n;
    if (!tot) return ASR_OK;
    int grid = (int)((tot + 255) / 256 < 4096 ? (tot + 255) / 256 : 4096);
    hipLaunchKernelGGL(rowscale_kernel, dim3(grid), dim3(256), 0, stream, x, ldx, s, B, relu, y, ldy,
                       (size_t)rows, n);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}

// masked mean over time of a batch-major tensor: out[b,f] = sum_t x[b,t,f]*mask[b,t] / sum_t mask[b,t]
// (libs/lasagne_libs/recurrent_layers.py:1211); one workgroup per (b), threads over f.
__global__ void masked_mean_kernel(const float* x, const float* mask, int T, int F, float* out) {
    const int b = blockIdx.x;
    float msum = 0.f;
    for (int t = 0; t < T; ++t) msum += mask[(size_t)b * T + t];
    for (int f = threadIdx.x; f < F; f += blockDim.x) {
        float acc = 0.f;
        for (int t = 0; t < T; ++t) acc += x[((size_t)b * T + t) * F + f] * mask[(size_t)b * T + t];
        out[(size_t)b * F + f] = acc / msum;
    }
}
extern "C" int asr_masked_mean(const float* x, const float* mask, int B, int T, int F, float* out,
                               hipStream_t stream) {
    if (B < 0 || T <= 0 || F <= 0 || !x || !mask || !out) return ASR_ERR_ARG;
    if (!B) return ASR_OK;
    hipLaunchKernelGGL(masked_mean_kernel, dim3(B), dim3(64), 0, stream, x, mask, T, F, out);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}
// backward of the masked mean fused with tanh': dpre[b,t,f] = ds[b,f]/msum[b]*mask[b,t]*(1-pre^2)
__global__ void masked_mean_bwd_kernel(const float* ds, const float* mask, const float* pre, int T,
                                       int F, float* dpre) {
    const int b = blockIdx.x;
    float msum = 0.f;
    for (int t = 0; t < T; ++t) msum += mask[(size_t)b * T + t];
    for (size_t i = threadIdx.x; i < (size_t)T * F; i += blockDim.x) {
        int t = (int)(i / F), f = (int)(i % F);
        float p = pre[(size_t)b * T * F + i];
        dpre[(size_t)b * T * F + i] = ds[(size_t)b * F + f] / msum * mask[(size_t)b * T + t] * (1.f - p * p);
    }
}
extern "C" int asr_masked_mean_tanh_bwd(const float* ds, const float* mask, const float* pre, int B,
                                        int T, int F, float* dpre, hipStream_t stream) {
    if (B < 0 || T <= 0 || F <= 0 || !ds || !mask || !pre || !dpre) return ASR_ERR_ARG;
    if (!B) return ASR_OK;
    hipLaunchKernelGGL(masked_mean_bwd_kernel, dim3(B), dim3(256), 0, stream, ds, mask, pre, T, F, dpre);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}

// ------------------------------------------------------------------ strided sequence pooling
// The conditioning prologue of the cond-LN layers pools an activation over time per utterance:
//   mode 0: masked mean of tanh(.)  (recurrent_layers.py:945-946, :1210-1211)
//   mode 1: unmasked max of relu(.) (:1469-1470, :1734-1735)
//   mode 2: masked mean of the values as they are (no activation derivative in the backward): the
//           SummarizingLayer T.mean(axis=1) of libs/builder_utils.py:5-22 with an all-ones mask
// x holds the ACTIVATED values, element (b,t,f) at x[b*sb + t*st + f] (time-major own-input pooling:
// sb = F, st = B*F; batch-major i-vector pooling: sb = T*F, st = F); mask is (B,T) contiguous.
__global__ void seq_pool_fwd_kernel(const float* __restrict__ x, size_t sb, size_t st,
                                    const float* __restrict__ mask, int T, int F, int mode,
                                    float* __restrict__ out) {
    const int b = blockIdx.x;
    const int f = blockIdx.y * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const float* xb = x + (size_t)b * sb + f;
    if (mode != 1) {
        float acc = 0.f, msum = 0.f;
        for (int t = 0; t < T; ++t) {
            const float m = mask[(size_t)b * T + t];
            acc += xb[(size_t)t * st] * m;
            msum += m;
        }
        out[(size_t)b * F + f] = acc / msum;
    } else {
        float mx = xb[0];
        for (int t = 1; t < T; ++t) mx = fmaxf(mx, xb[(size_t)t * st]);
        out[(size_t)b * F + f] = mx;
    }
}
extern "C" int asr_seq_pool_fwd(const float* x, size_t sb, size_t st, const float* mask, int B, int T, int F,
                                int mode, float* out, hipStream_t stream) {
    if (B < 0 || T <= 0 || F <= 0 || !x || !out || (mode != 1 && !mask) || mode < 0 || mode > 2) return ASR_ERR_ARG;
    if (!B) return ASR_OK;
    hipLaunchKernelGGL(seq_pool_fwd_kernel, dim3(B, (F + 63) / 64), dim3(64), 0, stream, x, sb, st, mask, T, F,
                       mode, out);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}
// backward incl. the activation derivative; dx has the layout of x.
//   mode 0: dx = ds[b,f] / sum_t mask * mask[b,t] * (1 - x^2)
//   mode 1: dx = ds[b,f] * (x == s[b,f]) * (x > 0)   (Theano's max gradient reaches every tied frame)
__global__ void seq_pool_bwd_kernel(const float* __restrict__ ds, const float* __restrict__ s,
                                    const float* __restrict__ x, size_t sb, size_t st,
                                    const float* __restrict__ mask, int T, int F, int mode,
                                    float* __restrict__ dx) {
    const int b = blockIdx.x;
    const int f = blockIdx.y * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const size_t base = (size_t)b * sb + f;
    const float d = ds[(size_t)b * F + f];
    if (mode != 1) {
        float msum = 0.f;
        for (int t = 0; t < T; ++t) msum += mask[(size_t)b * T + t];
        const float dn = d / msum;
        for (int t = 0; t < T; ++t) {
            const float p = mode == 0 ? x[base + (size_t)t * st] : 0.f;
            dx[base + (size_t)t * st] = dn * mask[(size_t)b * T + t] * (1.f - p * p);
        }
    } else {
        const float mx = s[(size_t)b * F + f];
        for (int t = 0; t < T; ++t) {
            const float p = x[base + (size_t)t * st];
            dx[base + (size_t)t * st] = (p == mx && p > 0.f) ? d : 0.f;
        }
    }
}
extern "C" int asr_seq_pool_bwd(const float* ds, const float* s, const float* x, size_t sb, size_t st,
                                const float* mask, int B, int T, int F, int mode, float* dx,
                                hipStream_t stream) {
    if (B < 0 || T <= 0 || F <= 0 || !ds || !dx || (mode != 2 && !x) || (mode != 1 && !mask) || (mode == 1 && !s) ||
        mode < 0 || mode > 2)
        return ASR_ERR_ARG;
    if (!B) return ASR_OK;
    hipLaunchKernelGGL(seq_pool_bwd_kernel, dim3(B, (F + 63) / 64), dim3(64), 0, stream, ds, s, x, sb, st, mask,
                       T, F, mode, dx);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}

// ------------------------------------------------------------------ gate-major <-> gate-interleaved
// dir 0: dst[r][u*4+g] = src[r][g*N+u]   (reference stacking -> engine layout);  dir 1: inverse
__global__ void perm4_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t rows, int N,
                             int dir) {
    const size_t tot = rows * 4 * (size_t)N;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < tot;
         i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / (4 * (size_t)N);
        const int c = (int)(i % (4 * (size_t)N));
        if (dir == 0) { const int u = c >> 2, g = c & 3; dst[i] = src[r * 4 * N + (size_t)g * N + u]; }
        else          { const int g = c / N, u = c % N;  dst[i] = src[r * 4 * N + (size_t)u * 4 + g]; }
    }
}
extern "C" int asr_perm4(const float* src, float* dst, int rows, int N, int dir, hipStream_t stream) {
    if (rows < 0 || N <= 0 || !src || !dst) return ASR_ERR_ARG;
    size_t tot = (size_t)rows * 4 * N;
    if (!tot) return ASR_OK;
    int grid = (int)((tot + 255) / 256 < 4096 ? (tot + 255) / 256 : 4096);
    hipLaunchKernelGGL(perm4_kernel, dim3(grid), dim3(256), 0, stream, src, dst, (size_t)rows, N, dir);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}
