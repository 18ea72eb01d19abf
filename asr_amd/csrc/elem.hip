// HBM-bound helper kernels around the scan: layout permutes, row layer-norm (input-side LN hoisted
// out of the recurrence), deterministic column reductions for the vector-parameter gradients,
// bias/activation for the small LHUC / conditioning MLPs, masked mean pooling.
#include "common.h"
#include <mutex>

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

// the same with the row held in registers (n <= 256 NV): one read of x, every load issued before the first use
template <int NV>
__global__ __launch_bounds__(256) void ln_rows_fwd_reg_kernel(const float* __restrict__ x, int n,
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
    float v[NV], al[NV], be[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = threadIdx.x + 256 * k;
        const int cc = c < n ? c : n - 1;
        v[k] = xr[cc] + (pr ? pr[cc] : 0.f);
        al[k] = alpha[ra * n + cc];
        be[k] = (beta ? beta[ra * n + cc] : 0.f) + (bias ? bias[cc] : 0.f);
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) s += (threadIdx.x + 256 * k < n) ? v[k] : 0.f;
    const float mean = block_sum_any(s, sm, groups) / ng;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) { const float d = v[k] - mean; q += (threadIdx.x + 256 * k < n) ? d * d : 0.f; }
    const float var = block_sum_any(q, sm, groups) / ng;
    const float rs = 1.0f / sqrtf(var + ASR_EPS);
    if (rstd && threadIdx.x < groups) rstd[r * groups + threadIdx.x] = rs;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = threadIdx.x + 256 * k;
        if (c < n) {
            const float xh = (v[k] - mean) * rs;
            if (xhat) xhat[r * n + c] = xh;
            y[r * n + c] = al[k] * xh + be[k];
        }
    }
}

extern "C" int asr_ln_rows_fwd(const float* x, int rows, int n, const float* pre, int pmod,
                               const float* alpha, const float* beta, int amod, const float* bias,
                               float* y, float* xhat, float* rstd, int groups, hipStream_t stream) {
    if (rows < 0 || n <= 0 || !x || !y || !alpha || (groups != 1 && groups != 4) || n % groups) return ASR_ERR_ARG;
    if (pre && pmod <= 0) return ASR_ERR_ARG;
    if (rows == 0) return ASR_OK;
#define LN_FWD_(NV_) hipLaunchKernelGGL((ln_rows_fwd_reg_kernel<NV_>), dim3(rows), dim3(256), 0, stream, x, n, pre, \
                                        pmod, alpha, beta, amod, bias, y, xhat, rstd, groups)
    if (n <= 1024) LN_FWD_(4); else if (n <= 2048) LN_FWD_(8); else if (n <= 4096) LN_FWD_(16);
    else hipLaunchKernelGGL(ln_rows_fwd_kernel, dim3(rows), dim3(256), 0, stream, x, n, pre, pmod, alpha,
                            beta, amod, bias, y, xhat, rstd, groups);
#undef LN_FWD_
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

template <int NV>
__global__ __launch_bounds__(256) void ln_rows_bwd_reg_kernel(const float* __restrict__ dy, int n,
                                                              const float* __restrict__ xhat,
                                                              const float* __restrict__ rstd,
                                                              const float* __restrict__ alpha, int amod,
                                                              float* __restrict__ dx, int groups) {
    __shared__ float sm[16];
    const size_t r = blockIdx.x;
    const size_t ra = amod ? r % amod : 0;
    const float ng = (float)(n / groups);
    float d[NV], xh[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = threadIdx.x + 256 * k;
        const int cc = c < n ? c : n - 1;
        d[k] = dy[r * n + cc] * alpha[ra * n + cc];
        xh[k] = xhat[r * n + cc];
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const bool ok = threadIdx.x + 256 * k < n;
        s1 += ok ? d[k] : 0.f;
        s2 += ok ? d[k] * xh[k] : 0.f;
    }
    const float m1 = block_sum_any(s1, sm, groups) / ng;
    const float m2 = block_sum_any(s2, sm, groups) / ng;
    const float rs = rstd[r * groups + (groups == 4 ? (threadIdx.x & 3) : 0)];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = threadIdx.x + 256 * k;
        if (c < n) dx[r * n + c] = rs * (d[k] - m1 - xh[k] * m2);
    }
}

extern "C" int asr_ln_rows_bwd(const float* dy, int rows, int n, const float* xhat, const float* rstd,
                               const float* alpha, int amod, float* dx, int groups, hipStream_t stream) {
    if (rows < 0 || n <= 0 || !dy || !xhat || !rstd || !alpha || !dx || (groups != 1 && groups != 4) || n % groups)
        return ASR_ERR_ARG;
    if (rows == 0) return ASR_OK;
#define LN_BWD_(NV_) hipLaunchKernelGGL((ln_rows_bwd_reg_kernel<NV_>), dim3(rows), dim3(256), 0, stream, dy, n, xhat, \
                                        rstd, alpha, amod, dx, groups)
    if (n <= 1024) LN_BWD_(4); else if (n <= 2048) LN_BWD_(8); else if (n <= 4096) LN_BWD_(16);
    else hipLaunchKernelGGL(ln_rows_bwd_kernel, dim3(rows), dim3(256), 0, stream, dy, n, xhat, rstd, alpha,
                            amod, dx, groups);
#undef LN_BWD_
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
    const int per = ((rows + gmod - 1) / gmod + CR_SPLIT - 1) / CR_SPLIT;   // group-rows per split, sized for the
                                                                            // fullest group (rows % gmod != 0)
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

// The same reduction for callers that cannot allocate or hand over a workspace (code under HIP graph
// capture, tight serving loops): the stage-1 partials live in a file-scope device array.  Its name is
// only a host-side handle; the device address is looked up once per device (hipGetSymbolAddress) and
// that pointer is what the kernels get.  Every call on a GPU shares that GPU's array: calls that
// overlap on one GPU (two streams) race.  Nothing in it has to start at zero (stage 1 writes every
// partial stage 2 reads).
#define CR_STATIC_MAX 32768                       // gmod * n of one call
static __device__ __align__(256) float cr_static_scratch[(size_t)CR_SPLIT * CR_STATIC_MAX];   // 4 MiB per GPU
#define CR_MAX_DEVICES 64
static float* cr_static_ptr[CR_MAX_DEVICES];
static std::mutex cr_static_mu;

static int cr_static_lookup(float** out, hipStream_t stream) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= CR_MAX_DEVICES) return ASR_ERR_LAUNCH;
    // every GPU holds its own copy of the array: a stream of another device must not get this device's address
    if (stream) {
        int sdev = -1;
        if (hipStreamGetDevice(stream, &sdev) != hipSuccess) return ASR_ERR_LAUNCH;
        if (sdev != dev) return ASR_ERR_ARG;
    }
    if (cr_static_ptr[dev]) { *out = cr_static_ptr[dev]; return ASR_OK; }    // resolved before: no lock, no HIP call
    std::lock_guard<std::mutex> lk(cr_static_mu);
    if (!cr_static_ptr[dev]) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(cr_static_scratch)) != hipSuccess || !p) return ASR_ERR_LAUNCH;
        cr_static_ptr[dev] = (float*)p;
    }
    *out = cr_static_ptr[dev];
    return ASR_OK;
}

extern "C" int asr_colreduce_static(const float* x, int ldx, const float* y, int ldy, int rows, int n, int gmod,
                                    float* out, int accumulate, hipStream_t stream) {
    if (rows < 0 || n <= 0 || gmod <= 0 || (!x && rows != 0) || (!out && x)) return ASR_ERR_ARG;
    if ((size_t)gmod * (size_t)n > CR_STATIC_MAX) return ASR_ERR_LIMIT;
    float* ws = nullptr;
    const int rc = cr_static_lookup(&ws, stream);
    if (rc != ASR_OK) return rc;
    if (rows == 0 && !x) return ASR_OK;              // the warm-up form (see the header): address resolved, nothing launched
    dim3 g1((n + 63) / 64, CR_SPLIT, gmod);
    hipLaunchKernelGGL(colreduce1_kernel, g1, dim3(64), 0, stream, x, ldx, y, ldy, rows, n, gmod, ws);
    ASR_CHECK_LAUNCH();
    const size_t tot = (size_t)gmod * n;
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

// The (B,F) algebra of the inter / intra variance costs of scripts/hypernets/exp_prj_hyper.py:108-124 on per-sequence
// statistics st = [mu ; S0 ; S2] (masked mean, plain sum and sum of squares over time of one direction's inner
// hidden sequence, each (Bg, F)):  inter = -0.5 mean_f var_b(mu),  intra = 0.5 mean_{b,f} (S2 - 2 mu S0 + T mu^2) / len_b,
// and the coefficients (a, c0, c1) with which asr_affine_rows turns the sequence into the gradient of
// w_inter * inter + w_intra * intra for the rows [b0, b0 + Bloc) of this process (tpad: per-row count of frames the
// sums ran over when the rows come from shards with different padding, else T).  One workgroup; sums in double, in a
// fixed order.
__global__ __launch_bounds__(256) void var_costs_kernel(const float* __restrict__ st, const float* __restrict__ len,
                                                         const float* __restrict__ tpad, int Bg, int F, int T, float w_inter, float w_intra, int b0,
                                                         int Bloc, float* __restrict__ coef, float* __restrict__ costs,
                                                         int accumulate) {
    __shared__ double red[2][256];
    const float* mu = st; const float* S0 = st + (size_t)Bg * F; const float* S2 = st + (size_t)2 * Bg * F;
    double acc_inter = 0.0, acc_intra = 0.0;
    for (int f = threadIdx.x; f < F; f += 256) {
        double sm = 0.0;
        for (int b = 0; b < Bg; ++b) sm += (double)mu[(size_t)b * F + f];
        const double mbar = sm / Bg;
        double var = 0.0;
        for (int b = 0; b < Bg; ++b) {
            const double m = (double)mu[(size_t)b * F + f], L = (double)len[b];
            const double Tb = tpad ? (double)tpad[b] : (double)T;          // frames the sums of row b ran over
            const double s0 = (double)S0[(size_t)b * F + f], s2 = (double)S2[(size_t)b * F + f];
            var += (m - mbar) * (m - mbar);
            acc_intra += (s2 - 2.0 * m * s0 + Tb * m * m) / L;
            const int bl = b - b0;
            if (coef && bl >= 0 && bl < Bloc) {
                const double a = ((double)w_intra / ((double)Bg * F)) / L;
                coef[((size_t)0 * Bloc + bl) * F + f] = (float)a;
                coef[((size_t)1 * Bloc + bl) * F + f] = (float)(-a * m);
                coef[((size_t)2 * Bloc + bl) * F + f] = (float)(-(double)w_intra * (s0 - Tb * m) / ((double)Bg * F * L * L)
                                                                - (double)w_inter * (m - mbar) / ((double)F * Bg * L));
            }
        }
        acc_inter += var / Bg;
    }
    red[0][threadIdx.x] = acc_inter; red[1][threadIdx.x] = acc_intra;
    __syncthreads();
    if (threadIdx.x < 2) {
        double s_ = 0.0;
        for (int i = 0; i < 256; ++i) s_ += red[threadIdx.x][i];
        const float v = threadIdx.x == 0 ? (float)(-0.5 * s_ / F) : (float)(0.5 * s_ / ((double)Bg * F));
        costs[threadIdx.x] = accumulate ? costs[threadIdx.x] + v : v;
    }
}
extern "C" int asr_var_costs(const float* st, const float* len, const float* tpad, int Bg, int F, int T, float w_inter, float w_intra,
                             int b0, int Bloc, float* coef, float* costs, int accumulate, hipStream_t stream) {
    if (!st || !len || !costs || Bg <= 0 || F <= 0 || T <= 0 || b0 < 0 || Bloc < 0 || b0 + Bloc > Bg) return ASR_ERR_ARG;
    hipLaunchKernelGGL(var_costs_kernel, dim3(1), dim3(256), 0, stream, st, len, tpad, Bg, F, T, w_inter, w_intra, b0, Bloc,
                       coef, costs, accumulate);
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

// ------------------------------------------------------------------ inverted dropout
// lasagne DropoutLayer(p) in front of every layer and of the output layer of models/gating_hyper_nets.py
// (:276-307): y = x * keep / (1 - p), keep ~ Bernoulli(1 - p); identity when deterministic.  Theano's MRG
// stream cannot be reproduced, so the mask comes from a counter-based generator and is never stored:
// Philox4x32-10 keyed by `seed`, counter = (row * ceil(n/4) + column group, `stream`), one call per 4
// consecutive columns; the backward pass calls the same entry point on the gradient with the same
// (seed, stream) and regenerates the mask.  keep <=> u >= p with u = (word >> 8) * 2^-24.
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0,
                                              unsigned k1, unsigned out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y,
                                                      int ldy, size_t rows, int n, float p, float scale,
                                                      unsigned long long seed, unsigned long long stream) {
    const size_t ng = (size_t)(n + 3) / 4, tot = rows * ng;
    const bool vec = !(n & 3) && !(ldx & 3) && !(ldy & 3) && !(((uintptr_t)x | (uintptr_t)y) & 15);
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < tot; i += (size_t)gridDim.x * 256) {
        const size_t r = i / ng;
        const int g = (int)(i % ng);
        unsigned w[4];
        philox4x32_10((unsigned)i, (unsigned)(i >> 32), (unsigned)stream, (unsigned)(stream >> 32), (unsigned)seed,
                      (unsigned)(seed >> 32), w);
        if (vec) {
            f32x4 v = *(const f32x4*)(x + r * ldx + g * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ((float)(w[e] >> 8) * 5.9604644775390625e-08f >= p) ? v[e] * scale : 0.f;
            *(f32x4*)(y + r * ldy + g * 4) = v;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = g * 4 + e;
                if (c < n) y[r * ldy + c] = ((float)(w[e] >> 8) * 5.9604644775390625e-08f >= p) ? x[r * ldx + c] * scale : 0.f;
            }
        }
    }
}
extern "C" int asr_dropout(const float* x, int ldx, float* y, int ldy, int rows, int n, float p,
                           unsigned long long seed, unsigned long long stream_id, hipStream_t stream) {
    if (rows < 0 || n <= 0 || !x || !y || !(p >= 0.f) || !(p < 1.f) || ldx < n || ldy < n) return ASR_ERR_ARG;
    const size_t tot = (size_t)rows * ((n + 3) / 4);
    if (!tot) return ASR_OK;
    const int grid = (int)((tot + 255) / 256 < 8192 ? (tot + 255) / 256 : 8192);
    hipLaunchKernelGGL(dropout_kernel, dim3(grid), dim3(256), 0, stream, x, ldx, y, ldy, (size_t)rows, n, p,
                       1.0f / (1.0f - p), seed, stream_id);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}

// masked mean over time of a batch-major tensor: out[b,f] = sum_t x[b,t,f]*mask[b,t] / sum_t mask[b,t]
// (libs/lasagne_libs/recurrent_layers.py:1211); one workgroup per (b), threads over f.
// (masked_mean / masked_mean_tanh_bwd are the batch-major (B,T,F) cases of the strided pooling kernels below)

// ------------------------------------------------------------------ strided sequence pooling
// The conditioning prologue of the cond-LN layers pools an activation over time per utterance:
//   mode 0: masked mean of tanh(.)  (recurrent_layers.py:945-946, :1210-1211)
//   mode 1: unmasked max of relu(.) (:1469-1470, :1734-1735)
//   mode 2: masked mean of the values as they are (no activation derivative in the backward): the
//           SummarizingLayer T.mean(axis=1) of libs/builder_utils.py:5-22 with an all-ones mask
// x holds the ACTIVATED values, element (b,t,f) at x[b*sb + t*st + f] (time-major own-input pooling:
// sb = F, st = B*F; batch-major i-vector pooling: sb = T*F, st = F); mask is (B,T) contiguous.
// Forward: one 1024-thread workgroup per (utterance, 64 features): 16 time groups x 64 feature lanes, 8 loads
// in flight per lane, the 16 partial results combined in group order (deterministic).  The one-thread-per-
// feature loop over T this replaces took 0.5 ms per call at T = 2000 (latency-bound, 16 workgroups).
#define SP_TG 16
__global__ __launch_bounds__(1024) void seq_pool_fwd_kernel(const float* __restrict__ x, size_t sb, size_t st,
                                                            const float* __restrict__ mask, int T, int F, int mode,
                                                            float* __restrict__ out) {
    __shared__ float racc[SP_TG][64];
    __shared__ float rms[SP_TG];
    const int b = blockIdx.x, fl = threadIdx.x & 63, tg = threadIdx.x >> 6;
    const int f = blockIdx.y * 64 + fl;
    const bool fv = f < F;
    const float* xb = x + (size_t)b * sb + (fv ? f : 0);
    const float* mb = mode != 1 ? mask + (size_t)b * T : nullptr;
    const float NEG = -3.402823466e38f;
    float acc = mode == 1 ? NEG : 0.f, ms = 0.f;
    for (int t0 = tg; t0 < T; t0 += SP_TG * 8) {
        float v[8], m[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int t = t0 + i * SP_TG;
            const int tt = t < T ? t : T - 1;
            v[i] = xb[(size_t)tt * st];
            m[i] = mode != 1 ? mb[tt] : 1.f;
            if (t >= T) { m[i] = 0.f; if (mode == 1) v[i] = NEG; }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (mode != 1) { acc += v[i] * m[i]; ms += m[i]; }
            else acc = fmaxf(acc, v[i]);
        }
    }
    racc[tg][fl] = acc;
    if (fl == 0) rms[tg] = ms;
    __syncthreads();
    if (tg == 0 && fv) {
        float a = racc[0][fl], msum = rms[0];
#pragma unroll
        for (int g = 1; g < SP_TG; ++g) {
            if (mode != 1) { a += racc[g][fl]; msum += rms[g]; }
            else a = fmaxf(a, racc[g][fl]);
        }
        out[(size_t)b * F + f] = mode != 1 ? a / msum : a;
    }
}
static int seq_pool_fwd_launch(const float* x, size_t sb, size_t st, const float* mask, int B, int T, int F, int mode,
                               float* out, hipStream_t stream) {
    hipLaunchKernelGGL(seq_pool_fwd_kernel, dim3(B, (F + 63) / 64), dim3(1024), 0, stream, x, sb, st, mask, T, F,
                       mode, out);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}
extern "C" int asr_seq_pool_fwd(const float* x, size_t sb, size_t st, const float* mask, int B, int T, int F,
                                int mode, float* out, hipStream_t stream) {
    if (B < 0 || T <= 0 || F <= 0 || !x || !out || (mode != 1 && !mask) || mode < 0 || mode > 2) return ASR_ERR_ARG;
    if (!B) return ASR_OK;
    return seq_pool_fwd_launch(x, sb, st, mask, B, T, F, mode, out, stream);
}
// backward incl. the activation derivative; dx has the layout of x.
//   mode 0: dx = ds[b,f] / sum_t mask * mask[b,t] * (1 - x^2)
//   mode 1: dx = ds[b,f] * (x == s[b,f]) * (x > 0)   (Theano's max gradient reaches every tied frame)
// One workgroup per (utterance, 64 time steps): 4 time groups x 64 feature lanes.
#define SP_TC 64
__global__ __launch_bounds__(256) void seq_pool_bwd_kernel(const float* __restrict__ ds, const float* __restrict__ s,
                                                           const float* __restrict__ x, size_t sb, size_t st,
                                                           const float* __restrict__ mask, int T, int F, int mode,
                                                           float* __restrict__ dx) {
    __shared__ float red[256];
    const int b = blockIdx.x, tid = threadIdx.x, fl = tid & 63, tg = tid >> 6;
    const int t0 = blockIdx.y * SP_TC, t1 = min(T, t0 + SP_TC);
    float msum = 1.f;
    if (mode != 1) {                                     // sum of the utterance's mask, fixed reduction tree
        float a = 0.f;
        for (int k = tid; k < T; k += 256) a += mask[(size_t)b * T + k];
        red[tid] = a;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) red[tid] += red[tid + o];
            __syncthreads();
        }
        msum = red[0];
    }
    for (int f0 = 0; f0 < F; f0 += 64) {
        const int f = f0 + fl;
        if (f >= F) continue;
        const size_t base = (size_t)b * sb + f;
        const float d = ds[(size_t)b * F + f];
        if (mode != 1) {
            const float dn = d / msum;
            for (int t = t0 + tg; t < t1; t += 4) {
                const float p = mode == 0 ? x[base + (size_t)t * st] : 0.f;
                dx[base + (size_t)t * st] = dn * mask[(size_t)b * T + t] * (1.f - p * p);
            }
        } else {
            const float mx = s[(size_t)b * F + f];
            for (int t = t0 + tg; t < t1; t += 4) {
                const float p = x[base + (size_t)t * st];
                dx[base + (size_t)t * st] = (p == mx && p > 0.f) ? d : 0.f;
            }
        }
    }
}
static int seq_pool_bwd_launch(const float* ds, const float* s, const float* x, size_t sb, size_t st,
                               const float* mask, int B, int T, int F, int mode, float* dx, hipStream_t stream) {
    hipLaunchKernelGGL(seq_pool_bwd_kernel, dim3(B, (T + SP_TC - 1) / SP_TC), dim3(256), 0, stream, ds, s, x, sb, st,
                       mask, T, F, mode, dx);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}
extern "C" int asr_seq_pool_bwd(const float* ds, const float* s, const float* x, size_t sb, size_t st,
                                const float* mask, int B, int T, int F, int mode, float* dx,
                                hipStream_t stream) {
    if (B < 0 || T <= 0 || F <= 0 || !ds || !dx || (mode != 2 && !x) || (mode != 1 && !mask) || (mode == 1 && !s) ||
        mode < 0 || mode > 2)
        return ASR_ERR_ARG;
    if (!B) return ASR_OK;
    return seq_pool_bwd_launch(ds, s, x, sb, st, mask, B, T, F, mode, dx, stream);
}
// batch-major (B,T,F) masked mean and the backward of masked-mean-of-tanh: dpre = ds/msum * mask * (1 - pre^2)
extern "C" int asr_masked_mean(const float* x, const float* mask, int B, int T, int F, float* out,
                               hipStream_t stream) {
    if (B < 0 || T <= 0 || F <= 0 || !x || !mask || !out) return ASR_ERR_ARG;
    if (!B) return ASR_OK;
    return seq_pool_fwd_launch(x, (size_t)T * F, (size_t)F, mask, B, T, F, 2, out, stream);
}
extern "C" int asr_masked_mean_tanh_bwd(const float* ds, const float* mask, const float* pre, int B,
                                        int T, int F, float* dpre, hipStream_t stream) {
    if (B < 0 || T <= 0 || F <= 0 || !ds || !mask || !pre || !dpre) return ASR_ERR_ARG;
    if (!B) return ASR_OK;
    return seq_pool_bwd_launch(ds, nullptr, pre, (size_t)T * F, (size_t)F, mask, B, T, F, 0, dpre, stream);
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

// ------------------------------------------------------------------ activation probe (tests only)
// y[i] = the library's sigmoid (kind 0) / tanh (kind 1) / exp (kind 2) of x[i]: the very inline functions the
// scans and the softmax use (common.h), so their accuracy can be measured against float64 from the host.
__global__ void act_eval_kernel(int kind, const float* __restrict__ x, float* __restrict__ y, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        y[i] = kind == 0 ? sigmoidf_(v) : kind == 1 ? tanhf_(v) : expf_(v);
    }
}
extern "C" int asr_act_eval(int kind, const float* x, float* y, size_t n, hipStream_t stream) {
    if (kind < 0 || kind > 2 || !x || !y) return ASR_ERR_ARG;
    if (!n) return ASR_OK;
    const int grid = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL(act_eval_kernel, dim3(grid), dim3(256), 0, stream, kind, x, y, n);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}
