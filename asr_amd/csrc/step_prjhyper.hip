// ProjectionHyperLSTMLayer (libs/lasagne_libs/hyper_layers.py:14-523; step :319-417, step_masked
// :419-456, scan :492-519), first HIP path: one time step = a short chain of small kernels, the loop
// over T runs on the host inside this library (no per-step Python, no allocation, one stream).
//
// Why not a mode of the persistent scans: both cells are N units wide with cell layer norm and a
// projection to F, coupled inside the step through three scale maps (F x 12N); folded weights are
// ~190 KB per workgroup at N = 512 and the step has 4-6 all-to-all exchanges.  This version keeps the
// weights in L2 and is launch-bound (~12 launches per step, forward); a persistent kernel is the next
// step.
//
// Layouts here are the reference's: gate vectors are gate-major [i | f | c | o] blocks of N, sequences
// time-major (T,B,.).  The recurrent state that is carried is r = o*tanh(outcell) (N wide, masked like
// the reference's hid): hid_t = r_t . W_prj is linear, so switching r is switching hid (:441-444).
#include "common.h"

// ------------------------------------------------------------------------------------------------
// C[Bm x Nc] = A[Bm x K] . W[K x Nc] (+ bias) (+ C if beta), Bm <= 16: one 16x16 MFMA tile per wave,
// W streamed from L2, A re-read per k-step (a few KB, L1 resident).
__global__ __launch_bounds__(256) void skinny_nn_kernel(const float* __restrict__ A, int lda,
                                                        const float* __restrict__ W, int ldw,
                                                        const float* __restrict__ bias, float* __restrict__ Cm,
                                                        int ldc, int Bm, int K, int Nc, int beta) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, bl = lane & 15;
    const int n0 = (blockIdx.x * 4 + wave) * 16;
    if (n0 >= Nc) return;
    const int col = n0 + bl;
    const bool cv = col < Nc, rv = bl < Bm;
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    for (int k0 = 0; k0 < K; k0 += 32) {
        float a[8], b[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int k = k0 + 4 * s + q;
            a[s] = (rv && k < K) ? A[(size_t)bl * lda + k] : 0.f;
            b[s] = (cv && k < K) ? W[(size_t)k * ldw + col] : 0.f;
        }
#pragma unroll
        for (int s = 0; s < 8; s += 2) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s + 1], b[s + 1], acc1, 0, 0, 0);
        }
    }
    const f32x4 acc = acc0 + acc1;                      // rows q*4..q*4+3 (batch), column bl
    if (cv) {
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = q * 4 + r;
            if (row < Bm) {
                float* p = Cm + (size_t)row * ldc + col;
                *p = acc[r] + bv + (beta ? *p : 0.f);
            }
        }
    }
}

// C[Bm x Kc] = A[Bm x Nn] . W[Kc x Nn]^T (+ C if beta); Nn % 4 == 0, 16-byte aligned rows.
__global__ __launch_bounds__(256) void skinny_nt_kernel(const float* __restrict__ A, int lda,
                                                        const float* __restrict__ W, int ldw,
                                                        float* __restrict__ Cm, int ldc, int Bm, int Nn, int Kc,
                                                        int beta) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, bl = lane & 15;
    const int k0 = (blockIdx.x * 4 + wave) * 16;
    if (k0 >= Kc) return;
    const int kout = k0 + bl;
    const bool kv = kout < Kc, rv = bl < Bm;
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    for (int g = 0; g < Nn; g += 16) {                  // MFMA k index q <-> columns g+4q .. g+4q+3
        const int c = g + 4 * q;
        f32x4 av = {0, 0, 0, 0}, wv = {0, 0, 0, 0};
        if (c < Nn) {
            if (rv) av = *(const f32x4*)(A + (size_t)bl * lda + c);
            if (kv) wv = *(const f32x4*)(W + (size_t)kout * ldw + c);
        }
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], wv[0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], wv[1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], wv[2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], wv[3], acc1, 0, 0, 0);
    }
    const f32x4 acc = acc0 + acc1;
    if (kv) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = q * 4 + r;
            if (row < Bm) {
                float* p = Cm + (size_t)row * ldc + kout;
                *p = acc[r] + (beta ? *p : 0.f);
            }
        }
    }
}

static int skinny_nn(const float* A, int lda, const float* W, int ldw, const float* bias, float* Cm, int ldc,
                     int Bm, int K, int Nc, int beta, hipStream_t st) {
    hipLaunchKernelGGL(skinny_nn_kernel, dim3((Nc + 63) / 64), dim3(256), 0, st, A, lda, W, ldw, bias, Cm, ldc, Bm, K,
                       Nc, beta);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}
static int skinny_nt(const float* A, int lda, const float* W, int ldw, float* Cm, int ldc, int Bm, int Nn, int Kc,
                     int beta, hipStream_t st) {
    hipLaunchKernelGGL(skinny_nt_kernel, dim3((Kc + 63) / 64), dim3(256), 0, st, A, lda, W, ldw, Cm, ldc, Bm, Nn, Kc,
                       beta);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}

// ------------------------------------------------------------------------------------------------
struct PrjDir {                 // mirrored in include/asr_amd.h (asr_prjhyper_dir)
    int reverse;
    const float* inner_in;      // (T,B,4N) x.W_in_to_inner + b_inner                 (:316)
    const float* outer_in;      // (T,B,4N) x.W_in_to_outer                           (:317)
    const float* Wio;           // (2F,4N) [W_inner_to_inner ; W_outer_to_inner]
    const float* Wi_p;          // (N,F)
    const float* S;             // (F,12N) [to_in | to_outer | to_bias]
    const float* s;             // (12N)
    const float* Wo_o;          // (F,4N)
    const float* Wo_p;          // (N,F)
    const float *ai, *bi_c, *ao, *bo_c;   // cell layer norm alpha / beta (N)
    float* hid;                 // (T,B,2F) [inner_hid | outer_hid]
    // saved for BPTT
    float *gi, *go;             // (T,B,4N) post-activation gates
    float *ci, *co;             // (T,B,N)  cell state (masked)
    float *ri, *ro;             // (T,B,N)  r state (masked)
    float *xhi, *xho;           // (T,B,N)  normalised cell
    float *rsi, *rso;           // (T,B)    rstd
    float* sc;                  // (T,B,12N) scale_outer
    float* a;                   // (T,B,4N)  outer_hid_prev . W_outer_to_outer
    // backward
    const float* dhid;          // (T,B,2F) dLoss/d hid
    float *dii, *doi;           // (T,B,4N) d inner gates ; d outer_input
    float* dsc;                 // (T,B,12N)
    float* da;                  // (T,B,4N)
    float* dh;                  // (T,B,2F) total d hid_t = [d inner_hid | d outer_hid]
    float *douti, *douto;       // (T,B,N) clipped d outcell
};

struct PrjArgs {
    int T, B, N, F, ndir;
    int b0, Btot;               // this call handles batch rows [b0, b0+B) of sequences with Btot rows per step
    float clip;
    const float* mask; int msb, mst;
    PrjDir d[2];
    int t[2], tp[2];            // time index of this step / of the previous forward step (-1: none)
    float* tg;                  // (ndir,B,4N) recurrent part of the inner gates
    float *g_c[2], *g_r[2];     // (ndir,B,N) carried d cell / d r state: [0] inner, [1] outer
    float* g_h;                 // (ndir,B,2F) d hid from the step processed before (the future)
};

__device__ __forceinline__ float block_sum(float v, float* sm) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    return sm[0] + sm[1] + sm[2] + sm[3];
}

#define PRJ_MAXU 8      // units per thread: N <= 2048

// one workgroup per (batch row, direction)
template <int OUTER>
__global__ __launch_bounds__(256) void prj_cell_fwd_kernel(PrjArgs A) {
    __shared__ float sm[4];
    const int b = blockIdx.x, dir = blockIdx.y;
    const PrjDir& D = A.d[dir];
    const int N = A.N, t = A.t[dir], tp = A.tp[dir];
    const size_t row = (size_t)t * A.Btot + A.b0 + b, rowp = (size_t)(tp < 0 ? 0 : tp) * A.Btot + A.b0 + b;
    const float m = A.mask[(size_t)(A.b0 + b) * A.msb + (size_t)t * A.mst];
    const float* alpha = OUTER ? D.ao : D.ai;
    const float* beta = OUTER ? D.bo_c : D.bi_c;
    float* gsv = (OUTER ? D.go : D.gi) + row * 4 * N;
    float* cst = OUTER ? D.co : D.ci;
    float* rst = OUTER ? D.ro : D.ri;
    float cn[PRJ_MAXU], og[PRJ_MAXU], cp[PRJ_MAXU];
    float s1 = 0.f;
#pragma unroll
    for (int u = 0; u < PRJ_MAXU; ++u) {
        const int n = threadIdx.x + u * 256;
        cn[u] = 0.f; og[u] = 0.f; cp[u] = 0.f;
        if (n < N) {
            float g[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (OUTER) {
                    const float* sc = D.sc + row * 12 * N;
                    g[k] = D.outer_in[row * 4 * N + k * N + n] * sc[k * N + n] +
                           D.a[row * 4 * N + k * N + n] * sc[4 * N + k * N + n] + sc[8 * N + k * N + n];   // :375-377
                } else {
                    g[k] = D.inner_in[row * 4 * N + k * N + n] + A.tg[((size_t)dir * A.B + b) * 4 * N + k * N + n];   // :329-331
                }
            }
            const float ig = sigmoidf_(g[0]), fg = sigmoidf_(g[1]), gc = tanhf_(g[2]), o = sigmoidf_(g[3]);
            cp[u] = tp < 0 ? 0.f : cst[rowp * N + n];
            cn[u] = fg * cp[u] + ig * gc;                                  // :355 | :401
            og[u] = o;
            gsv[n] = ig; gsv[N + n] = fg; gsv[2 * N + n] = gc; gsv[3 * N + n] = o;
            s1 += cn[u];
        }
    }
    const float mean = block_sum(s1, sm) / N;
    float s2 = 0.f;
#pragma unroll
    for (int u = 0; u < PRJ_MAXU; ++u) {
        const int n = threadIdx.x + u * 256;
        if (n < N) { const float dlt = cn[u] - mean; s2 += dlt * dlt; }
    }
    const float rstd = 1.0f / sqrtf(block_sum(s2, sm) / N + ASR_EPS);   // ln :9-12
    if (threadIdx.x == 0) (OUTER ? D.rso : D.rsi)[row] = rstd;
#pragma unroll
    for (int u = 0; u < PRJ_MAXU; ++u) {
        const int n = threadIdx.x + u * 256;
        if (n < N) {
            const float xh = (cn[u] - mean) * rstd;
            (OUTER ? D.xho : D.xhi)[row * N + n] = xh;
            const float r = og[u] * tanhf_(alpha[n] * xh + beta[n]);     // :357-363 | :403-409
            const float rp = tp < 0 ? 0.f : rst[rowp * N + n];
            cst[row * N + n] = m != 0.f ? cn[u] : cp[u];                   // :441-444
            rst[row * N + n] = m != 0.f ? r : rp;
        }
    }
}

template <int OUTER>
__global__ __launch_bounds__(256) void prj_cell_bwd_kernel(PrjArgs A) {
    __shared__ float sm[4];
    const int b = blockIdx.x, dir = blockIdx.y;
    const PrjDir& D = A.d[dir];
    const int N = A.N, t = A.t[dir], tp = A.tp[dir];
    const size_t row = (size_t)t * A.Btot + A.b0 + b, rowp = (size_t)(tp < 0 ? 0 : tp) * A.Btot + A.b0 + b;
    const bool on = A.mask[(size_t)(A.b0 + b) * A.msb + (size_t)t * A.mst] != 0.f;
    const float* alpha = OUTER ? D.ao : D.ai;
    const float* beta = OUTER ? D.bo_c : D.bi_c;
    const float* gsv = (OUTER ? D.go : D.gi) + row * 4 * N;
    const float* cst = OUTER ? D.co : D.ci;
    const float* xhat = (OUTER ? D.xho : D.xhi) + row * N;
    const float rstd = (OUTER ? D.rso : D.rsi)[row];
    float* g_c = A.g_c[OUTER] + ((size_t)dir * A.B + b) * N;
    float* g_r = A.g_r[OUTER] + ((size_t)dir * A.B + b) * N;
    float* dout_sv = (OUTER ? D.douto : D.douti) + row * N;
    float dxh[PRJ_MAXU], do_[PRJ_MAXU];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int u = 0; u < PRJ_MAXU; ++u) {
        const int n = threadIdx.x + u * 256;
        dxh[u] = 0.f; do_[u] = 0.f;
        if (n < N && on) {
            const float gr = g_r[n], o = gsv[3 * N + n];
            const float tc = tanhf_(alpha[n] * xhat[n] + beta[n]);
            do_[u] = gr * tc;
            const float dout = clipf_(gr * o * (1.f - tc * tc), A.clip);    // :358-361 | :404-407
            dout_sv[n] = dout;
            dxh[u] = dout * alpha[n];
            s1 += dxh[u]; s2 += dxh[u] * xhat[n];
        } else if (n < N) {
            dout_sv[n] = 0.f;
        }
    }
    const float m1 = block_sum(s1, sm) / N;
    const float m2 = block_sum(s2, sm) / N;
#pragma unroll
    for (int u = 0; u < PRJ_MAXU; ++u) {
        const int n = threadIdx.x + u * 256;
        if (n >= N) continue;
        float dg[4] = {0.f, 0.f, 0.f, 0.f};
        if (on) {
            const float ig = gsv[n], fg = gsv[N + n], gc = gsv[2 * N + n], o = gsv[3 * N + n];
            const float cprev = tp < 0 ? 0.f : cst[rowp * N + n];
            const float dct = g_c[n] + rstd * (dxh[u] - m1 - xhat[n] * m2);
            dg[0] = clipf_(dct * gc * ig * (1.f - ig), A.clip);
            dg[1] = clipf_(dct * cprev * fg * (1.f - fg), A.clip);
            dg[2] = clipf_(dct * ig * (1.f - gc * gc), A.clip);
            dg[3] = clipf_(do_[u] * o * (1.f - o), A.clip);
            g_c[n] = dct * fg;
            g_r[n] = 0.f;            // a switched-on row passes nothing straight through to t-1
        }                            // off: the state was copied, g_c / g_r (total) pass through unchanged
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const size_t i4 = row * 4 * N + k * N + n;
            if (OUTER) {
                const float* sc = D.sc + row * 12 * N;
                D.doi[i4] = dg[k] * sc[k * N + n];                              // d outer_input
                D.da[i4] = dg[k] * sc[4 * N + k * N + n];                       // d (outer_hid_prev.W)
                float* ds = D.dsc + row * 12 * N;
                ds[k * N + n] = dg[k] * D.outer_in[i4];
                ds[4 * N + k * N + n] = dg[k] * D.a[i4];
                ds[8 * N + k * N + n] = dg[k];
            } else {
                D.dii[i4] = dg[k];
            }
        }
    }
}

// dh[t] = dhid[t] + g_h  (both halves), for both directions
__global__ void prj_dh_kernel(PrjArgs A) {
    const int dir = blockIdx.y;
    const PrjDir& D = A.d[dir];
    const int n2 = A.B * 2 * A.F;
    const size_t base = ((size_t)A.t[dir] * A.Btot + A.b0) * 2 * A.F;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += gridDim.x * blockDim.x)
        D.dh[base + i] = D.dhid[base + i] + A.g_h[(size_t)dir * n2 + i];
}

struct asr_prjhyper_desc {      // mirrored in include/asr_amd.h
    int T, B, N, F, ndir;
    int b0, Btot;               // rows [b0, b0+B) of Btot; B <= 16 per call
    float grad_clip;
    const float* mask; int mask_stride_b, mask_stride_t;
    PrjDir dir[2];
};

extern "C" size_t asr_prjhyper_ws_bytes(int B, int N, int F, int ndir) {
    return ((size_t)ndir * B * (4 * N + 4 * N + 2 * F) + (size_t)B * 2 * F) * sizeof(float) + 256;
}

static int prj_check(const asr_prjhyper_desc* d, void* ws, size_t ws_bytes) {
    if (!d || !ws || d->T < 0 || d->B <= 0 || d->B > 16 || d->N <= 0 || d->F <= 0 || (d->N & 3) || (d->F & 3) ||
        d->N > 256 * PRJ_MAXU || d->ndir < 1 || d->ndir > 2 || !d->mask || d->b0 < 0 || d->b0 + d->B > d->Btot)
        return ASR_ERR_ARG;
    if (ws_bytes < asr_prjhyper_ws_bytes(d->B, d->N, d->F, d->ndir)) return ASR_ERR_WS;
    return ASR_OK;
}

static void prj_args(const asr_prjhyper_desc* d, void* ws, PrjArgs& A, float** zeros) {
    A.T = d->T; A.B = d->B; A.N = d->N; A.F = d->F; A.ndir = d->ndir; A.clip = d->grad_clip;
    A.b0 = d->b0; A.Btot = d->Btot;
    A.mask = d->mask; A.msb = d->mask_stride_b; A.mst = d->mask_stride_t;
    A.d[0] = d->dir[0]; A.d[1] = d->dir[d->ndir > 1 ? 1 : 0];
    float* p = (float*)ws;
    const size_t bn = (size_t)d->ndir * d->B * d->N;
    A.tg = p; p += 4 * bn;
    A.g_c[0] = p; p += bn; A.g_c[1] = p; p += bn; A.g_r[0] = p; p += bn; A.g_r[1] = p; p += bn;
    A.g_h = p; p += (size_t)d->ndir * d->B * 2 * d->F;
    *zeros = p;                                        // (B,2F) zeros: hid before the first step
}

// Batches of more than 16 rows are run by the caller in chunks of <= 16 rows (pointers advanced).
extern "C" int asr_prjhyper_fwd(const asr_prjhyper_desc* d, void* ws, size_t ws_bytes, hipStream_t st) {
    int rc = prj_check(d, ws, ws_bytes);
    if (rc) return rc;
    for (int k = 0; k < d->ndir; ++k) {
        const PrjDir& D = d->dir[k];
        if (!D.inner_in || !D.outer_in || !D.Wio || !D.Wi_p || !D.S || !D.s || !D.Wo_o || !D.Wo_p || !D.ai || !D.bi_c ||
            !D.ao || !D.bo_c || !D.hid || !D.gi || !D.go || !D.ci || !D.co || !D.ri || !D.ro || !D.xhi || !D.xho ||
            !D.rsi || !D.rso || !D.sc || !D.a)
            return ASR_ERR_ARG;
    }
    if (d->T == 0) return ASR_OK;
    PrjArgs A; float* zeros;
    prj_args(d, ws, A, &zeros);
    const int T = d->T, B = d->B, N = d->N, F = d->F;
    if (hipMemsetAsync(zeros, 0, (size_t)B * 2 * F * sizeof(float), st) != hipSuccess) return ASR_ERR_LAUNCH;
    for (int step = 0; step < T; ++step) {
        for (int k = 0; k < d->ndir; ++k) {
            A.t[k] = A.d[k].reverse ? T - 1 - step : step;
            A.tp[k] = step == 0 ? -1 : (A.d[k].reverse ? A.t[k] + 1 : A.t[k] - 1);
        }
        if (d->ndir == 1) { A.t[1] = A.t[0]; A.tp[1] = A.tp[0]; }
        for (int k = 0; k < d->ndir; ++k) {              // recurrent part of the inner gates
            const PrjDir& D = A.d[k];
            const float* hprev = A.tp[k] < 0 ? zeros : D.hid + ((size_t)A.tp[k] * d->Btot + d->b0) * 2 * F;
            if ((rc = skinny_nn(hprev, 2 * F, D.Wio, 4 * N, nullptr, A.tg + (size_t)k * B * 4 * N, 4 * N, B, 2 * F, 4 * N,
                                0, st))) return rc;
        }
        hipLaunchKernelGGL(prj_cell_fwd_kernel<0>, dim3(B, d->ndir), dim3(256), 0, st, A);
        ASR_CHECK_LAUNCH();
        for (int k = 0; k < d->ndir; ++k) {
            const PrjDir& D = A.d[k];
            float* hrow = D.hid + ((size_t)A.t[k] * d->Btot + d->b0) * 2 * F;
            const float* hprev = A.tp[k] < 0 ? zeros : D.hid + ((size_t)A.tp[k] * d->Btot + d->b0) * 2 * F;
            const size_t row0 = (size_t)A.t[k] * d->Btot + d->b0;
            if ((rc = skinny_nn(D.ri + row0 * N, N, D.Wi_p, F, nullptr, hrow, 2 * F, B, N, F, 0, st))) return rc;   // :364
            if ((rc = skinny_nn(hrow, 2 * F, D.S, 12 * N, D.s, D.sc + row0 * 12 * N, 12 * N, B, F, 12 * N, 0, st)))  // :367
                return rc;
            if ((rc = skinny_nn(hprev + F, 2 * F, D.Wo_o, 4 * N, nullptr, D.a + row0 * 4 * N, 4 * N, B, F, 4 * N, 0, st)))
                return rc;
        }
        hipLaunchKernelGGL(prj_cell_fwd_kernel<1>, dim3(B, d->ndir), dim3(256), 0, st, A);
        ASR_CHECK_LAUNCH();
        for (int k = 0; k < d->ndir; ++k) {
            const PrjDir& D = A.d[k];
            const size_t row0 = (size_t)A.t[k] * d->Btot + d->b0;
            if ((rc = skinny_nn(D.ro + row0 * N, N, D.Wo_p, F, nullptr, D.hid + row0 * 2 * F + F, 2 * F, B, N, F, 0, st)))  // :410
                return rc;
        }
    }
    return ASR_OK;
}

extern "C" int asr_prjhyper_bwd(const asr_prjhyper_desc* d, void* ws, size_t ws_bytes, hipStream_t st) {
    int rc = prj_check(d, ws, ws_bytes);
    if (rc) return rc;
    for (int k = 0; k < d->ndir; ++k) {
        const PrjDir& D = d->dir[k];
        if (!D.outer_in || !D.Wio || !D.Wi_p || !D.S || !D.Wo_o || !D.Wo_p || !D.ai || !D.bi_c || !D.ao || !D.bo_c ||
            !D.gi || !D.go || !D.ci || !D.co || !D.xhi || !D.xho || !D.rsi || !D.rso || !D.sc || !D.a || !D.dhid ||
            !D.dii || !D.doi || !D.dsc || !D.da || !D.dh || !D.douti || !D.douto)
            return ASR_ERR_ARG;
    }
    if (d->T == 0) return ASR_OK;
    PrjArgs A; float* zeros;
    prj_args(d, ws, A, &zeros);
    const int T = d->T, B = d->B, N = d->N, F = d->F;
    // carried gradients start at zero (tg is not used backward)
    if (hipMemsetAsync(A.g_c[0], 0, ((size_t)4 * d->ndir * B * N + (size_t)d->ndir * B * 2 * F) * sizeof(float), st) !=
        hipSuccess)
        return ASR_ERR_LAUNCH;
    for (int step = T - 1; step >= 0; --step) {
        for (int k = 0; k < d->ndir; ++k) {
            A.t[k] = A.d[k].reverse ? T - 1 - step : step;
            A.tp[k] = step == 0 ? -1 : (A.d[k].reverse ? A.t[k] + 1 : A.t[k] - 1);
        }
        if (d->ndir == 1) { A.t[1] = A.t[0]; A.tp[1] = A.tp[0]; }
        hipLaunchKernelGGL(prj_dh_kernel, dim3(4, d->ndir), dim3(256), 0, st, A);
        ASR_CHECK_LAUNCH();
        for (int k = 0; k < d->ndir; ++k) {              // d r_out (total) += d outer_hid . W_outer_prj^T
            const PrjDir& D = A.d[k];
            const float* dh = D.dh + ((size_t)A.t[k] * d->Btot + d->b0) * 2 * F;
            if ((rc = skinny_nt(dh + F, 2 * F, D.Wo_p, F, A.g_r[1] + (size_t)k * B * N, N, B, F, N, 1, st))) return rc;
        }
        hipLaunchKernelGGL(prj_cell_bwd_kernel<1>, dim3(B, d->ndir), dim3(256), 0, st, A);
        ASR_CHECK_LAUNCH();
        for (int k = 0; k < d->ndir; ++k) {
            const PrjDir& D = A.d[k];
            float* dh = D.dh + ((size_t)A.t[k] * d->Btot + d->b0) * 2 * F;
            const size_t row0 = (size_t)A.t[k] * d->Btot + d->b0;
            // d inner_hid_t += d scale_outer . S^T ; d r_in (total) += d inner_hid . W_inner_prj^T
            if ((rc = skinny_nt(D.dsc + row0 * 12 * N, 12 * N, D.S, 12 * N, dh, 2 * F, B, 12 * N, F, 1, st))) return rc;
            if ((rc = skinny_nt(dh, 2 * F, D.Wi_p, F, A.g_r[0] + (size_t)k * B * N, N, B, F, N, 1, st))) return rc;
        }
        hipLaunchKernelGGL(prj_cell_bwd_kernel<0>, dim3(B, d->ndir), dim3(256), 0, st, A);
        ASR_CHECK_LAUNCH();
        for (int k = 0; k < d->ndir; ++k) {              // d hid_{t-1} = d inner gates . Wio^T ; oh half += da . Wo_o^T
            const PrjDir& D = A.d[k];
            const size_t row0 = (size_t)A.t[k] * d->Btot + d->b0;
            float* gh = A.g_h + (size_t)k * B * 2 * F;
            if ((rc = skinny_nt(D.dii + row0 * 4 * N, 4 * N, D.Wio, 4 * N, gh, 2 * F, B, 4 * N, 2 * F, 0, st))) return rc;
            if ((rc = skinny_nt(D.da + row0 * 4 * N, 4 * N, D.Wo_o, 4 * N, gh + F, 2 * F, B, 4 * N, F, 1, st))) return rc;
        }
    }
    return ASR_OK;
}
