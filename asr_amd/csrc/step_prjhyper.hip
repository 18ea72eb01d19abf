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
#include "scan_common.h"

#include "step_skinny.h"

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
};

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

// NU = units per thread (N <= 256 NU).  Every load is unconditional (out-of-range units read unit N-1, rows
// that are switched off are masked by select on the values): hipcc waits for a load at the first control-flow
// join behind it, which made the guarded version of the outer cell take 19 us per step.
template <int OUTER, int NU>
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
    const float* cpv = (OUTER ? D.co : D.ci) + rowp * N;
    const float* xhat = (OUTER ? D.xho : D.xhi) + row * N;
    const float rstd = (OUTER ? D.rso : D.rsi)[row];
    float* g_c = A.g_c[OUTER] + ((size_t)dir * A.B + b) * N;
    float* g_r = A.g_r[OUTER] + ((size_t)dir * A.B + b) * N;
    float* dout_sv = (OUTER ? D.douto : D.douti) + row * N;
    const float* sc = D.sc + row * 12 * N;
    const float* oin = D.outer_in + row * 4 * N;
    const float* av = D.a + row * 4 * N;
    float gr[NU], gcar[NU], xh[NU], al[NU], be[NU], g4[NU][4], cpr[NU];
    float s_in[NU][4], s_ou[NU][4], x_in[NU][4], x_a[NU][4];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int n = min((int)threadIdx.x + u * 256, N - 1);
        gr[u] = g_r[n]; gcar[u] = g_c[n]; xh[u] = xhat[n]; al[u] = alpha[n]; be[u] = beta[n];
        cpr[u] = cpv[n];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            g4[u][k] = gsv[k * N + n];
            if (OUTER) {
                s_in[u][k] = sc[k * N + n]; s_ou[u][k] = sc[4 * N + k * N + n];
                x_in[u][k] = oin[k * N + n]; x_a[u][k] = av[k * N + n];
            }
        }
    }
    float dxh[NU], do_[NU];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int n = threadIdx.x + u * 256;
        const bool act = on && n < N;
        const float tc = tanhf_(al[u] * xh[u] + be[u]);
        do_[u] = act ? gr[u] * tc : 0.f;
        const float dout = act ? clipf_(gr[u] * g4[u][3] * (1.f - tc * tc), A.clip) : 0.f;    // :358-361 | :404-407
        if (n < N) dout_sv[n] = dout;
        dxh[u] = dout * al[u];
        s1 += dxh[u]; s2 += dxh[u] * xh[u];
    }
    const float m1 = block_sum(s1, sm) / N;
    const float m2 = block_sum(s2, sm) / N;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int n = threadIdx.x + u * 256;
        if (n >= N) continue;                            // stores only below
        float dg[4] = {0.f, 0.f, 0.f, 0.f};
        const float ig = g4[u][0], fg = g4[u][1], gc = g4[u][2], o = g4[u][3];
        const float cprev = tp < 0 ? 0.f : cpr[u];
        const float dct = gcar[u] + rstd * (dxh[u] - m1 - xh[u] * m2);
        if (on) {
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
                D.doi[i4] = dg[k] * s_in[u][k];                                 // d outer_input
                D.da[i4] = dg[k] * s_ou[u][k];                                  // d (outer_hid_prev.W)
                float* ds = D.dsc + row * 12 * N;
                ds[k * N + n] = dg[k] * x_in[u][k];
                ds[4 * N + k * N + n] = dg[k] * x_a[u][k];
                ds[8 * N + k * N + n] = dg[k];
            } else {
                D.dii[i4] = dg[k];
            }
        }
    }
}

template <int OUTER>
static void prj_cell_bwd_launch(const PrjArgs& A, int ndir, hipStream_t st) {
    const dim3 grid(A.B, ndir), block(256);
    if (A.N <= 256) hipLaunchKernelGGL((prj_cell_bwd_kernel<OUTER, 1>), grid, block, 0, st, A);
    else if (A.N <= 512) hipLaunchKernelGGL((prj_cell_bwd_kernel<OUTER, 2>), grid, block, 0, st, A);
    else if (A.N <= 1024) hipLaunchKernelGGL((prj_cell_bwd_kernel<OUTER, 4>), grid, block, 0, st, A);
    else hipLaunchKernelGGL((prj_cell_bwd_kernel<OUTER, PRJ_MAXU>), grid, block, 0, st, A);
}

struct asr_prjhyper_desc {      // mirrored in include/asr_amd.h
    int T, B, N, F, ndir;
    int b0, Btot;               // rows [b0, b0+B) of Btot; B <= 16 per call
    float grad_clip;
    const float* mask; int mask_stride_b, mask_stride_t;
    PrjDir dir[2];
};

extern "C" size_t asr_prjhyper_ws_bytes(int B, int N, int F, int ndir) {
    return ((size_t)ndir * B * (4 * N + 4 * N + 2 * F) + (size_t)B * 2 * F + SK_PART_FLOATS) * sizeof(float) +
           SK_TICKETS * sizeof(unsigned) + 256;
}

static int prj_check(const asr_prjhyper_desc* d, void* ws, size_t ws_bytes) {
    if (!d || !ws || d->T < 0 || d->B <= 0 || d->B > 16 || d->N <= 0 || d->F <= 0 || (d->N & 3) || (d->F & 3) ||
        d->N > 256 * PRJ_MAXU || d->ndir < 1 || d->ndir > 2 || !d->mask || d->b0 < 0 || d->b0 + d->B > d->Btot ||
        ((uintptr_t)ws & 15))
        return ASR_ERR_ARG;
    if (ws_bytes < asr_prjhyper_ws_bytes(d->B, d->N, d->F, d->ndir)) return ASR_ERR_WS;
    return ASR_OK;
}

struct PrjHost {
    PrjArgs A;
    float* zeros;               // (B,2F) zeros: hid before the first step
    float* part; unsigned* tickets;
    int nd;
    hipStream_t st;
};

static int prj_args(const asr_prjhyper_desc* d, void* ws, hipStream_t st, PrjHost& H) {
    PrjArgs& A = H.A;
    A.T = d->T; A.B = d->B; A.N = d->N; A.F = d->F; A.ndir = d->ndir; A.clip = d->grad_clip;
    A.b0 = d->b0; A.Btot = d->Btot;
    A.mask = d->mask; A.msb = d->mask_stride_b; A.mst = d->mask_stride_t;
    A.d[0] = d->dir[0]; A.d[1] = d->dir[d->ndir > 1 ? 1 : 0];
    float* p = (float*)ws;
    const size_t bn = (size_t)d->ndir * d->B * d->N;
    A.tg = p; p += 4 * bn;
    A.g_c[0] = p; p += bn; A.g_c[1] = p; p += bn; A.g_r[0] = p; p += bn; A.g_r[1] = p; p += bn;
    p += (size_t)d->ndir * d->B * 2 * d->F;             // (spare)
    H.zeros = p; p += (size_t)d->B * 2 * d->F;
    H.part = p; p += SK_PART_FLOATS;
    H.tickets = (unsigned*)p;
    H.nd = d->ndir; H.st = st;
    // the tickets start at zero (every launch leaves them at zero again)
    if (hipMemsetAsync(H.tickets, 0, SK_TICKETS * sizeof(unsigned), st) != hipSuccess) return ASR_ERR_LAUNCH;
    return ASR_OK;
}

// one skinny product per direction, one launch: pointers by direction
static int prj_mm(PrjHost& H, bool nt, const float* const A[2], int lda, const float* const W[2], int ldw,
                  const float* const bias[2], float* const Cm[2], int ldc, int K, int Nc, int beta) {
    SkinnyArgs S;
    for (int k = 0; k < SK_MAXP; ++k) {
        const int kk = k < H.nd ? k : 0;
        S.p[k].A = A[kk]; S.p[k].W = W[kk]; S.p[k].bias = bias ? bias[kk] : nullptr; S.p[k].C = Cm[kk];
    }
    S.lda = lda; S.ldw = ldw; S.ldc = ldc; S.Bm = H.A.B; S.K = K; S.Nc = Nc; S.beta = beta; S.nsplit = 1;
    S.part = H.part; S.tickets = H.tickets;
    return skinny(nt, S, H.nd, H.st);
}

static void prj_times(const asr_prjhyper_desc* d, PrjArgs& A, int step) {
    const int T = d->T;
    for (int k = 0; k < d->ndir; ++k) {
        A.t[k] = A.d[k].reverse ? T - 1 - step : step;
        A.tp[k] = step == 0 ? -1 : (A.d[k].reverse ? A.t[k] + 1 : A.t[k] - 1);
    }
    if (d->ndir == 1) { A.t[1] = A.t[0]; A.tp[1] = A.tp[0]; }
}

// Batches of more than 16 rows are run by the caller in chunks of <= 16 rows (b0 advanced).
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
    PrjHost H;
    if ((rc = prj_args(d, ws, st, H))) return rc;
    PrjArgs& A = H.A;
    const int T = d->T, B = d->B, N = d->N, F = d->F;
    if (hipMemsetAsync(H.zeros, 0, (size_t)B * 2 * F * sizeof(float), st) != hipSuccess) return ASR_ERR_LAUNCH;
    for (int step = 0; step < T; ++step) {
        prj_times(d, A, step);
        const float *hprev[2], *ohprev[2], *ri[2], *ro[2], *hrow_c[2];
        const float *Wio[2], *Wip[2], *Sm[2], *sb[2], *Woo[2], *Wop[2];
        float *tg[2], *hrow[2], *ohrow[2], *sc[2], *a[2];
        for (int k = 0; k < 2; ++k) {
            const int kk = k < d->ndir ? k : 0;
            const PrjDir& D = A.d[kk];
            const size_t row0 = (size_t)A.t[kk] * d->Btot + d->b0;
            hprev[k] = A.tp[kk] < 0 ? H.zeros : D.hid + ((size_t)A.tp[kk] * d->Btot + d->b0) * 2 * F;
            ohprev[k] = hprev[k] + F;
            hrow[k] = D.hid + row0 * 2 * F; hrow_c[k] = hrow[k]; ohrow[k] = hrow[k] + F;
            ri[k] = D.ri + row0 * N; ro[k] = D.ro + row0 * N;
            sc[k] = D.sc + row0 * 12 * N; a[k] = D.a + row0 * 4 * N;
            tg[k] = A.tg + (size_t)kk * B * 4 * N;
            Wio[k] = D.Wio; Wip[k] = D.Wi_p; Sm[k] = D.S; sb[k] = D.s; Woo[k] = D.Wo_o; Wop[k] = D.Wo_p;
        }
        // recurrent part of the inner gates                                                      :329-331
        if ((rc = prj_mm(H, false, hprev, 2 * F, Wio, 4 * N, nullptr, tg, 4 * N, 2 * F, 4 * N, 0))) return rc;
        hipLaunchKernelGGL(prj_cell_fwd_kernel<0>, dim3(B, d->ndir), dim3(256), 0, st, A);
        ASR_CHECK_LAUNCH();
        if ((rc = prj_mm(H, false, ri, N, Wip, F, nullptr, hrow, 2 * F, N, F, 0))) return rc;            // :364
        if ((rc = prj_mm(H, false, hrow_c, 2 * F, Sm, 12 * N, sb, sc, 12 * N, F, 12 * N, 0))) return rc;  // :367-370
        if ((rc = prj_mm(H, false, ohprev, 2 * F, Woo, 4 * N, nullptr, a, 4 * N, F, 4 * N, 0))) return rc;  // :376
        hipLaunchKernelGGL(prj_cell_fwd_kernel<1>, dim3(B, d->ndir), dim3(256), 0, st, A);
        ASR_CHECK_LAUNCH();
        if ((rc = prj_mm(H, false, ro, N, Wop, F, nullptr, ohrow, 2 * F, N, F, 0))) return rc;           // :410
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
    PrjHost H;
    if ((rc = prj_args(d, ws, st, H))) return rc;
    PrjArgs& A = H.A;
    const int T = d->T, B = d->B, N = d->N, F = d->F;
    // carried gradients start at zero (tg is not used backward)
    if (hipMemsetAsync(A.g_c[0], 0, (size_t)4 * d->ndir * B * N * sizeof(float), st) !=
        hipSuccess)
        return ASR_ERR_LAUNCH;
    // dh starts as dhid (rows of this chunk, every t); each step adds its recurrent part to dh of the step before
    for (int k = 0; k < d->ndir; ++k) {
        const PrjDir& D = d->dir[k];
        if (hipMemcpy2DAsync(D.dh + (size_t)d->b0 * 2 * F, (size_t)d->Btot * 2 * F * sizeof(float),
                             D.dhid + (size_t)d->b0 * 2 * F, (size_t)d->Btot * 2 * F * sizeof(float),
                             (size_t)B * 2 * F * sizeof(float), T, hipMemcpyDeviceToDevice, st) != hipSuccess)
            return ASR_ERR_LAUNCH;
    }
    for (int step = T - 1; step >= 0; --step) {
        prj_times(d, A, step);
        const float *dh_i[2], *dh_o[2], *dsc[2], *dii[2], *da[2];
        const float *Wio[2], *Wip[2], *Sm[2], *Woo[2], *Wop[2];
        float *dh_iw[2], *gr0[2], *gr1[2], *gh[2], *gh_o[2];
        for (int k = 0; k < 2; ++k) {
            const int kk = k < d->ndir ? k : 0;
            const PrjDir& D = A.d[kk];
            const size_t row0 = (size_t)A.t[kk] * d->Btot + d->b0;
            dh_iw[k] = D.dh + row0 * 2 * F; dh_i[k] = dh_iw[k]; dh_o[k] = dh_iw[k] + F;
            dsc[k] = D.dsc + row0 * 12 * N; dii[k] = D.dii + row0 * 4 * N; da[k] = D.da + row0 * 4 * N;
            gr0[k] = A.g_r[0] + (size_t)kk * B * N; gr1[k] = A.g_r[1] + (size_t)kk * B * N;
            gh[k] = D.dh + ((size_t)(A.tp[kk] < 0 ? A.t[kk] : A.tp[kk]) * d->Btot + d->b0) * 2 * F; gh_o[k] = gh[k] + F;
            Wio[k] = D.Wio; Wip[k] = D.Wi_p; Sm[k] = D.S; Woo[k] = D.Wo_o; Wop[k] = D.Wo_p;
        }
        // d r_out (total) += d outer_hid . W_outer_prj^T
        if ((rc = prj_mm(H, true, dh_o, 2 * F, Wop, F, nullptr, gr1, N, F, N, 1))) return rc;
        prj_cell_bwd_launch<1>(A, d->ndir, st);
        ASR_CHECK_LAUNCH();
        // d inner_hid_t += d scale_outer . S^T ; d r_in (total) += d inner_hid . W_inner_prj^T
        if ((rc = prj_mm(H, true, dsc, 12 * N, Sm, 12 * N, nullptr, dh_iw, 2 * F, 12 * N, F, 1))) return rc;
        if ((rc = prj_mm(H, true, dh_i, 2 * F, Wip, F, nullptr, gr0, N, F, N, 1))) return rc;
        prj_cell_bwd_launch<0>(A, d->ndir, st);
        ASR_CHECK_LAUNCH();
        // d hid_{t-1} (already holding dhid_{t-1}) += d inner gates . Wio^T ; its outer half += da . Wo_o^T
        if (step > 0) {
            if ((rc = prj_mm(H, true, dii, 4 * N, Wio, 4 * N, nullptr, gh, 2 * F, 4 * N, 2 * F, 1))) return rc;
            if ((rc = prj_mm(H, true, da, 4 * N, Woo, 4 * N, nullptr, gh_o, 2 * F, 4 * N, F, 1))) return rc;
        }
    }
    return ASR_OK;
}
