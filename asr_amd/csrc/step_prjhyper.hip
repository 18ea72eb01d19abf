// ProjectionHyperLSTMLayer (libs/lasagne_libs/hyper_layers.py:14-523; step :319-417, step_masked
// :419-456, scan :492-519), first HIP path: one time step = a short chain of small kernels, the loop
// over T runs on the host inside this library (no per-step Python, no allocation, one stream).
//
// Why not a mode of the persistent scans: both cells are N units wide with cell layer norm and a
// projection to F, coupled inside the step through three scale maps (F x 12N); folded weights are
// ~190 KB per workgroup at N = 512 and the step has 4-6 all-to-all exchanges (DESIGN.md 7 prices a persistent kernel).
//
// Both projections are folded into their consumers (exact algebra, gradients un-folded by the caller):
//   [tg | a]_t = [r_i | r_o]_{t-1} . Wf,  Wf (2N,8N) = [[W_ip.W_io[:F], 0], [W_op.W_io[F:], W_op.W_oo]]   (:329-331,:376)
//   sc_t       = r_i,t . W3 + s,          W3 (N,12N) = W_ip.S                                              (:364-370)
// so the carried state is r = o*tanh(outcell) of both cells ((T,B,2N), masked like the reference's hid: hid_t =
// r_t . W_prj is linear, so switching r is switching hid, :441-444), hid = R.W_prj is one GEMM after the loop, and a
// step is 2 skinny products + 2 cell kernels in each direction of time (7 + 7 before the fold: 334 -> see DESIGN.md).
//
// Layouts here are the reference's: gate vectors are gate-major [i | f | c | o] blocks of N, sequences
// time-major (T,B,.).
#include "scan_common.h"
#include <stdlib.h>
#include "step_skinny.h"

// ------------------------------------------------------------------------------------------------
struct PrjDir {                 // mirrored in include/asr_amd.h (asr_prjhyper_dir)
    int reverse;
    const float* inner_in;      // (T,B,4N) x.W_in_to_inner + b_inner                 (:316)
    const float* outer_in;      // (T,B,4N) x.W_in_to_outer                           (:317)
    const float* Wf;            // (2N,8N) folded recurrent matrix (above)
    const float* W3;            // (N,12N) W_inner_prj . [to_in | to_outer | to_bias]
    const float* s;             // (12N)
    const float *ai, *bi_c, *ao, *bo_c;   // cell layer norm alpha / beta (N)
    float* r;                   // (T,B,2N) [r_inner | r_outer] state (masked)
    // saved for BPTT
    float *gi, *go;             // (T,B,4N) post-activation gates
    float *ci, *co;             // (T,B,N)  cell state (masked)
    float *xhi, *xho;           // (T,B,N)  normalised cell
    float *rsi, *rso;           // (T,B)    rstd
    float* sc;                  // (T,B,12N) scale_outer
    float* a;                   // (T,B,4N) outer_hid_prev . W_outer_to_outer
    // backward
    float* dr;                  // (T,B,2N) d r_t from outside the loop (dhid . W_prj^T) on entry; rows that are switched
                                // off add what they pass on to the row of the step before (work)
    float* dta;                 // (T,B,8N) [d inner gate pre-activations | d a]
    float* doi;                 // (T,B,4N) d outer_input
    float* dsc;                 // (T,B,12N)
    float *douti, *douto;       // (T,B,N) clipped d outcell
};

struct PrjArgs {
    int T, B, N, F, ndir;
    int b0, Btot;               // this call handles batch rows [b0, b0+B) of sequences with Btot rows per step
    float clip;
    const float* mask; int msb, mst;
    PrjDir d[2];
    int t[2], tp[2];            // time index of this step / of the previous forward step (-1: none)
    float* g_c[2];              // (ndir,B,N) carried d cell: [0] inner, [1] outer
    // partial blocks of the skinny products ([split][16 rows][columns], step_skinny.h raw mode), summed by the cell
    // kernel that consumes them: forward pf = r_{t-1}.Wf (8N columns), p3 = r_inner,t.W3 (12N); backward
    // pbf = dta_{t+1}.Wf^T (2N: what step t+1 hands to d r_t), pb3 = dsc_t.W3^T (N)
    float *pf[2], *p3[2], *pbf[2], *pb3[2];
    int ns_f, ns_3, ns_bf, ns_b3;
    int Bpad;                   // rows of a partial block: B rounded up to whole 16-row chunks
    int first;                  // backward: the first step processed (nothing handed over from a later step)
};

// Sums of partial blocks: four blocks' loads are in flight together, every load unconditional (a slot past the last
// split re-reads the last one with weight 0: hipcc waits for a load at the first control-flow join behind it).
#define PART_SUM_BEGIN(ns) for (int sp0 = 0; sp0 < (ns); sp0 += 4) { _Pragma("unroll") for (int i4 = 0; i4 < 4; ++i4) { \
    const int sp = min(sp0 + i4, (ns) - 1); const float pw = sp0 + i4 < (ns) ? 1.f : 0.f;
#define PART_SUM_END } }

// one workgroup per (batch row, direction); NU = units per thread (N <= 256 NU); loads are unconditional
// (out-of-range units read unit N-1), stores guarded
template <int OUTER, int NU>
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
    float* rst = D.r + (OUTER ? N : 0);              // row stride 2N
    int nn[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) nn[u] = min((int)threadIdx.x + u * 256, N - 1);
    // recurrent products of the step: sums of the partial blocks
    float rec[NU][4];                                // inner: recurrent part of the gates (:329-331); outer: a (:376)
    float scv[NU][12];                               // outer: scale_outer (:364-370)
#pragma unroll
    for (int u = 0; u < NU; ++u) {
#pragma unroll
        for (int k = 0; k < 4; ++k) rec[u][k] = 0.f;
#pragma unroll
        for (int k = 0; k < 12; ++k) scv[u][k] = OUTER ? D.s[k * N + nn[u]] : 0.f;
    }
    if (tp >= 0) {
        const float* pf = A.pf[dir] + (size_t)b * 8 * N + (OUTER ? 4 * N : 0);
        PART_SUM_BEGIN(A.ns_f)
            const float* pp = pf + (size_t)sp * A.Bpad * 8 * N;
#pragma unroll
            for (int u = 0; u < NU; ++u) {
#pragma unroll
                for (int k = 0; k < 4; ++k) rec[u][k] += pw * pp[k * N + nn[u]];
            }
        PART_SUM_END
    }
    if (OUTER) {
        const float* p3 = A.p3[dir] + (size_t)b * 12 * N;
        PART_SUM_BEGIN(A.ns_3)
            const float* pp = p3 + (size_t)sp * A.Bpad * 12 * N;
#pragma unroll
            for (int u = 0; u < NU; ++u) {
#pragma unroll
                for (int k = 0; k < 12; ++k) scv[u][k] += pw * pp[k * N + nn[u]];
            }
        PART_SUM_END
    }
    float cn[NU], og[NU], cp[NU], xin[NU][4];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        cp[u] = cst[rowp * N + nn[u]];
#pragma unroll
        for (int k = 0; k < 4; ++k) xin[u][k] = (OUTER ? D.outer_in : D.inner_in)[row * 4 * N + k * N + nn[u]];
    }
    float s1 = 0.f;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int n = threadIdx.x + u * 256;
        float g[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            g[k] = OUTER ? xin[u][k] * scv[u][k] + rec[u][k] * scv[u][4 + k] + scv[u][8 + k]      // :375-377
                         : xin[u][k] + rec[u][k];                                                   // :329-331
        const float ig = sigmoidf_(g[0]), fg = sigmoidf_(g[1]), gc = tanhf_(g[2]), o = sigmoidf_(g[3]);
        if (tp < 0) cp[u] = 0.f;
        cn[u] = fg * cp[u] + ig * gc;                                  // :355 | :401
        og[u] = o;
        if (n < N) {
            gsv[n] = ig; gsv[N + n] = fg; gsv[2 * N + n] = gc; gsv[3 * N + n] = o;
            if (OUTER) {
#pragma unroll
                for (int k = 0; k < 4; ++k) D.a[row * 4 * N + k * N + n] = rec[u][k];
#pragma unroll
                for (int k = 0; k < 12; ++k) D.sc[row * 12 * N + k * N + n] = scv[u][k];
            }
            s1 += cn[u];
        }
    }
    const float mean = block_sum(s1, sm) / N;
    float s2 = 0.f;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int n = threadIdx.x + u * 256;
        if (n < N) { const float dlt = cn[u] - mean; s2 += dlt * dlt; }
    }
    const float rstd = 1.0f / sqrtf(block_sum(s2, sm) / N + ASR_EPS);   // ln :9-12
    if (threadIdx.x == 0) (OUTER ? D.rso : D.rsi)[row] = rstd;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int n = threadIdx.x + u * 256;
        if (n < N) {
            const float xh = (cn[u] - mean) * rstd;
            (OUTER ? D.xho : D.xhi)[row * N + n] = xh;
            const float r = og[u] * tanhf_(alpha[n] * xh + beta[n]);     // :357-363 | :403-409
            const float rp = tp < 0 ? 0.f : rst[rowp * 2 * N + n];
            cst[row * N + n] = m != 0.f ? cn[u] : cp[u];                   // :441-444
            rst[row * 2 * N + n] = m != 0.f ? r : rp;
        }
    }
}

template <int OUTER>
static void prj_cell_fwd_launch(const PrjArgs& A, int ndir, hipStream_t st) {
    const dim3 grid(A.B, ndir), block(256);
    if (A.N <= 256) hipLaunchKernelGGL((prj_cell_fwd_kernel<OUTER, 1>), grid, block, 0, st, A);
    else if (A.N <= 512) hipLaunchKernelGGL((prj_cell_fwd_kernel<OUTER, 2>), grid, block, 0, st, A);
    else if (A.N <= 1024) hipLaunchKernelGGL((prj_cell_fwd_kernel<OUTER, 4>), grid, block, 0, st, A);
    else hipLaunchKernelGGL((prj_cell_fwd_kernel<OUTER, PRJ_MAXU>), grid, block, 0, st, A);
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
    const float* dr_t = D.dr + row * 2 * N + (OUTER ? N : 0);
    float* dr_p = D.dr + rowp * 2 * N + (OUTER ? N : 0);
    float* dout_sv = (OUTER ? D.douto : D.douti) + row * N;
    const float* sc = D.sc + row * 12 * N;
    const float* oin = D.outer_in + row * 4 * N;
    const float* av = D.a + row * 4 * N;
    float gr[NU], gcar[NU], xh[NU], al[NU], be[NU], g4[NU][4], cpr[NU];
    float s_in[NU][4], s_ou[NU][4], x_in[NU][4], x_a[NU][4];
    int nn[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int n = nn[u] = min((int)threadIdx.x + u * 256, N - 1);
        gr[u] = dr_t[n]; gcar[u] = g_c[n]; xh[u] = xhat[n]; al[u] = alpha[n]; be[u] = beta[n];
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
    // total d r_t of this cell: + what the recurrent products hand over (their partial blocks)
    if (!A.first) {
        const float* pb = A.pbf[dir] + (size_t)b * 2 * N + (OUTER ? N : 0);
        PART_SUM_BEGIN(A.ns_bf)
#pragma unroll
            for (int u = 0; u < NU; ++u) gr[u] += pw * pb[(size_t)sp * A.Bpad * 2 * N + nn[u]];
        PART_SUM_END
    }
    if (!OUTER) {
        const float* pb = A.pb3[dir] + (size_t)b * N;
        PART_SUM_BEGIN(A.ns_b3)
#pragma unroll
            for (int u = 0; u < NU; ++u) gr[u] += pw * pb[(size_t)sp * A.Bpad * N + nn[u]];
        PART_SUM_END
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
        } else if (tp >= 0) {
            dr_p[n] += gr[u];        // off: the state was copied, its gradient passes on (g_c stays as it is)
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const size_t i4 = row * 4 * N + k * N + n;
            if (OUTER) {
                D.doi[i4] = dg[k] * s_in[u][k];                                 // d outer_input
                D.dta[row * 8 * N + 4 * N + k * N + n] = dg[k] * s_ou[u][k];     // d (outer_hid_prev.W)
                float* ds = D.dsc + row * 12 * N;
                ds[k * N + n] = dg[k] * x_in[u][k];
                ds[4 * N + k * N + n] = dg[k] * x_a[u][k];
                ds[8 * N + k * N + n] = dg[k];
            } else {
                D.dta[row * 8 * N + k * N + n] = dg[k];
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
    int b0, Btot;               // rows [b0, b0+B) of Btot; B <= 64 per call
    float grad_clip;
    const float* mask; int mask_stride_b, mask_stride_t;
    PrjDir dir[2];
};

// splits of the four products (functions of B, N and ndir only, so the workspace can be sized before the call)
#define PRJ_MAXB 64             // batch rows per call: 4 row chunks of 16 in every launch
struct PrjSplits { int f, w3, bf, b3; };
static int prj_target_wgs() {
    static int v = 0;
    if (!v) {
        const char* e = getenv("ASR_SK_WGS");            // tuning aid; default below
        v = e ? atoi(e) : 0;
        if (v < 64 || v > 4096) v = 512;
    }
    return v;
}
static PrjSplits prj_splits(int B, int N, int ndir) {
    const int tw = prj_target_wgs(), npz = ndir * ((B + 15) / 16);
    PrjSplits s;
    s.f = skinny_nsplit(false, N, 8 * N, npz, tw);        // the shorter of the two reduction ranges (zero block)
    s.w3 = skinny_nsplit(false, N, 12 * N, npz, tw);
    s.bf = skinny_nsplit(true, 4 * N, 2 * N, npz, tw);
    s.b3 = skinny_nsplit(true, 12 * N, N, npz, tw);
    return s;
}
static size_t prj_part_floats(int B, int N, int ndir) {
    const PrjSplits s = prj_splits(B, N, ndir);
    const size_t bp = (size_t)((B + 15) / 16) * 16;
    const size_t fw = (size_t)s.f * bp * 8 * N + (size_t)s.w3 * bp * 12 * N;
    const size_t bw = (size_t)s.bf * bp * 2 * N + (size_t)s.b3 * bp * N;
    return (fw > bw ? fw : bw) * ndir;
}

extern "C" size_t asr_prjhyper_ws_bytes(int B, int N, int F, int ndir) {
    (void)F;
    return ((size_t)2 * ndir * B * N + prj_part_floats(B, N, ndir)) * sizeof(float) + 256;
}

static int prj_check(const asr_prjhyper_desc* d, void* ws, size_t ws_bytes) {
    if (!d || !ws || d->T < 0 || d->B <= 0 || d->B > PRJ_MAXB || d->N <= 0 || d->F <= 0 || (d->N & 3) || (d->F & 3) ||
        d->N > 256 * PRJ_MAXU || d->ndir < 1 || d->ndir > 2 || !d->mask || d->b0 < 0 || d->b0 + d->B > d->Btot ||
        ((uintptr_t)ws & 15))
        return ASR_ERR_ARG;
    if (ws_bytes < asr_prjhyper_ws_bytes(d->B, d->N, d->F, d->ndir)) return ASR_ERR_WS;
    return ASR_OK;
}

struct PrjHost {
    PrjArgs A;
    int nd;
    hipStream_t st;
};

static void prj_args(const asr_prjhyper_desc* d, void* ws, hipStream_t st, bool backward, PrjHost& H) {
    PrjArgs& A = H.A;
    A.T = d->T; A.B = d->B; A.N = d->N; A.F = d->F; A.ndir = d->ndir; A.clip = d->grad_clip;
    A.b0 = d->b0; A.Btot = d->Btot;
    A.mask = d->mask; A.msb = d->mask_stride_b; A.mst = d->mask_stride_t;
    A.d[0] = d->dir[0]; A.d[1] = d->dir[d->ndir > 1 ? 1 : 0];
    float* p = (float*)ws;
    const int N = d->N;
    const size_t bn = (size_t)d->ndir * d->B * N;
    A.g_c[0] = p; p += bn; A.g_c[1] = p; p += bn;
    p = (float*)(((uintptr_t)p + 255) & ~(uintptr_t)255);
    const PrjSplits s = prj_splits(d->B, N, d->ndir);
    A.ns_f = s.f; A.ns_3 = s.w3; A.ns_bf = s.bf; A.ns_b3 = s.b3; A.first = 0;
    const size_t bp = (size_t)((d->B + 15) / 16) * 16;
    A.Bpad = (int)bp;
    for (int k = 0; k < 2; ++k) {
        const int kk = k < d->ndir ? k : 0;
        if (!backward) {
            float* q = p + (size_t)kk * ((size_t)s.f * bp * 8 * N + (size_t)s.w3 * bp * 12 * N);
            A.pf[k] = q; A.p3[k] = q + (size_t)s.f * bp * 8 * N; A.pbf[k] = A.pb3[k] = nullptr;
        } else {
            float* q = p + (size_t)kk * ((size_t)s.bf * bp * 2 * N + (size_t)s.b3 * bp * N);
            A.pbf[k] = q; A.pb3[k] = q + (size_t)s.bf * bp * 2 * N; A.pf[k] = A.p3[k] = nullptr;
        }
    }
    H.nd = d->ndir; H.st = st;
}

// one skinny product per direction, one launch: pointers by direction; the partial blocks go to Cp (raw mode).
// zfrom / zk0 / zk1: the zero block of Wf (step_skinny.h), zfrom < 0 for a dense matrix
static int prj_mm(PrjHost& H, bool nt, const float* const A[2], int lda, const float* const W[2], int ldw,
                  float* const Cp[2], int K, int Nc, int ns, int zfrom, int zk0, int zk1) {
    SkinnyArgs S;
    for (int k = 0; k < SK_MAXP; ++k) {
        const int kk = k < H.nd ? k : 0;
        S.p[k].A = A[kk]; S.p[k].W = W[kk]; S.p[k].bias = nullptr; S.p[k].C = Cp[kk];
    }
    S.lda = lda; S.ldw = ldw; S.ldc = Nc; S.Bm = H.A.B; S.K = K; S.Nc = Nc; S.beta = 0; S.nsplit = ns;
    S.raw = 1;
    if (zfrom < 0) skinny_no_zero_block(S, nt);
    else { S.zfrom = zfrom; S.zk0 = zk0; S.zk1 = zk1; }
    S.part = nullptr; S.tickets = nullptr;
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

// Batches of more than PRJ_MAXB rows are run by the caller in chunks (b0 advanced).
extern "C" int asr_prjhyper_fwd(const asr_prjhyper_desc* d, void* ws, size_t ws_bytes, hipStream_t st) {
    int rc = prj_check(d, ws, ws_bytes);
    if (rc) return rc;
    for (int k = 0; k < d->ndir; ++k) {
        const PrjDir& D = d->dir[k];
        if (!D.inner_in || !D.outer_in || !D.Wf || !D.W3 || !D.s || !D.ai || !D.bi_c || !D.ao || !D.bo_c || !D.r || !D.gi ||
            !D.go || !D.ci || !D.co || !D.xhi || !D.xho || !D.rsi || !D.rso || !D.sc || !D.a)
            return ASR_ERR_ARG;
    }
    if (d->T == 0) return ASR_OK;
    PrjHost H;
    prj_args(d, ws, st, false, H);
    PrjArgs& A = H.A;
    const int T = d->T, N = d->N;
    const bool zb = (N % 16) == 0;          // Wf's zero block [0,N) x [4N,8N) in whole 64-column blocks
    for (int step = 0; step < T; ++step) {
        prj_times(d, A, step);
        const float *rprev[2], *rrow[2], *Wf[2], *W3[2];
        for (int k = 0; k < 2; ++k) {
            const int kk = k < d->ndir ? k : 0;
            const PrjDir& D = A.d[kk];
            rprev[k] = D.r + ((size_t)(A.tp[kk] < 0 ? 0 : A.tp[kk]) * d->Btot + d->b0) * 2 * N;
            rrow[k] = D.r + ((size_t)A.t[kk] * d->Btot + d->b0) * 2 * N;
            Wf[k] = D.Wf; W3[k] = D.W3;
        }
        // recurrent part of the inner gates and a, from the previous r of both cells (zero before the first step:
        // no launch, the cells skip the partial blocks)                                            :329-331,:376
        if (step > 0 && (rc = prj_mm(H, false, rprev, 2 * N, Wf, 8 * N, A.pf, 2 * N, 8 * N, A.ns_f, zb ? 4 * N : -1, N, 2 * N)))
            return rc;
        prj_cell_fwd_launch<0>(A, d->ndir, st);
        ASR_CHECK_LAUNCH();
        if ((rc = prj_mm(H, false, rrow, 2 * N, W3, 12 * N, A.p3, N, 12 * N, A.ns_3, -1, 0, 0))) return rc;   // :364-370
        prj_cell_fwd_launch<1>(A, d->ndir, st);
        ASR_CHECK_LAUNCH();
    }
    return ASR_OK;
}

extern "C" int asr_prjhyper_bwd(const asr_prjhyper_desc* d, void* ws, size_t ws_bytes, hipStream_t st) {
    int rc = prj_check(d, ws, ws_bytes);
    if (rc) return rc;
    for (int k = 0; k < d->ndir; ++k) {
        const PrjDir& D = d->dir[k];
        if (!D.outer_in || !D.Wf || !D.W3 || !D.ai || !D.bi_c || !D.ao || !D.bo_c || !D.gi || !D.go || !D.ci || !D.co ||
            !D.xhi || !D.xho || !D.rsi || !D.rso || !D.sc || !D.a || !D.dr || !D.dta || !D.doi || !D.dsc || !D.douti ||
            !D.douto)
            return ASR_ERR_ARG;
    }
    if (d->T == 0) return ASR_OK;
    PrjHost H;
    prj_args(d, ws, st, true, H);
    PrjArgs& A = H.A;
    const int T = d->T, B = d->B, N = d->N;
    const bool zb = (N % 16) == 0;
    // carried cell gradients start at zero; dr holds the external part (dhid . W_prj^T) of every step on entry
    if (hipMemsetAsync(A.g_c[0], 0, (size_t)2 * d->ndir * B * N * sizeof(float), st) != hipSuccess) return ASR_ERR_LAUNCH;
    for (int step = T - 1; step >= 0; --step) {
        prj_times(d, A, step);
        A.first = step == T - 1;
        const float *dsc[2], *dta[2], *Wf[2], *W3[2];
        for (int k = 0; k < 2; ++k) {
            const int kk = k < d->ndir ? k : 0;
            const PrjDir& D = A.d[kk];
            const size_t row0 = (size_t)A.t[kk] * d->Btot + d->b0;
            dsc[k] = D.dsc + row0 * 12 * N; dta[k] = D.dta + row0 * 8 * N;
            Wf[k] = D.Wf; W3[k] = D.W3;
        }
        prj_cell_bwd_launch<1>(A, d->ndir, st);
        ASR_CHECK_LAUNCH();
        // d r_inner,t += d scale_outer . W3^T
        if ((rc = prj_mm(H, true, dsc, 12 * N, W3, 12 * N, A.pb3, 12 * N, N, A.ns_b3, -1, 0, 0))) return rc;
        prj_cell_bwd_launch<0>(A, d->ndir, st);
        ASR_CHECK_LAUNCH();
        // d r_{t-1} (both cells) += [d inner gates | d a] . Wf^T: handed to the cells of the next step processed
        if (step > 0 && (rc = prj_mm(H, true, dta, 8 * N, Wf, 8 * N, A.pbf, 8 * N, 2 * N, A.ns_bf, zb ? N : -1, 0, 4 * N))) return rc;
    }
    return ASR_OK;
}
