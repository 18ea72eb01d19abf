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
static int prj_persist_grid();
#define PRJ_PERSIST_MAXN 512             // one unit per thread (wider cells spill beside the resident weights)
// tw: workgroups a product's launch is split for
static PrjSplits prj_splits_tw(int B, int N, int ndir, int tw) {
    const int npz = ndir * ((B + 15) / 16);
    PrjSplits s;
    s.f = skinny_nsplit(false, N, 8 * N, npz, tw);        // the shorter of the two reduction ranges (zero block)
    s.w3 = skinny_nsplit(false, N, 12 * N, npz, tw);
    s.bf = skinny_nsplit(true, 4 * N, 2 * N, npz, tw);
    s.b3 = skinny_nsplit(true, 12 * N, N, npz, tw);
    return s;
}
// The persistent launch keeps every wave's slice of the weights on the chip (prj_persist_*_kernel): that needs ONE round
// of weight loads per wave and at most one block per half-workgroup in each of the four products.
static bool prj_resident(int B, int N, int ndir, const PrjSplits& s, int G) {
    const int npz = ndir * ((B + 15) / 16);
    auto cdiv = [](int a, int b) { return (a + b - 1) / b; };
    const int nb_f = cdiv(8 * N, 64), nb_3 = cdiv(12 * N, 64), nb_b3 = cdiv(N, 16), nb_bf = cdiv(2 * N, 16);
    return nb_f * s.f * npz <= 2 * G && nb_3 * s.w3 * npz <= 2 * G && nb_b3 * s.b3 * npz <= 2 * G && nb_bf * s.bf * npz <= 2 * G &&
           cdiv(cdiv(2 * N, 4), 4 * s.f) <= 16 && cdiv(cdiv(N, 4), 4 * s.w3) <= 16 &&
           cdiv(cdiv(12 * N, 64), 4 * s.b3) <= 4 && cdiv(cdiv(8 * N, 64), 4 * s.bf) <= 4;
}
// workgroups of the persistent launch for this shape (one per CU, two product blocks each), 0: step loop
static int prj_use_persist(int B, int N, int ndir) {
    const int g = N <= PRJ_PERSIST_MAXN ? prj_persist_grid() : 0;
    if (!g || g < B * ndir) return 0;                     // every cell needs a workgroup of its own
    return prj_resident(B, N, ndir, prj_splits_tw(B, N, ndir, 2 * g), g) ? g : 0;
}
static PrjSplits prj_splits(int B, int N, int ndir) {
    const int pg = prj_use_persist(B, N, ndir);
    return prj_splits_tw(B, N, ndir, pg ? 2 * pg : prj_target_wgs());
}
// The grid of the persistent launch: as many workgroups as its widest phase has work for (a cell each, two product
// blocks each), not the whole chip -- a small layer leaves the other CUs to whatever else runs on the GPU (two
// full-chip persistent grids of two processes can never be co-resident and would wait for each other until their
// deadlines; grids that fit side by side cannot).
static int prj_persist_wgs(int B, int N, int ndir, int cus) {
    const PrjSplits s = prj_splits_tw(B, N, ndir, 2 * cus);
    const int npz = ndir * ((B + 15) / 16);
    auto cdiv = [](int a, int b) { return (a + b - 1) / b; };
    int m = cdiv(8 * N, 64) * s.f;
    m = m > cdiv(12 * N, 64) * s.w3 ? m : cdiv(12 * N, 64) * s.w3;
    m = m > cdiv(N, 16) * s.b3 ? m : cdiv(N, 16) * s.b3;
    m = m > cdiv(2 * N, 16) * s.bf ? m : cdiv(2 * N, 16) * s.bf;
    int g = cdiv(m * npz, 2);
    if (g < B * ndir) g = B * ndir;
    return g < cus ? g : cus;
}
static size_t prj_part_floats(int B, int N, int ndir) {
    const PrjSplits s = prj_splits(B, N, ndir);
    const size_t bp = (size_t)((B + 15) / 16) * 16;
    const size_t fw = (size_t)s.f * bp * 8 * N + (size_t)s.w3 * bp * 12 * N;
    const size_t bw = (size_t)s.bf * bp * 2 * N + (size_t)s.b3 * bp * N;
    return (fw > bw ? fw : bw) * ndir;
}

// workspace: [sync block of the persistent launch (asr_scan_status(ws, 1) reads it) | carried d cell | partial blocks]
#define PRJ_SYNC_BYTES ((size_t)((SYNC_STRIDE * 4 + 255) / 256) * 256)
#ifdef SCAN_STAMPS
#define PRJ_STAMP_BYTES STAMP_BYTES
#else
#define PRJ_STAMP_BYTES 0
#endif
extern "C" size_t asr_prjhyper_ws_bytes(int B, int N, int F, int ndir) {
    (void)F;
    return PRJ_SYNC_BYTES + ((size_t)2 * ndir * B * N + prj_part_floats(B, N, ndir)) * sizeof(float) + 256 + PRJ_STAMP_BYTES;
}

static int prj_check(const asr_prjhyper_desc* d, void* ws, size_t ws_bytes) {
    if (!d || !ws || d->T < 0 || d->B <= 0 || d->B > PRJ_MAXB || d->N <= 0 || d->F <= 0 || (d->N & 3) || (d->F & 3) ||
        d->N > 256 * PRJ_MAXU || d->ndir < 1 || d->ndir > 2 || !d->mask || d->b0 < 0 || d->b0 + d->B > d->Btot ||
        ((uintptr_t)ws & 15))
        return ASR_ERR_ARG;
    if (ws_bytes < asr_prjhyper_ws_bytes(d->B, d->N, d->F, d->ndir)) return ASR_ERR_WS;
    return ASR_OK;
}

// in-kernel phase stamps of the persistent launch (diagnostic build only): the workspace tail
static unsigned long long* prj_dbg(const asr_prjhyper_desc* d, void* ws) {
    return PRJ_STAMP_BYTES ? (unsigned long long*)((char*)ws + asr_prjhyper_ws_bytes(d->B, d->N, d->F, d->ndir) - PRJ_STAMP_BYTES)
                           : nullptr;
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
    float* p = (float*)((char*)ws + PRJ_SYNC_BYTES);
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

// ------------------------------------------------------------------------------------------------
// The same steps as ONE persistent launch per pass: the four kernels of a step become four phases
// of a loop inside one kernel, a kernel boundary (~5 us in this chain) becomes an exchange of scan_common.h (sc1 stores,
// every storing wave drains, one arrival per workgroup on a sharded counter; consumers poll, then read with sc1 loads:
// ~2.3 us).  One workgroup per CU (the sc1 hand-off is measured for that), as many as the chip has CUs; a phase's
// blocks -- the (output block, split, direction x row chunk) triples of a skinny product, the (row, direction) pairs of
// a cell -- are dealt round-robin to the workgroups.  The folded weights (50 MB at 512 units, both directions) are
// streamed from L2 / Infinity Cache every step as before.  Results are those of the step loop bit for bit when the
// splits are the same (same bodies, same summation order); the splits follow the grid, so they are the step loop's
// only when ASR_SK_WGS equals the CU count.

__device__ __forceinline__ void prj_sk_args(SkinnyArgs& S, const PrjArgs& A, const float* const Ain[2], int lda,
                                            bool w3, int ldw, float* const Cp[2], int K, int Nc, int ns, int zfrom,
                                            int zk0, int zk1, bool nt) {
#pragma unroll
    for (int k = 0; k < SK_MAXP; ++k) {
        S.p[k].A = Ain[k]; S.p[k].W = w3 ? A.d[k].W3 : A.d[k].Wf; S.p[k].bias = nullptr; S.p[k].C = Cp[k];
    }
    S.lda = lda; S.ldw = ldw; S.ldc = Nc; S.Bm = A.B; S.K = K; S.Nc = Nc; S.beta = 0; S.nsplit = ns; S.raw = 1;
    S.nchunk = (A.B + 15) / 16;
    if (zfrom < 0) { S.zfrom = nt ? 0 : 0x7fffffff; S.zk0 = 0; S.zk1 = 0; }
    else { S.zfrom = zfrom; S.zk0 = zk0; S.zk1 = zk1; }
    S.part = nullptr; S.tickets = nullptr;
}


// Workgroups of 512 threads: a product phase runs two blocks side by side (threads 0-255 and 256-511: eight waves of
// loads in flight per CU), a cell uses all 512 threads for its row.  Every phase has the shape
//     loads that do not depend on the exchange before it  ->  wait for that exchange  ->  the dependent loads, the
//     arithmetic, the stores that other workgroups read  ->  publish  ->  the stores only later kernels read
// so the weights of a product stream in under the cell phase before it, a cell's inputs under the product before it, and
// the saved tensors' stores (21 floats per unit and cell) drain under the phase after.
#define PRJ_PT 512

// Sum of ns partial blocks in split order, SL loads in flight, every load unconditional (PART_SUM_BEGIN's rule); the
// number of slots follows ns, so that two splits do not pay four loads.
template <int SL, class F>
__device__ __forceinline__ void prj_part_sum_sl(int ns, F f) {
    for (int sp0 = 0; sp0 < ns; sp0 += SL) {
#pragma unroll
        for (int i4 = 0; i4 < SL; ++i4) f(min(sp0 + i4, ns - 1), sp0 + i4 < ns ? 1.f : 0.f);
    }
}
template <class F>
__device__ __forceinline__ void prj_part_sum(int ns, F f) {
    if (ns == 1) prj_part_sum_sl<1>(ns, f);
    else if (ns == 2) prj_part_sum_sl<2>(ns, f);
    else prj_part_sum_sl<4>(ns, f);
}
// two workgroup-wide sums with one pair of barriers (sm: 2 NW floats)
template <int NW>
__device__ __forceinline__ void block_sum2(float& a, float& b, float* sm) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6] = a; sm[NW + (threadIdx.x >> 6)] = b; }
    __syncthreads();
    a = sm[0] + sm[1] + sm[2] + sm[3]; b = sm[NW] + sm[NW + 1] + sm[NW + 2] + sm[NW + 3];
    if (NW == 8) { a += sm[4] + sm[5] + sm[6] + sm[7]; b += sm[NW + 4] + sm[NW + 5] + sm[NW + 6] + sm[NW + 7]; }
}

// The cells of prj_cell_fwd_kernel / prj_cell_bwd_kernel in that shape (same expressions, same order).
template <int OUTER, int NU, class WaitF, class PubF>
__device__ __forceinline__ bool prj_cell_fwd_pbody(const PrjArgs& A, int b, int dir, int t, int tp, float* sm, WaitF waitf,
                                                   PubF pubf) {
    const PrjDir& D = A.d[dir];
    const int N = A.N;
    const size_t row = (size_t)t * A.Btot + A.b0 + b, rowp = (size_t)(tp < 0 ? 0 : tp) * A.Btot + A.b0 + b;
    const float m = A.mask[(size_t)(A.b0 + b) * A.msb + (size_t)t * A.mst];
    const float* alpha = OUTER ? D.ao : D.ai;
    const float* beta = OUTER ? D.bo_c : D.bi_c;
    float* gsv = (OUTER ? D.go : D.gi) + row * 4 * N;
    float* cst = OUTER ? D.co : D.ci;
    float* rst = D.r + (OUTER ? N : 0);              // row stride 2N
    int nn[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) nn[u] = min((int)threadIdx.x + u * PRJ_PT, N - 1);
    float rec[NU][4], scv[NU][12], cp[NU], rpv[NU], xin[NU][4], al[NU], be[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {                   // ---- independent of the exchange
#pragma unroll
        for (int k = 0; k < 4; ++k) rec[u][k] = 0.f;
#pragma unroll
        for (int k = 0; k < 12; ++k) scv[u][k] = OUTER ? D.s[k * N + nn[u]] : 0.f;
        cp[u] = ldx<true>(cst + rowp * N + nn[u]);                          // this workgroup's own stores of the step before
        rpv[u] = ldx<true>(rst + rowp * 2 * N + nn[u]);
        al[u] = alpha[nn[u]]; be[u] = beta[nn[u]];
#pragma unroll
        for (int k = 0; k < 4; ++k) xin[u][k] = (OUTER ? D.outer_in : D.inner_in)[row * 4 * N + k * N + nn[u]];
    }
    if (!waitf()) return false;
    if (tp >= 0) {                                   // ---- sums of the partial blocks
        const float* pf = A.pf[dir] + (size_t)b * 8 * N + (OUTER ? 4 * N : 0);
        prj_part_sum(A.ns_f, [&](int sp, float pw) {
            const float* pp = pf + (size_t)sp * A.Bpad * 8 * N;
#pragma unroll
            for (int u = 0; u < NU; ++u) {
#pragma unroll
                for (int k = 0; k < 4; ++k) rec[u][k] += pw * ldx<true>(pp + k * N + nn[u]);
            }
        });
    }
    if (OUTER) {
        const float* p3 = A.p3[dir] + (size_t)b * 12 * N;
        prj_part_sum(A.ns_3, [&](int sp, float pw) {
            const float* pp = p3 + (size_t)sp * A.Bpad * 12 * N;
#pragma unroll
            for (int u = 0; u < NU; ++u) {
#pragma unroll
                for (int k = 0; k < 12; ++k) scv[u][k] += pw * ldx<true>(pp + k * N + nn[u]);
            }
        });
    }
    float cn[NU], g4[NU][4];
    float s1 = 0.f;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int n = threadIdx.x + u * PRJ_PT;
        float g[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            g[k] = OUTER ? xin[u][k] * scv[u][k] + rec[u][k] * scv[u][4 + k] + scv[u][8 + k]      // :375-377
                         : xin[u][k] + rec[u][k];                                                   // :329-331
        g4[u][0] = sigmoidf_(g[0]); g4[u][1] = sigmoidf_(g[1]); g4[u][2] = tanhf_(g[2]); g4[u][3] = sigmoidf_(g[3]);
        if (tp < 0) cp[u] = 0.f;
        cn[u] = g4[u][1] * cp[u] + g4[u][0] * g4[u][2];                // :355 | :401
        if (n < N) s1 += cn[u];
    }
    const float mean = block_sum<PRJ_PT / 64>(s1, sm) / N;
    float s2 = 0.f;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int n = threadIdx.x + u * PRJ_PT;
        if (n < N) { const float dlt = cn[u] - mean; s2 += dlt * dlt; }
    }
    const float rstd = 1.0f / sqrtf(block_sum<PRJ_PT / 64>(s2, sm) / N + ASR_EPS);   // ln :9-12
    float xh[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int n = threadIdx.x + u * PRJ_PT;
        xh[u] = (cn[u] - mean) * rstd;
        const float r = g4[u][3] * tanhf_(al[u] * xh[u] + be[u]);      // :357-363 | :403-409
        const float rp = tp < 0 ? 0.f : rpv[u];
        if (n < N) {
            stx<true>(cst + row * N + n, m != 0.f ? cn[u] : cp[u]);      // :441-444
            stx<true>(rst + row * 2 * N + n, m != 0.f ? r : rp);
        }
    }
    pubf();
    if (threadIdx.x == 0) (OUTER ? D.rso : D.rsi)[row] = rstd;        // ---- read by later kernels only
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int n = threadIdx.x + u * PRJ_PT;
        if (n < N) {
            gsv[n] = g4[u][0]; gsv[N + n] = g4[u][1]; gsv[2 * N + n] = g4[u][2]; gsv[3 * N + n] = g4[u][3];
            if (OUTER) {
#pragma unroll
                for (int k = 0; k < 4; ++k) D.a[row * 4 * N + k * N + n] = rec[u][k];
#pragma unroll
                for (int k = 0; k < 12; ++k) D.sc[row * 12 * N + k * N + n] = scv[u][k];
            }
            (OUTER ? D.xho : D.xhi)[row * N + n] = xh[u];
        }
    }
    return true;
}

template <int OUTER, int NU, class WaitF, class PubF>
__device__ __forceinline__ bool prj_cell_bwd_pbody(const PrjArgs& A, int b, int dir, int t, int tp, bool first, float* sm,
                                                   WaitF waitf, PubF pubf) {
    const PrjDir& D = A.d[dir];
    const int N = A.N;
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
    float gr[NU], gcar[NU], xh[NU], al[NU], be[NU], g4[NU][4], cpr[NU], drp[NU];
    float s_in[NU][4], s_ou[NU][4], x_in[NU][4], x_a[NU][4];
    int nn[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {                   // ---- independent of the exchange
        const int n = nn[u] = min((int)threadIdx.x + u * PRJ_PT, N - 1);
        gr[u] = ldx<true>(dr_t + n); gcar[u] = ldx<true>(g_c + n); xh[u] = xhat[n]; al[u] = alpha[n]; be[u] = beta[n];
        cpr[u] = cpv[n];
        drp[u] = ldx<true>(dr_p + n);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            g4[u][k] = gsv[k * N + n];
            if (OUTER) {
                s_in[u][k] = sc[k * N + n]; s_ou[u][k] = sc[4 * N + k * N + n];
                x_in[u][k] = oin[k * N + n]; x_a[u][k] = av[k * N + n];
            }
        }
    }
    if (!waitf()) return false;
    // total d r_t of this cell: + what the recurrent products hand over (their partial blocks)
    if (!first) {
        const float* pb = A.pbf[dir] + (size_t)b * 2 * N + (OUTER ? N : 0);
        prj_part_sum(A.ns_bf, [&](int sp, float pw) {
#pragma unroll
            for (int u = 0; u < NU; ++u) gr[u] += pw * ldx<true>(pb + (size_t)sp * A.Bpad * 2 * N + nn[u]);
        });
    }
    if (!OUTER) {
        const float* pb = A.pb3[dir] + (size_t)b * N;
        prj_part_sum(A.ns_b3, [&](int sp, float pw) {
#pragma unroll
            for (int u = 0; u < NU; ++u) gr[u] += pw * ldx<true>(pb + (size_t)sp * A.Bpad * N + nn[u]);
        });
    }
    float dxh[NU], do_[NU], dout[NU];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int n = threadIdx.x + u * PRJ_PT;
        const bool act = on && n < N;
        const float tc = tanhf_(al[u] * xh[u] + be[u]);
        do_[u] = act ? gr[u] * tc : 0.f;
        dout[u] = act ? clipf_(gr[u] * g4[u][3] * (1.f - tc * tc), A.clip) : 0.f;    // :358-361 | :404-407
        dxh[u] = dout[u] * al[u];
        s1 += dxh[u]; s2 += dxh[u] * xh[u];
    }
    block_sum2<PRJ_PT / 64>(s1, s2, sm);
    const float m1 = s1 / N, m2 = s2 / N;
    float dg[NU][4], gcn[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int n = threadIdx.x + u * PRJ_PT;
#pragma unroll
        for (int k = 0; k < 4; ++k) dg[u][k] = 0.f;
        const float ig = g4[u][0], fg = g4[u][1], gc = g4[u][2], o = g4[u][3];
        const float cprev = tp < 0 ? 0.f : cpr[u];
        const float dct = gcar[u] + rstd * (dxh[u] - m1 - xh[u] * m2);
        gcn[u] = dct * fg;
        if (on) {
            dg[u][0] = clipf_(dct * gc * ig * (1.f - ig), A.clip);
            dg[u][1] = clipf_(dct * cprev * fg * (1.f - fg), A.clip);
            dg[u][2] = clipf_(dct * ig * (1.f - gc * gc), A.clip);
            dg[u][3] = clipf_(do_[u] * o * (1.f - o), A.clip);
        }
        if (n >= N) continue;                            // stores only below: what the products of this step read
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (OUTER) {
                stx<true>(D.dta + row * 8 * N + 4 * N + k * N + n, dg[u][k] * s_ou[u][k]);     // d (outer_hid_prev.W)
                float* ds = D.dsc + row * 12 * N;
                stx<true>(ds + k * N + n, dg[u][k] * x_in[u][k]);
                stx<true>(ds + 4 * N + k * N + n, dg[u][k] * x_a[u][k]);
                stx<true>(ds + 8 * N + k * N + n, dg[u][k]);
            } else {
                stx<true>(D.dta + row * 8 * N + k * N + n, dg[u][k]);
            }
        }
    }
    pubf();
#pragma unroll
    for (int u = 0; u < NU; ++u) {                   // ---- this workgroup's own carry, and what later kernels read
        const int n = threadIdx.x + u * PRJ_PT;
        if (n >= N) continue;
        dout_sv[n] = dout[u];
        if (on) stx<true>(g_c + n, gcn[u]);
        else if (tp >= 0) stx<true>(dr_p + n, drp[u] + gr[u]);   // off: the state was copied, its gradient passes on
        if (OUTER) {
#pragma unroll
            for (int k = 0; k < 4; ++k) D.doi[row * 4 * N + k * N + n] = dg[u][k] * s_in[u][k];     // d outer_input
        }
    }
    return true;
}

#define PRJ_PERSIST_SETUP \
    GroupSync gs; \
    STAMP_DECL \
    const int wg = blockIdx.x, G = gridDim.x; \
    group_sync_init(gs, sync, G, wg, flag); \
    bool pending = false;                    /* a publish whose wait is still to come (uniform over the grid) */ \
    int ph = 0;                              /* phase of the step (diagnostic stamps: 4 per phase) */ \
    auto waitf = [&]() -> bool { STAMP(4 * ph); if (!pending) return true; pending = false; const bool ok_ = group_wait(gs); \
                                 STAMP(4 * ph + 1); return ok_; }; \
    auto pubf = [&]() { STAMP(4 * ph + 2); group_publish(gs); pending = true; STAMP(4 * ph + 3); };

// LDS of the persistent kernels: [8 waves][16][64] f32x4 = 128 KB (each wave's slice of W3) | partial sums of the
// waves of a product block: 2 halves x 3 waves x 4 KB (forward), x 1 KB (backward) | 8 floats | flag
#define PRJ_LDS_W (8 * 16 * 64 * 4)                  /* floats */
#define PRJ_PERSIST_LDS ((PRJ_LDS_W + 2 * 3 * 4 * 64 * 4) * 4 + 256)

// Every wave needs ONE round of weight loads in each product and every half-workgroup has at most one block per product
// (prj_resident, checked on the host; otherwise the layer runs the step loop).  A wave's slices are the same bytes every
// step, so they are loaded once: Wf's 16 f32x4 per lane stay in registers, W3's in LDS -- the launch streams no weights
// at all (the step loop: 50 MB per pass at 512 units, from the Infinity Cache).
template <int NU>
__global__ __launch_bounds__(PRJ_PT) void prj_persist_fwd_kernel(PrjArgs A, unsigned* sync, int zb, unsigned long long* dbg) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // wave-uniform values the compiler must know to be uniform: the product bodies build buffer descriptors from the
    // (direction-selected) operand pointers, and a descriptor from a value it takes to be divergent costs a waterfall
    // loop around every load and store
    const int half = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8), lane = threadIdx.x & 63;
    f32x4 (*wl)[64] = (f32x4 (*)[64])smem + (threadIdx.x >> 6) * 16;    // this wave's [16][64]
    f32x4 (*red)[4][64] = (f32x4 (*)[4][64])(smem + PRJ_LDS_W) + half * 3;
    float* sm = smem + PRJ_LDS_W + 2 * 3 * 4 * 64 * 4;
    int* flag = (int*)(sm + 16);
    PRJ_PERSIST_SETUP
    const int N = A.N, npz = A.ndir * ((A.B + 15) / 16);
    const int nb_f = (8 * N + 63) / 64, nb_3 = (12 * N + 63) / 64;
    const int tot_f = nb_f * A.ns_f * npz, tot_3 = nb_3 * A.ns_3 * npz;
    const int cb = wg % A.B, cd = wg / A.B;          // this workgroup's cell (workgroups >= B ndir have none)
    const bool cell = wg < A.B * A.ndir;
    SkinnyArgs Sf, S3;
    {
        const float* z[2] = {nullptr, nullptr};
        prj_sk_args(Sf, A, z, 2 * N, false, 8 * N, A.pf, 2 * N, 8 * N, A.ns_f, zb ? 4 * N : -1, N, 2 * N, false);
        prj_sk_args(S3, A, z, 2 * N, true, 12 * N, A.p3, N, 12 * N, A.ns_3, -1, 0, 0, false);
    }
    // the one block of this half-workgroup in each product
    const int vb = 2 * wg + half;
    const bool lf = vb < tot_f, l3 = vb < tot_3;
    const int bxf = vb % nb_f, byf = (vb / nb_f) % A.ns_f, bzf = vb / (nb_f * A.ns_f);
    const int bx3 = vb % nb_3, by3 = (vb / nb_3) % A.ns_3, bz3 = vb / (nb_3 * A.ns_3);
    SkinnyNNCtx cf = skinny_nn_ctx(Sf, bxf, byf, bzf, lf), c3 = skinny_nn_ctx(S3, bx3, by3, bz3, l3);
    f32x4 wf[16];
    {
        f32x4 w3[16];
        skinny_nn_wload<16>(S3, c3, bz3, c3.g0, w3);
#pragma unroll
        for (int i = 0; i < 16; ++i) wl[i][lane] = w3[i];
    }
    skinny_nn_wload<16>(Sf, cf, bzf, cf.g0, wf);
    bool ok = true;
    for (int step = 0; step < A.T && ok; ++step) {
        int t[2], tp[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            t[k] = A.d[k].reverse ? A.T - 1 - step : step;
            tp[k] = step == 0 ? -1 : (A.d[k].reverse ? t[k] + 1 : t[k] - 1);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            Sf.p[k].A = A.d[k].r + ((size_t)(tp[k] < 0 ? 0 : tp[k]) * A.Btot + A.b0) * 2 * N;
            S3.p[k].A = A.d[k].r + ((size_t)t[k] * A.Btot + A.b0) * 2 * N;
        }
        ph = 0;
        if (step > 0) {                                                 // [tg | a] = r_{t-1} . Wf
            ok = skinny_nn_pbody<16>(Sf, cf, byf, bzf, [&](int i) { return wf[i]; }, red, lf, waitf);
            __syncthreads();
            if (!ok) break;
            pubf();
        }
        ph = 1;
        if (cell) ok = prj_cell_fwd_pbody<0, NU>(A, cb, cd, cd ? t[1] : t[0], cd ? tp[1] : tp[0], sm, waitf, pubf);
        else { ok = waitf(); pubf(); }
        if (!ok) break;
        ph = 2;
        {                                                               // scale_outer = r_inner,t . W3 + s
            ok = skinny_nn_pbody<16>(S3, c3, by3, bz3, [&](int i) { return wl[i][lane]; }, red, l3, waitf);
            __syncthreads();
            if (!ok) break;
            pubf();
        }
        ph = 3;
        if (cell) ok = prj_cell_fwd_pbody<1, NU>(A, cb, cd, cd ? t[1] : t[0], cd ? tp[1] : tp[0], sm, waitf, pubf);
        else { ok = waitf(); pubf(); }
    }
    STAMP_DUMP(dbg, blockIdx.x);
}

template <int NU>
__global__ __launch_bounds__(PRJ_PT) void prj_persist_bwd_kernel(PrjArgs A, unsigned* sync, int zb, unsigned long long* dbg) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // wave-uniform values the compiler must know to be uniform: the product bodies build buffer descriptors from the
    // (direction-selected) operand pointers, and a descriptor from a value it takes to be divergent costs a waterfall
    // loop around every load and store
    const int half = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8), lane = threadIdx.x & 63;
    f32x4 (*wl)[64] = (f32x4 (*)[64])smem + (threadIdx.x >> 6) * 16;    // this wave's [16][64]
    f32x4 (*red)[64] = (f32x4 (*)[64])(smem + PRJ_LDS_W) + half * 3;
    float* sm = smem + PRJ_LDS_W + 2 * 3 * 4 * 64 * 4;
    int* flag = (int*)(sm + 16);
    PRJ_PERSIST_SETUP
    const int N = A.N, npz = A.ndir * ((A.B + 15) / 16);
    const int nb_b3 = (N + 15) / 16, nb_bf = (2 * N + 15) / 16;
    const int tot_b3 = nb_b3 * A.ns_b3 * npz, tot_bf = nb_bf * A.ns_bf * npz;
    const int cb = wg % A.B, cd = wg / A.B;
    const bool cell = wg < A.B * A.ndir;
    SkinnyArgs S3, Sf;
    {
        const float* z[2] = {nullptr, nullptr};
        prj_sk_args(S3, A, z, 12 * N, true, 12 * N, A.pb3, 12 * N, N, A.ns_b3, -1, 0, 0, true);
        prj_sk_args(Sf, A, z, 8 * N, false, 8 * N, A.pbf, 8 * N, 2 * N, A.ns_bf, zb ? N : -1, 0, 4 * N, true);
    }
    const int vb = 2 * wg + half;
    const bool l3 = vb < tot_b3, lf = vb < tot_bf;
    const int bx3 = vb % nb_b3, by3 = (vb / nb_b3) % A.ns_b3, bz3 = vb / (nb_b3 * A.ns_b3);
    const int bxf = vb % nb_bf, byf = (vb / nb_bf) % A.ns_bf, bzf = vb / (nb_bf * A.ns_bf);
    SkinnyNTCtx c3 = skinny_nt_ctx(S3, bx3, by3, bz3, l3), cf = skinny_nt_ctx(Sf, bxf, byf, bzf, lf);
    f32x4 wf[16];
    {
        f32x4 w3[16];
        skinny_nt_wload<4>(S3, c3, bz3, c3.c0, w3);
#pragma unroll
        for (int i = 0; i < 16; ++i) wl[i][lane] = w3[i];
    }
    skinny_nt_wload<4>(Sf, cf, bzf, cf.c0, wf);
    bool ok = true;
    for (int step = A.T - 1; step >= 0 && ok; --step) {
        int t[2], tp[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            t[k] = A.d[k].reverse ? A.T - 1 - step : step;
            tp[k] = step == 0 ? -1 : (A.d[k].reverse ? t[k] + 1 : t[k] - 1);
        }
        const bool first = step == A.T - 1;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const size_t row0 = (size_t)t[k] * A.Btot + A.b0;
            S3.p[k].A = A.d[k].dsc + row0 * 12 * N; Sf.p[k].A = A.d[k].dta + row0 * 8 * N;
        }
        ph = 0;
        if (cell) ok = prj_cell_bwd_pbody<1, NU>(A, cb, cd, cd ? t[1] : t[0], cd ? tp[1] : tp[0], first, sm, waitf, pubf);
        else { ok = waitf(); pubf(); }
        if (!ok) break;
        ph = 1;
        {                                                               // d r_inner,t += d scale_outer . W3^T
            ok = skinny_nt_pbody<4>(S3, c3, by3, bz3, [&](int i) { return wl[i][lane]; }, red, l3, waitf);
            __syncthreads();
            if (!ok) break;
            pubf();
        }
        ph = 2;
        if (cell) ok = prj_cell_bwd_pbody<0, NU>(A, cb, cd, cd ? t[1] : t[0], cd ? tp[1] : tp[0], first, sm, waitf, pubf);
        else { ok = waitf(); pubf(); }
        if (!ok) break;
        ph = 3;
        if (step > 0) {                                                 // d r_{t-1} += [d inner gates | d a] . Wf^T
            ok = skinny_nt_pbody<4>(Sf, cf, byf, bzf, [&](int i) { return wf[i]; }, red, lf, waitf);
            __syncthreads();
            if (!ok) break;
            pubf();
        }
    }
    STAMP_DUMP(dbg, blockIdx.x);
}

static int prj_mode_ = 0;
extern "C" int asr_prjhyper_set_mode(int mode) { const int old = prj_mode_; prj_mode_ = mode ? 1 : 0; return old; }

// 0: step loop; otherwise the number of workgroups of the persistent launch (one per CU)
static int prj_persist_grid() {
    static int on = -1;
    if (on < 0) { const char* e = getenv("ASR_PRJ_PERSIST"); on = e ? atoi(e) != 0 : 1; }
    if (!on || prj_mode_) return 0;
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 16) return 0;
    return n;
}

static int prj_persist_launch(const PrjArgs& A, unsigned* sync, int zb, bool backward, int G, hipStream_t st,
                              unsigned long long* dbg) {
    const void* fn;
    const int N = A.N;
    if (N > PRJ_PT) return ASR_ERR_LIMIT;
    fn = backward ? (const void*)prj_persist_bwd_kernel<1> : (const void*)prj_persist_fwd_kernel<1>;
    if (!scan_grid_fits((size_t)G, PRJ_PERSIST_LDS)) return ASR_ERR_LIMIT;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, PRJ_PERSIST_LDS) != hipSuccess) return ASR_ERR_LAUNCH;
    void* kargs[] = {(void*)&A, (void*)&sync, (void*)&zb, (void*)&dbg};
    if (hipLaunchKernel(fn, dim3(G), dim3(PRJ_PT), kargs, PRJ_PERSIST_LDS, st) != hipSuccess) return ASR_ERR_LAUNCH;
    ASR_CHECK_LAUNCH();
    return ASR_OK;
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
    if (hipMemsetAsync(ws, 0, PRJ_SYNC_BYTES, st) != hipSuccess) return ASR_ERR_LAUNCH;
    if (const int cus = prj_use_persist(d->B, N, d->ndir))
        return prj_persist_launch(A, (unsigned*)ws, zb ? 1 : 0, false, prj_persist_wgs(d->B, N, d->ndir, cus), st, prj_dbg(d, ws));
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
    if (hipMemsetAsync(ws, 0, PRJ_SYNC_BYTES, st) != hipSuccess) return ASR_ERR_LAUNCH;
    if (const int cus = prj_use_persist(d->B, N, d->ndir))
        return prj_persist_launch(A, (unsigned*)ws, zb ? 1 : 0, true, prj_persist_wgs(d->B, N, d->ndir, cus), st, prj_dbg(d, ws));
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
