// Shared by the step-loop layers (step_prjhyper.hip, step_scaling.hip): the skinny products of one time step and a
// workgroup-wide sum.  Kernels have internal linkage: every translation unit that includes this gets its own copy.
#pragma once
#include "scan_common.h"

// ------------------------------------------------------------------------------------------------
// Skinny products of the step: Bm batch rows in chunks of 16 (one MFMA tile of rows), both directions and every row chunk in
// one launch (blockIdx.z = direction * nchunk + chunk).  They are
// L2-latency-bound, so the reduction axis is cut into 4 * nsplit slices: the 4 waves of a workgroup
// reduce through LDS, the nsplit workgroups of an output block hand their partial blocks over in the
// launch (cdna_hip_programming.md Guideline 16: 16-byte sc1 stores, vmcnt(0), one ticket atomic; the
// last arriver sums the partials with sc1 loads IN SPLIT ORDER, so the result does not depend on
// arrival order) -- no second launch, deterministic.
#define SK_MAXP 2
#define SK_TICKETS 4096
#define SK_PART_FLOATS (512 * 1024)
#ifndef SK_TARGET_WGS
#define SK_TARGET_WGS 512      // workgroups per launch the reduction axis is split for
#endif
struct SkinnyProb { const float* A; const float* W; const float* bias; float* C; };
struct SkinnyArgs {
    SkinnyProb p[SK_MAXP];
    int lda, ldw, ldc, Bm, K, Nc, beta, nsplit;     // K: reduction length, Nc: output columns
    int nchunk;                                     // row chunks of 16: ceil(Bm / 16)
    // raw != 0: no combine in this launch.  Split sp writes its partial block of rows to C + (sp * 16 nchunk + row) * Nc
    // (rows >= Bm hold zeros) and the NEXT kernel on the stream sums the nsplit blocks in split order while it loads its
    // inputs (no bias, no beta).  The in-launch combine (tickets, an agent acquire, nsplit dependent round trips
    // of the last arriver) was about half of a 13 us product at the ProjectionHyper shapes.
    int raw;
    // a known zero block of W is skipped: output blocks whose first column (nn) / row of W (nt) is >= zfrom (nn) or
    // < zfrom (nt) reduce over [zk0, zk1) only; everything else over [0, K)
    int zfrom, zk0, zk1;
    float* part; unsigned* tickets;
};

// true for the workgroup that must write the output block; acc then holds the full sum
template <int NV>
__device__ __forceinline__ bool skinny_combine(const SkinnyArgs& S, f32x4 (&acc)[NV], int blk, int nblk, int lane) {
    if (S.nsplit == 1) return true;
    const int split = blockIdx.y, prob = blockIdx.z;       // prob: (direction, row chunk)
    const unsigned tile_bytes = 64u * NV * 16u;
    const size_t slot = (size_t)(prob * nblk + blk) * S.nsplit;
    __amdgpu_buffer_rsrc_t rs = make_rsrc_(S.part + slot * (64 * NV * 4), (unsigned)S.nsplit * tile_bytes);
#pragma unroll
    for (int j = 0; j < NV; ++j) st_sc1_f4(rs, (unsigned)split * tile_bytes + (unsigned)(lane * NV + j) * 16u, acc[j]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned* tk = S.tickets + prob * nblk + blk;
    unsigned old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    old = __shfl(old, 0);
    if (old != (unsigned)S.nsplit - 1u) return false;
    if (lane == 0) __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // next launch
    // several of these workgroups share a CU, which is outside what the sc1-load hand-off is measured for
    // (MI355X_MICROARCH.md, table under 'Valid forms': one workgroup per CU): keep the agent acquire as well
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#pragma unroll
    for (int j = 0; j < NV; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int sp = 0; sp < S.nsplit; ++sp) {
#pragma unroll
        for (int j = 0; j < NV; ++j) acc[j] += ld_sc1_f4(rs, (unsigned)sp * tile_bytes + (unsigned)(lane * NV + j) * 16u);
    }
    return true;
}

// C[Bm x Nc] = A[Bm x K] . W[K x Nc] (+ bias) (+ C if beta); Nc % 4 == 0, rows 16-byte aligned.
// Workgroup = one 16 x 64 output block: lane (q, bl) owns rows 4q..4q+3 of columns n0+4bl..n0+4bl+3
// (tile j of the four 16x16 MFMA tiles = columns n0+4bl+j: one 16-byte W load feeds four MFMAs).
static __global__ __launch_bounds__(256) void skinny_nn_kernel(SkinnyArgs S) {
    __shared__ f32x4 red[4][4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, bl = lane & 15;
    const int blk = blockIdx.x;
    const SkinnyProb& P = S.p[blockIdx.z / S.nchunk];
    const int row0 = (blockIdx.z % S.nchunk) * 16;
    const int col = blk * 64 + 4 * bl;
    const bool cv = col < S.Nc, rv = row0 + bl < S.Bm;
    const bool zb = blk * 64 >= S.zfrom;
    const int klo = zb ? S.zk0 : 0, khi = zb ? S.zk1 : S.K;
    const int ngrp = (khi - klo + 3) >> 2, nsl = 4 * S.nsplit;
    const int per = (ngrp + nsl - 1) / nsl;
    const int g0 = ((int)blockIdx.y * 4 + wave) * per;
    const int g1 = min(g0 + per, ngrp);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc[4] = {zero, zero, zero, zero};
    const float* Ap = P.A + (size_t)(row0 + bl) * S.lda;
    const float* Wp = P.W + col;
    for (int g = g0; g < g1; g += 8) {
        float a[8]; f32x4 w[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int k = klo + 4 * (g + s) + q;
            const bool kv = (g + s) < g1 && k < khi;
            a[s] = (kv && rv) ? Ap[k] : 0.f;
            w[s] = (kv && cv) ? *(const f32x4*)(Wp + (size_t)k * S.ldw) : zero;
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], w[s][j], acc[j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) red[wave][j][lane] = acc[j];
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = red[0][j][lane] + red[1][j][lane] + red[2][j][lane] + red[3][j][lane];
    if (S.raw) {
        if (cv) {
            float* pc = P.C + ((size_t)blockIdx.y * 16 * S.nchunk + row0 + q * 4) * S.Nc + col;
#pragma unroll
            for (int r = 0; r < 4; ++r) *(f32x4*)(pc + (size_t)r * S.Nc) = f32x4{acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
        }
        return;
    }
    if (!skinny_combine<4>(S, acc, blk, gridDim.x, lane)) return;
    if (cv) {
        const f32x4 bv = P.bias ? *(const f32x4*)(P.bias + col) : zero;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = row0 + q * 4 + r;
            if (row < S.Bm) {
                f32x4* pc = (f32x4*)(P.C + (size_t)row * S.ldc + col);
                f32x4 v = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
                v += bv;
                if (S.beta) v += *pc;
                *pc = v;
            }
        }
    }
}

// C[Bm x Nc] = A[Bm x K] . W[Nc x K]^T (+ C if beta); K % 4 == 0, rows 16-byte aligned.  Workgroup = one
// 16-column output tile; the K axis goes in 64-column chunks (lane (q, bl) reads 16 bytes of row bl per load, the four
// lanes of a row 64 contiguous bytes: MFMA k index q <-> column 16e + 4q + i of the chunk).
static __global__ __launch_bounds__(256) void skinny_nt_kernel(SkinnyArgs S) {
    __shared__ f32x4 red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, bl = lane & 15;
    const int blk = blockIdx.x;
    const SkinnyProb& P = S.p[blockIdx.z / S.nchunk];
    const int row0 = (blockIdx.z % S.nchunk) * 16;
    const int kout = blk * 16 + bl;
    const bool kv = kout < S.Nc, rv = row0 + bl < S.Bm;
    const bool zb = blk * 16 < S.zfrom;
    const int klo = zb ? S.zk0 : 0, khi = zb ? S.zk1 : S.K;
    const int nch = (khi - klo + 63) >> 6, nsl = 4 * S.nsplit;
    const int per = (nch + nsl - 1) / nsl;
    const int c0 = ((int)blockIdx.y * 4 + wave) * per;
    const int c1 = min(c0 + per, nch);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc0 = zero, acc1 = zero;
    const float* Ap = P.A + (size_t)(row0 + bl) * S.lda + klo + 4 * q;
    const float* Wp = P.W + (size_t)kout * S.ldw + klo + 4 * q;
    for (int ch = c0; ch < c1; ch += 2) {
        f32x4 av[8], wv[8];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int off = (ch + u) * 64 + 16 * e;
                const bool ok = (ch + u) < c1 && klo + off + 4 * q < khi;
                av[u * 4 + e] = (ok && rv) ? *(const f32x4*)(Ap + off) : zero;
                wv[u * 4 + e] = (ok && kv) ? *(const f32x4*)(Wp + off) : zero;
            }
        }
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[v][0], wv[v][0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[v][1], wv[v][1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[v][2], wv[v][2], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[v][3], wv[v][3], acc1, 0, 0, 0);
        }
    }
    red[wave][lane] = acc0 + acc1;
    __syncthreads();
    if (wave != 0) return;
    f32x4 acc[1] = {red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]};
    if (S.raw) {
        if (kv) {
            float* pc = P.C + ((size_t)blockIdx.y * 16 * S.nchunk + row0 + q * 4) * S.Nc + kout;
#pragma unroll
            for (int r = 0; r < 4; ++r) pc[(size_t)r * S.Nc] = acc[0][r];
        }
        return;
    }
    if (!skinny_combine<1>(S, acc, blk, gridDim.x, lane)) return;
    if (kv) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = row0 + q * 4 + r;
            if (row < S.Bm) {
                float* pc = P.C + (size_t)row * S.ldc + kout;
                *pc = acc[0][r] + (S.beta ? *pc : 0.f);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The same products as bodies of a PERSISTENT step kernel (raw mode only; step_prjhyper.hip).  What other workgroups of the
// launch wrote (A) or will read (the partial block) goes through sc1 accesses (scan_common.h's protocol); the weight
// loads of the first round do not depend on the exchange before the phase, so they are issued BEFORE `waitf()` (the wait
// for that exchange) and stream in under the phase before; no wave leaves early (the workgroup meets at barriers
// afterwards); `live` = false: this half-workgroup has no block in this phase and only keeps the barriers company.
// Products, chains and summation order are those of skinny_nn_kernel / skinny_nt_kernel.
template <bool P> __device__ __forceinline__ float ldx(const float* p) {
    if (P) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
template <bool P> __device__ __forceinline__ void stx(float* p, float v) {
    if (P) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
#define SK_OOB 0xfffffff0u      // a buffer offset past every range: the load returns 0

// A wave's slice of W for its first round: the same bytes every step (its block and split are fixed, the weights do
// not change inside a launch), so a kernel whose waves need one round only may load it ONCE and keep it (registers, LDS).
struct SkinnyNNCtx { int g0, g1, klo, khi, col, row0, q, bl; bool cv, rv; };
__device__ __forceinline__ SkinnyNNCtx skinny_nn_ctx(const SkinnyArgs& S, int bx, int by, int bz, bool live) {
    SkinnyNNCtx c;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((threadIdx.x >> 6) & 3);
    c.q = lane >> 4; c.bl = lane & 15;
    c.row0 = (bz % S.nchunk) * 16;
    c.col = bx * 64 + 4 * c.bl;
    c.cv = c.col < S.Nc; c.rv = c.row0 + c.bl < S.Bm;
    const bool zb = bx * 64 >= S.zfrom;
    c.klo = zb ? S.zk0 : 0; c.khi = zb ? S.zk1 : S.K;
    const int ngrp = (c.khi - c.klo + 3) >> 2, nsl = 4 * S.nsplit;
    const int per = (ngrp + nsl - 1) / nsl;
    c.g0 = (by * 4 + wave) * per;
    c.g1 = live ? min(c.g0 + per, ngrp) : c.g0;
    return c;
}
template <int LD>
__device__ __forceinline__ void skinny_nn_wload(const SkinnyArgs& S, const SkinnyNNCtx& c, int bz, int g, f32x4 (&w)[LD]) {
    const SkinnyProb Pb = (bz / S.nchunk) ? S.p[SK_MAXP - 1] : S.p[0];      // SK_MAXP == 2: no dynamic index
    const float* Wp = Pb.W + c.col;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    // k of MFMA step s = 4 i + j for the lanes with MFMA k index q: base + 16 i + 4 q + j, so that a lane's A values of
    // four steps are 16 contiguous bytes (skinny_nn_pbody loads them as one dwordx4: a quarter of the requests)
    const int kend = min(c.khi, c.klo + 4 * c.g1);
#pragma unroll
    for (int s = 0; s < LD; ++s) {
        const int k = c.klo + 4 * g + 16 * (s >> 2) + 4 * c.q + (s & 3);
        w[s] = (k < kend && c.cv) ? *(const f32x4*)(Wp + (size_t)k * S.ldw) : zero;
    }
}
// wget(s): the first round's weights (what skinny_nn_wload(.., g0, w) loads: registers, or read from LDS where they are
// used); red: 3 slots of [4][64] f32x4 per half-workgroup
template <int LD, class WGet, class WaitF>
__device__ __forceinline__ bool skinny_nn_pbody(const SkinnyArgs& S, const SkinnyNNCtx& c, int by, int bz, WGet wget,
                                                f32x4 (*red)[4][64], bool live, WaitF waitf) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((threadIdx.x >> 6) & 3);
    const SkinnyProb Pb = (bz / S.nchunk) ? S.p[SK_MAXP - 1] : S.p[0];
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc[4] = {zero, zero, zero, zero};
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc_(Pb.A, (unsigned)S.Bm * (unsigned)S.lda * 4u);
    const unsigned arow = (unsigned)((c.row0 + c.bl) * S.lda) * 4u;
    if (!waitf()) return false;
    {                                                       // ONE round: g1 - g0 <= LD (the host checks)
        f32x4 a4[LD / 4];
        const int kend = min(c.khi, c.klo + 4 * c.g1);
#pragma unroll
        for (int i = 0; i < LD / 4; ++i) {
            const int k = c.klo + 4 * c.g0 + 16 * i + 4 * c.q;             // k .. k + 3: steps 4 i .. 4 i + 3
            a4[i] = ldb_sc1_f4(rsA, (k < kend && c.rv) ? arow + (unsigned)k * 4u : SK_OOB, 0);
        }
        __builtin_amdgcn_sched_barrier(0);                  // every load in flight before the first MFMA (hipcc sinks them otherwise)
#pragma unroll
        for (int s = 0; s < LD; ++s) {
            const f32x4 ws = wget(s);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[s >> 2][s & 3], ws[j], acc[j], 0, 0, 0);
        }
    }
    if (wave) {
#pragma unroll
        for (int j = 0; j < 4; ++j) red[wave - 1][j][lane] = acc[j];
    }
    __syncthreads();
    if (wave == 0 && live && c.cv) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = acc[j] + red[0][j][lane] + red[1][j][lane] + red[2][j][lane];
        const size_t o = ((size_t)by * 16 * S.nchunk + c.row0 + c.q * 4) * S.Nc + c.col;
        const __amdgpu_buffer_rsrc_t rsC = make_rsrc_(Pb.C, (unsigned)S.nsplit * 16u * S.nchunk * (unsigned)S.Nc * 4u);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            st_sc1_f4(rsC, (unsigned)(o + (size_t)r * S.Nc) * 4u, f32x4{acc[0][r], acc[1][r], acc[2][r], acc[3][r]});
    }
    return true;
}

struct SkinnyNTCtx { int c0, c1, klo, khi, kout, row0, q, bl; bool kv, rv; };
__device__ __forceinline__ SkinnyNTCtx skinny_nt_ctx(const SkinnyArgs& S, int bx, int by, int bz, bool live) {
    SkinnyNTCtx c;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((threadIdx.x >> 6) & 3);
    c.q = lane >> 4; c.bl = lane & 15;
    c.row0 = (bz % S.nchunk) * 16;
    c.kout = bx * 16 + c.bl;
    c.kv = c.kout < S.Nc; c.rv = c.row0 + c.bl < S.Bm;
    const bool zb = bx * 16 < S.zfrom;
    c.klo = zb ? S.zk0 : 0; c.khi = zb ? S.zk1 : S.K;
    const int nch = (c.khi - c.klo + 63) >> 6, nsl = 4 * S.nsplit;
    const int per = (nch + nsl - 1) / nsl;
    c.c0 = (by * 4 + wave) * per;
    c.c1 = live ? min(c.c0 + per, nch) : c.c0;
    return c;
}
template <int CH>
__device__ __forceinline__ void skinny_nt_wload(const SkinnyArgs& S, const SkinnyNTCtx& c, int bz, int ch, f32x4 (&wv)[4 * CH]) {
    const SkinnyProb Pb = (bz / S.nchunk) ? S.p[SK_MAXP - 1] : S.p[0];
    const float* Wp = Pb.W + (size_t)c.kout * S.ldw + c.klo + 4 * c.q;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < CH; ++u) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int off = (ch + u) * 64 + 16 * e;
            const bool ok = (ch + u) < c.c1 && c.klo + off + 4 * c.q < c.khi;
            wv[u * 4 + e] = (ok && c.kv) ? *(const f32x4*)(Wp + off) : zero;
        }
    }
}
// red: 3 slots of [64] f32x4 per half-workgroup
template <int CH, class WGet, class WaitF>
__device__ __forceinline__ bool skinny_nt_pbody(const SkinnyArgs& S, const SkinnyNTCtx& c, int by, int bz, WGet wget,
                                                f32x4 (*red)[64], bool live, WaitF waitf) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((threadIdx.x >> 6) & 3);
    const SkinnyProb Pb = (bz / S.nchunk) ? S.p[SK_MAXP - 1] : S.p[0];
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc0 = zero, acc1 = zero;
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc_(Pb.A, (unsigned)S.Bm * (unsigned)S.lda * 4u);
    const unsigned aoff = (unsigned)((c.row0 + c.bl) * S.lda + c.klo + 4 * c.q) * 4u;
    if (!waitf()) return false;
    {                                                       // ONE round: c1 - c0 <= CH (the host checks)
        f32x4 av[4 * CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int off = (c.c0 + u) * 64 + 16 * e;
                const bool ok = (c.c0 + u) < c.c1 && c.klo + off + 4 * c.q < c.khi;
                av[u * 4 + e] = ldb_sc1_f4(rsA, (ok && c.rv) ? aoff + (unsigned)off * 4u : SK_OOB, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);                  // every load in flight before the first MFMA
#pragma unroll
        for (int v = 0; v < 4 * CH; ++v) {
            const f32x4 ws = wget(v);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[v][0], ws[0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[v][1], ws[1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[v][2], ws[2], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[v][3], ws[3], acc1, 0, 0, 0);
        }
    }
    f32x4 acc = acc0 + acc1;
    if (wave) red[wave - 1][lane] = acc;
    __syncthreads();
    if (wave == 0 && live && c.kv) {
        acc = acc + red[0][lane] + red[1][lane] + red[2][lane];
        float* pc = Pb.C + ((size_t)by * 16 * S.nchunk + c.row0 + c.q * 4) * S.Nc + c.kout;
#pragma unroll
        for (int r = 0; r < 4; ++r) stx<true>(pc + (size_t)r * S.Nc, acc[r]);
    }
    return true;
}

// splits of the reduction axis for a launch of np problems: every wave keeps at least one full round of loads
static inline int skinny_nsplit(bool nt, int K, int Nc, int np, int target_wgs) {
    const int nblk = nt ? (Nc + 15) / 16 : (Nc + 63) / 64;
    const int maxsplit = nt ? K / 256 : K / 64;
    int ns = target_wgs / (np * nblk);
    if (ns > maxsplit) ns = maxsplit;
    return ns < 1 ? 1 : ns;
}

// np problems of one shape (the directions) in one launch.  S.nsplit > 0 is the caller's choice (raw mode: its partial
// buffers are sized for it), S.nsplit <= 0 lets this function choose (in-launch combine only).  S.zfrom must be set (skinny_no_zero_block() for a dense W).
static inline void skinny_no_zero_block(SkinnyArgs& S, bool nt) { S.zfrom = nt ? 0 : 0x7fffffff; S.zk0 = 0; S.zk1 = 0; }
static int skinny(bool nt, SkinnyArgs& S, int np, hipStream_t st) {
    const int nblk = nt ? (S.Nc + 15) / 16 : (S.Nc + 63) / 64;
    S.nchunk = (S.Bm + 15) / 16;
    if (S.nchunk < 1) S.nchunk = 1;
    const int npz = np * S.nchunk;         // (direction, row chunk) pairs: grid z
    int ns = S.nsplit;
    if (ns <= 0) {                         // combine inside the launch: bounded by the ticket and partial buffers
        ns = skinny_nsplit(nt, S.K, S.Nc, npz, SK_TARGET_WGS);
        if (npz * nblk > SK_TICKETS) ns = 1;
        while (ns > 1 && (size_t)npz * nblk * ns * (nt ? 256 : 1024) > SK_PART_FLOATS) --ns;
    }
    if (ns > 1 && !S.raw && (!S.part || !S.tickets)) return ASR_ERR_ARG;
    S.nsplit = ns;
    if (nt) hipLaunchKernelGGL(skinny_nt_kernel, dim3(nblk, ns, npz), dim3(256), 0, st, S);
    else hipLaunchKernelGGL(skinny_nn_kernel, dim3(nblk, ns, npz), dim3(256), 0, st, S);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}

// sum over a workgroup of NW waves (sm: NW floats); every thread of the workgroup calls it
template <int NW = 4>
__device__ __forceinline__ float block_sum(float v, float* sm) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = sm[0] + sm[1] + sm[2] + sm[3];
    if (NW == 8) r += sm[4] + sm[5] + sm[6] + sm[7];
    return r;
}

#define PRJ_MAXU 8      // units per thread: N <= 2048

