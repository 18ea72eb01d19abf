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

__device__ __forceinline__ float block_sum(float v, float* sm) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    return sm[0] + sm[1] + sm[2] + sm[3];
}

#define PRJ_MAXU 8      // units per thread: N <= 2048

