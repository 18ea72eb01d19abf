// K1: exact-fp32 MFMA GEMM for the all-timestep projections and the weight-gradient contractions.
//   C[M,N] = alpha * op(A)[M,K] * op(B)[K,N] + beta * C  (+ bias[N] broadcast over rows)
// op = identity or transpose, row-major storage with leading dimensions.  Arithmetic is
// v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: bit-for-bit an fmaf chain, no reduced precision).
// Tile: 128x128x16 per 256-thread workgroup, 2x2 waves, each wave 2x2 MFMA tiles of 32x32.
// LDS images are k-major (As[k][m], Bs[k][n]) so that every MFMA operand read is a conflict-free
// ds_read_b32 of 32 consecutive floats per lane half; double-buffered.
#include "common.h"

#define BM 128
#define BN 128
#define BK 16
#define LDSP 132   // padded row (floats): 128 + 4 keeps 16-B alignment and breaks bank aliasing

// Optional bf16 mode (BASELINE.json north_star: "a stated bound in the optional bf16 mode"): the same kernel
// with BF = true rounds both operands to bf16 (round to nearest even, v_cvt_pk_bf16_f32) while staging them
// into LDS and multiplies with v_mfma_f32_32x32x16_bf16, fp32 accumulate; HBM tensors stay fp32.  LDS image
// [k / 8][row][k % 8]: a lane's 8 consecutive k of its row are one conflict-free ds_read_b128.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// element (r,c) of op(X): X stored row-major with leading dim ld; if TR, op(X)[r][c] = X[c][r]
template <bool TR>
__device__ __forceinline__ float ldg_op(const float* X, int ld, int r, int c, int R, int C) {
    if (r >= R || c >= C) return 0.0f;
    return TR ? X[(size_t)c * ld + r] : X[(size_t)r * ld + c];
}

template <bool TA, bool TB, bool BF>
__global__ __launch_bounds__(256) void gemm_f32_kernel(int M, int N, int K, float alpha,
                                                        const float* __restrict__ A, int lda,
                                                        const float* __restrict__ B, int ldb,
                                                        float beta, float* __restrict__ C, int ldc,
                                                        const float* __restrict__ bias, int ntn,
                                                        int kchunk, float* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) float As[2][BK][LDSP];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][LDSP];
    __bf16 (*Ah)[2][BM][8] = (__bf16 (*)[2][BM][8])&As[0][0][0];      // bf16 images in the same storage
    __bf16 (*Bh)[2][BN][8] = (__bf16 (*)[2][BN][8])&Bs[0][0][0];

    // XCD-aware tile order: blocks b and b+8 share an L2; give each XCD a contiguous strip of
    // tiles along N so the A row-panel is re-used from that L2.
    int nblk = gridDim.x;
    int bid = blockIdx.x;
    {
        int q = nblk / 8, r = nblk % 8, x = bid % 8;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + bid / 8;
    }
    const int tm = bid / ntn, tn = bid % ntn;
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;       // 2x2 waves, each 64x64

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // staging registers: each thread moves 2 x 4 A and 2 x 4 B elements per K-tile, as ONE 16-byte global
    // load per group of 4 along the operand's contiguous dimension whenever the group is in range and
    // 16-byte aligned (wave-uniform test of base pointer and leading dimension), element-wise otherwise
    f32x4 ra[2], rb[2];
    const bool vecA = ((((uintptr_t)A) & 15) == 0) && ((lda & 3) == 0);
    const bool vecB = ((((uintptr_t)B) & 15) == 0) && ((ldb & 3) == 0);
    // X stored [R][C] row-major (C contiguous); group of 4 along C starting at (r, c)
    auto ld4 = [&](const float* X, int ld, int r, int c, int R, int C, bool vec) -> f32x4 {
        if (vec && r < R && c + 3 < C) return *(const f32x4*)(X + (size_t)r * ld + c);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r < R) {
#pragma unroll
            for (int i = 0; i < 4; ++i) if (c + i < C) v[i] = X[(size_t)r * ld + c + i];
        }
        return v;
    };
    // interior tiles (whole tile in range, all loads aligned): straight-line 16-byte loads with no
    // bounds logic at all.  Any branch behind a load makes hipcc wait for the load at the join, which
    // would put the global latency of every K-tile in front of its MFMAs instead of behind them.
    auto load_tiles_fast = [&](int k0) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int idx = e * 256 + tid;
            if (TA) ra[e] = *(const f32x4*)(A + (size_t)(k0 + idx / 32) * lda + m0 + (idx % 32) * 4);
            else    ra[e] = *(const f32x4*)(A + (size_t)(m0 + idx / 4) * lda + k0 + (idx % 4) * 4);
            if (TB) rb[e] = *(const f32x4*)(B + (size_t)(n0 + idx / 4) * ldb + k0 + (idx % 4) * 4);
            else    rb[e] = *(const f32x4*)(B + (size_t)(k0 + idx / 32) * ldb + n0 + (idx % 32) * 4);
        }
    };
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int idx = e * 256 + tid;
            if (TA) ra[e] = ld4(A, lda, k0 + idx / 32, m0 + (idx % 32) * 4, K, M, vecA);     // stored [K][M]
            else    ra[e] = ld4(A, lda, m0 + idx / 4, k0 + (idx % 4) * 4, M, K, vecA);       // stored [M][K]
            if (TB) rb[e] = ld4(B, ldb, n0 + idx / 4, k0 + (idx % 4) * 4, N, K, vecB);       // stored [N][K]
            else    rb[e] = ld4(B, ldb, k0 + idx / 32, n0 + (idx % 32) * 4, K, N, vecB);     // stored [K][N]
        }
    };
    auto store_tiles = [&](int buf) {
        if (BF) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int idx = e * 256 + tid;
                if (TA) {                                   // 4 consecutive rows of one k
                    const int k = idx / 32, m4 = (idx % 32) * 4;
#pragma unroll
                    for (int i = 0; i < 4; ++i) Ah[buf][k >> 3][m4 + i][k & 7] = (__bf16)ra[e][i];
                } else {                                    // 4 consecutive k of one row: one 8-byte store
                    const int m = idx / 4, k4 = (idx % 4) * 4;
                    *(bf16x4*)&Ah[buf][k4 >> 3][m][k4 & 7] = bf16x4{(__bf16)ra[e][0], (__bf16)ra[e][1], (__bf16)ra[e][2], (__bf16)ra[e][3]};
                }
                if (TB) {
                    const int n = idx / 4, k4 = (idx % 4) * 4;
                    *(bf16x4*)&Bh[buf][k4 >> 3][n][k4 & 7] = bf16x4{(__bf16)rb[e][0], (__bf16)rb[e][1], (__bf16)rb[e][2], (__bf16)rb[e][3]};
                } else {
                    const int k = idx / 32, n4 = (idx % 32) * 4;
#pragma unroll
                    for (int i = 0; i < 4; ++i) Bh[buf][k >> 3][n4 + i][k & 7] = (__bf16)rb[e][i];
                }
            }
            return;
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int idx = e * 256 + tid;
            if (TA) *(f32x4*)&As[buf][idx / 32][(idx % 32) * 4] = ra[e];
            else {
#pragma unroll
                for (int i = 0; i < 4; ++i) As[buf][(idx % 4) * 4 + i][idx / 4] = ra[e][i];
            }
            if (TB) {
#pragma unroll
                for (int i = 0; i < 4; ++i) Bs[buf][(idx % 4) * 4 + i][idx / 4] = rb[e][i];
            } else *(f32x4*)&Bs[buf][idx / 32][(idx % 32) * 4] = rb[e];
        }
    };

    // split-K: blockIdx.y owns k in [kbeg, kend); partial tiles go to `part` and are summed in a
    // fixed order by gemm_splitk_reduce_kernel (deterministic, no atomics)
    const int kbeg = blockIdx.y * kchunk;
    const int kend = (kbeg + kchunk < K) ? kbeg + kchunk : K;
    const int Ksave = K;
    K = kend;                                  // loads beyond kend read as zero
    const int nk = (kend - kbeg + BK - 1) / BK;
    const int lr = lane & 31, lk = lane >> 5;
    auto mfma_tile = [&](int buf) {
        if (BF) {                                // K = 16 per instruction: lane half lk holds k = 8 lk .. 8 lk + 7
            const bf16x8 a0 = *(const bf16x8*)&Ah[buf][lk][wm * 64 + lr][0], a1 = *(const bf16x8*)&Ah[buf][lk][wm * 64 + 32 + lr][0];
            const bf16x8 b0 = *(const bf16x8*)&Bh[buf][lk][wn * 64 + lr][0], b1 = *(const bf16x8*)&Bh[buf][lk][wn * 64 + 32 + lr][0];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
            return;
        }
        // all LDS operands of the tile are requested up front (32 VGPRs): LDS returns in order, so the
        // first MFMAs start after the first reads and the rest of the read latency hides under them
        float a0[BK / 2], a1[BK / 2], b0[BK / 2], b1[BK / 2];
#pragma unroll
        for (int k2 = 0; k2 < BK / 2; ++k2) {
            a0[k2] = As[buf][2 * k2 + lk][wm * 64 + lr];
            a1[k2] = As[buf][2 * k2 + lk][wm * 64 + 32 + lr];
            b0[k2] = Bs[buf][2 * k2 + lk][wn * 64 + lr];
            b1[k2] = Bs[buf][2 * k2 + lk][wn * 64 + 32 + lr];
        }
#pragma unroll
        for (int k2 = 0; k2 < BK / 2; ++k2) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[k2], b0[k2], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[k2], b1[k2], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[k2], b0[k2], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[k2], b1[k2], acc[1][1], 0, 0, 0);
        }
    };
    // workgroup-uniform: the whole tile is in range and aligned; only a last partial K-tile is guarded
    const int nk_full = (kend - kbeg) / BK;
    const bool interior = vecA && vecB && (m0 + BM <= M) && (n0 + BN <= N) && nk_full > 0;
    if (interior) {
        load_tiles_fast(kbeg);
        store_tiles(0);
        __syncthreads();
        for (int kt = 0; kt + 1 < nk_full; ++kt) {
            load_tiles_fast(kbeg + (kt + 1) * BK);          // in flight under this tile's 32 MFMAs:
            __builtin_amdgcn_sched_barrier(0);              // keep the scheduler from sinking the loads
            mfma_tile(kt & 1);                              // to their first use behind the MFMAs
            __builtin_amdgcn_sched_barrier(0);
            store_tiles((kt & 1) ^ 1);
            __syncthreads();
        }
        if (nk > nk_full) {                                 // K tail: one guarded tile
            load_tiles(kbeg + nk_full * BK);
            mfma_tile((nk_full - 1) & 1);
            store_tiles(nk_full & 1);
            __syncthreads();
            mfma_tile(nk_full & 1);
        } else {
            mfma_tile((nk_full - 1) & 1);
        }
    } else {
        load_tiles(kbeg);
        store_tiles(0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nk) load_tiles(kbeg + (kt + 1) * BK);
            mfma_tile(buf);
            if (kt + 1 < nk) store_tiles(buf ^ 1);
            __syncthreads();
        }
    }

    K = Ksave;
    if (part) {
        float* P = part + (size_t)blockIdx.y * M * N;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = n0 + wn * 64 + j * 32 + (lane & 31);
                if (col >= N) continue;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                    if (row < M) P[(size_t)row * N + col] = acc[i][j][e];
                }
            }
        return;
    }
    // epilogue.  C/D map of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + (lane & 31);
            if (col >= N) continue;
            const float bv = bias ? bias[col] : 0.0f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (row < M) {
                    float* p = C + (size_t)row * ldc + col;
                    float v = alpha * acc[i][j][e] + bv;
                    if (beta != 0.0f) v += beta * (*p);
                    *p = v;
                }
            }
        }
}

__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(const float* __restrict__ part, int ns,
                                                                  int M, int N, float alpha, float beta,
                                                                  float* __restrict__ C, int ldc,
                                                                  const float* __restrict__ bias) {
    const size_t tot = (size_t)M * N;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < tot; i += (size_t)gridDim.x * 256) {
        float acc = 0.f;
        int s = 0;
        for (; s + 8 <= ns; s += 8) {                   // 8 partials in flight, summed in split order
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = part[(size_t)(s + k) * tot + i];
#pragma unroll
            for (int k = 0; k < 8; ++k) acc += v[k];
        }
        for (; s < ns; ++s) acc += part[(size_t)s * tot + i];
        const int row = (int)(i / N), col = (int)(i % N);
        float* p = C + (size_t)row * ldc + col;
        float v = alpha * acc + (bias ? bias[col] : 0.f);
        if (beta != 0.f) v += beta * (*p);
        *p = v;
    }
}

// ws (optional): scratch for split-K partials.  Split-K is used when the output has too few tiles to
// fill the chip and K is long (the weight-gradient contractions, K = T*B).
static int gemm_launch(bool bf16, int transA, int transB, int M, int N, int K, float alpha, const float* A,
                       int lda, const float* B, int ldb, float beta, float* C, int ldc,
                       const float* bias, float* ws, size_t ws_bytes, hipStream_t stream) {
    if (M < 0 || N < 0 || K < 0 || !C) return ASR_ERR_ARG;
    if (M == 0 || N == 0) return ASR_OK;
    if (K > 0 && (!A || !B)) return ASR_ERR_ARG;
    const int ntm = (M + BM - 1) / BM, ntn = (N + BN - 1) / BN;
    const int tiles = ntm * ntn;
    int ns = 1;
    // second clause: the small problems of a TBPTT window (M = 1600 rows or K = 1600); kept to M <= 2048 so that the
    // forward GEMMs of whole utterance batches are never split (their K is 123..1024): a row's posteriors then do
    // not depend, even by rounding, on how many rows share the batch (tests/test_gpu_fullsize.py)
    if (ws && ((tiles < 768 && K >= 2048) || (tiles < 256 && K >= 512 && M <= 2048))) {                    // fewer than 3 workgroups per CU: split K so
        ns = (1024 + tiles / 2) / tiles;                     // that ~4 per CU are in flight (a 128-tile
                                                             // dW_x ran on half the chip: 40 TFLOP/s)
        const int maxk = K / (BK * 8);                       // at least 8 K-tiles per split
        if (ns > maxk) ns = maxk;
        if (ns > 64) ns = 64;
        const size_t cap = ws_bytes / ((size_t)M * N * sizeof(float));
        if ((size_t)ns > cap) ns = (int)cap;
        if (ns < 1) ns = 1;
    }
    int kchunk = K > 0 ? K : 1;
    if (ns > 1) {
        kchunk = (((K + ns - 1) / ns + BK - 1) / BK) * BK;
        ns = (K + kchunk - 1) / kchunk;
    }
    float* part = ns > 1 ? ws : nullptr;
    dim3 grid(tiles, ns), block(256);
#define LAUNCH(TA_, TB_)                                                                                \
    do {                                                                                                \
        if (bf16) hipLaunchKernelGGL((gemm_f32_kernel<TA_, TB_, true>), grid, block, 0, stream, M, N, K, alpha, A, \
                                     lda, B, ldb, beta, C, ldc, bias, ntn, kchunk, part);               \
        else hipLaunchKernelGGL((gemm_f32_kernel<TA_, TB_, false>), grid, block, 0, stream, M, N, K, alpha, A, \
                                lda, B, ldb, beta, C, ldc, bias, ntn, kchunk, part);                    \
    } while (0)
    if (!transA && !transB) LAUNCH(false, false);
    else if (!transA && transB) LAUNCH(false, true);
    else if (transA && !transB) LAUNCH(true, false);
    else LAUNCH(true, true);
#undef LAUNCH
    ASR_CHECK_LAUNCH();
    if (ns > 1) {
        const size_t tot = (size_t)M * N;
        const int g = (int)((tot + 255) / 256 < 2048 ? (tot + 255) / 256 : 2048);
        hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(g), dim3(256), 0, stream, part, ns, M, N, alpha,
                           beta, C, ldc, bias);
        ASR_CHECK_LAUNCH();
    }
    return ASR_OK;
}

extern "C" int asr_gemm_f32_ws(int transA, int transB, int M, int N, int K, float alpha, const float* A,
                               int lda, const float* B, int ldb, float beta, float* C, int ldc,
                               const float* bias, float* ws, size_t ws_bytes, hipStream_t stream) {
    return gemm_launch(false, transA, transB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, ws, ws_bytes, stream);
}
// the optional bf16 mode: fp32 tensors in memory, operands rounded to bf16 on the way into LDS, fp32 accumulate
extern "C" int asr_gemm_bf16_ws(int transA, int transB, int M, int N, int K, float alpha, const float* A,
                                int lda, const float* B, int ldb, float beta, float* C, int ldc,
                                const float* bias, float* ws, size_t ws_bytes, hipStream_t stream) {
    return gemm_launch(true, transA, transB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, ws, ws_bytes, stream);
}

extern "C" int asr_gemm_f32(int transA, int transB, int M, int N, int K, float alpha, const float* A,
                            int lda, const float* B, int ldb, float beta, float* C, int ldc,
                            const float* bias, hipStream_t stream) {
    return asr_gemm_f32_ws(transA, transB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, nullptr, 0,
                           stream);
}
