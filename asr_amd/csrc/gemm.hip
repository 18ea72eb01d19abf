// K1: fp32 GEMM for the all-timestep projections and the weight-gradient contractions: gemm_f32_kernel on the fp32
// matrix cores, and gemm_split_kernel (further down) with fp32-grade products on the bf16 matrix cores.
//   C[M,N] = alpha * op(A)[M,K] * op(B)[K,N] + beta * C  (+ bias[N] broadcast over rows)
// op = identity or transpose, row-major storage with leading dimensions.  Arithmetic is
// v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: bit-for-bit an fmaf chain, no reduced precision).
// Tile: 128x128x16 per 256-thread workgroup, 2x2 waves, each wave 2x2 MFMA tiles of 32x32.
// LDS images are k-major (As[k][m], Bs[k][n]) so that every MFMA operand read is a conflict-free
// ds_read_b32 of 32 consecutive floats per lane half; double-buffered.
#include "common.h"
#include <stdlib.h>

#define BM 128
#define BN 128
#define BK 16
#define LDSP 132   // padded row (floats): 128 + 4 keeps 16-B alignment and breaks bank aliasing

// The bf16 matrix cores are used by gemm_split_kernel below: for the optional bf16 mode (BASELINE.json north_star:
// "a stated bound in the optional bf16 mode") and for fp32-grade products from split operands.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// element (r,c) of op(X): X stored row-major with leading dim ld; if TR, op(X)[r][c] = X[c][r]
template <bool TR>
__device__ __forceinline__ float ldg_op(const float* X, int ld, int r, int c, int R, int C) {
    if (r >= R || c >= C) return 0.0f;
    return TR ? X[(size_t)c * ld + r] : X[(size_t)r * ld + c];
}

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_f32_kernel(int M, int N, int K, float alpha,
                                                        const float* __restrict__ A, int lda,
                                                        const float* __restrict__ B, int ldb,
                                                        float beta, float* __restrict__ C, int ldc,
                                                        const float* __restrict__ bias, int ntn,
                                                        int kchunk, float* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) float As[2][BK][LDSP];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][LDSP];

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

// ------------------------------------------------------------------------------------------------
// fp32-grade GEMM on the bf16 matrix cores ("bf16x3"): every fp32 operand element x is split while it is staged
// into LDS into three bf16 values x = h + m + l (h = rn(x), m = rn(x - h), l = rn(x - h - m); the two
// differences are exact in fp32 and 3 x 8 significand bits cover fp32's 24), and the product is the sum
// of the partial products a_i * b_j, each exact in the fp32 accumulator of v_mfma_f32_32x32x16_bf16:
//   TERMS = 8: all but a_l * b_l          -> dropped part <= 2^-32 |a b|, below fp32's own product rounding
//   TERMS = 6: also without a_m*b_l, a_l*b_m -> dropped part <= 2^-23 |a b| worst case
//   TERMS = 3: a_h*b_h + a_h*b_m + a_m*b_h from TWO parts per operand ("bf16x2": 16 significand bits per operand,
//              dropped part <= 2^-15 |a b|): the reduced-precision mode with a bound for every config, layer-norm
//              stacks included (tests/test_gpu_bf16.py)
//   TERMS = 1: a_h * b_h only (the plain bf16 mode)
// 6 bf16 MFMAs (32 cycles each) replace the 8 v_mfma_f32_32x32x2_f32 (64 cycles each) of a 32x32x16 block.  How the
// partial products are summed matters more than how many there are (see mfma_tile): measured against float64 on
// Gaussian operands, K = 1024, the kernel has rms error 8.6e-8 and mean error 2e-10 of the result's rms, the fp32
// MFMA kernel 2.1e-7 and 3e-11 (tools/ubench/gemm_split_rms.py, tests/test_gpu_kernels.py); 150-157 TFLOP/s on the
// cfg3 shapes against 104-120.  A second set of accumulators for the correction terms (kept across the K loop) cost
// the second resident workgroup per CU (118 TFLOP/s).
// LDS images per operand and split part, chosen by the operand's storage order so that both the global
// loads and the LDS stores are 16-byte / conflict-free:
//   k contiguous in memory ([row][k]):  [k / 8][row][k % 8] bf16 (k halves 64 B apart in bank space), written 8 B per
//                                        lane, read ds_read_b128
//   row contiguous in memory ([k][row]): [k][row] bf16 with 320-byte k rows, written 8 B per lane (4 rows of one
//                                        k), read with the transposing ds_read_b64_tr_b16 (4 k x 16 rows per
//                                        16-lane group): two reads give a lane the 8 k of its row
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
#define SPL_KROW 320                      // bytes per k row of the row-contiguous image: 128 bf16 + 64 B, so that the
                                          // 4 k rows x 2 column groups of a transposed read fall on 64 distinct banks
#define SPL_IMG (16 * SPL_KROW)           // bytes per image (>= 2 * SPL_KHALF of the other layout)
#define SPL_KHALF (128 * 16 + 64)         // bytes per k half of the k-contiguous image: + 16 banks, so that the two
                                          // k halves a 16-lane group of ds_write_b64 touches do not share banks

// two fp32 values -> packed bf16 pairs of their split parts (one v_cvt_pk_bf16_f32 per part)
template <int NP>
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned* out) {
    out[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{x0, x1}, bf16x2));
#if defined(SPL_ABL) && SPL_ABL == 1
    if (NP > 1) { out[1] = out[0]; out[2] = out[0]; return; }
#endif
    if (NP > 1) {
        const float r0 = x0 - __uint_as_float(out[0] << 16), r1 = x1 - __uint_as_float(out[0] & 0xffff0000u);
        out[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r0, r1}, bf16x2));
        if (NP > 2) {
            const float q0 = r0 - __uint_as_float(out[1] << 16), q1 = r1 - __uint_as_float(out[1] & 0xffff0000u);
            out[2] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{q0, q1}, bf16x2));
        }
    }
}

// PA / PB: the operand arrives PRE-SPLIT (asr_split_bf16x3: three bf16 planes of the tensor's own row-major shape,
// rows padded with zeros to a multiple of 8 columns; `A` / `B` then point at plane 0, lda / ldb count bf16 elements
// and plA / plB are the plane strides).  Staging such an operand is one 16-byte load and one ds_write_b128 per
// thread, part and K-tile with no arithmetic (odd tiles flip the sign bits of A's parts), where the on-the-fly split
// costs ~13 VALU instructions per element pair: a tensor that feeds several products (a layer's input: x.W_x,
// x.W_xhat and the two weight gradients) is split once.  Products and their order are the same as on the fly:
// results are bit-identical.
#ifdef GEMM_STAMPS
__device__ unsigned long long gemm_stamp_acc[8];
#endif
template <bool TA, bool TB, int TERMS, bool PA = false, bool PB = false>
__global__ __launch_bounds__(256, 2) void gemm_split_kernel(int M, int N, int K, float alpha,
                                                          const float* __restrict__ A, int lda,
                                                          const float* __restrict__ B, int ldb,
                                                          float beta, float* __restrict__ C, int ldc,
                                                          const float* __restrict__ bias, int ntn,
                                                          int kchunk, float* __restrict__ part, size_t plA, size_t plB) {
    constexpr int NP = TERMS == 1 ? 1 : TERMS == 3 ? 2 : 3;
    static_assert(!(PA || PB) || TERMS == 6, "pre-split operands: the 6-product form only");
    constexpr bool AKM = TA, BKM = !TB;        // operand stored [k][row] (row contiguous)
    __shared__ __attribute__((aligned(16))) unsigned char sm[2][2][NP][SPL_IMG];   // [buf][A|B][part]

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
    const int wm = wave >> 1, wn = wave & 1;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    f32x4 ra[2], rb[2];
    u32x4s qa[NP], qb[NP];                                 // pre-split operands: 8 bf16 per thread and part
    const bool vecA = ((((uintptr_t)A) & 15) == 0) && ((lda & (PA ? 7 : 3)) == 0);
    const bool vecB = ((((uintptr_t)B) & 15) == 0) && ((ldb & (PB ? 7 : 3)) == 0);
    // pre-split operand, stored [k][row] (kmaj): thread -> k = tid / 16, rows 8 (tid % 16) ..;  stored [row][k]: row =
    // tid / 2, k = 8 (tid % 2) ..  Columns between the matrix and its padded width hold zeros.
    auto load_pre = [&](u32x4s* q, const float* X, int ld, size_t pl, bool kmaj, int k0, int r0, int nrows, int kdim,
                        bool fast) {
        const unsigned short* P = (const unsigned short*)X;
        const int rr = kmaj ? k0 + (tid >> 4) : r0 + (tid >> 1);
        const int cc = kmaj ? r0 + (tid & 15) * 8 : k0 + (tid & 1) * 8;
        const bool ok = fast || (kmaj ? (rr < kdim && cc < ld) : (rr < nrows && cc < ld && cc < kdim));
#pragma unroll
        for (int p = 0; p < NP; ++p)
            q[p] = ok ? *(const u32x4s*)(P + (size_t)p * pl + (size_t)rr * ld + cc) : u32x4s{0, 0, 0, 0};
    };
    auto store_pre = [&](unsigned char (*img)[SPL_IMG], const u32x4s* q, bool kmaj, bool neg) {
        const size_t off = kmaj ? (size_t)(tid >> 4) * SPL_KROW + (tid & 15) * 16
                                : (size_t)(tid & 1) * SPL_KHALF + (tid >> 1) * 16;
        const unsigned flip = neg ? 0x80008000u : 0u;
#pragma unroll
        for (int p = 0; p < NP; ++p) *(u32x4s*)(img[p] + off) = q[p] ^ flip;
    };
    auto ld4 = [&](const float* X, int ld, int r, int c, int R, int Cc, bool vec) -> f32x4 {
        if (vec && r < R && c + 3 < Cc) return *(const f32x4*)(X + (size_t)r * ld + c);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r < R) {
#pragma unroll
            for (int i = 0; i < 4; ++i) if (c + i < Cc) v[i] = X[(size_t)r * ld + c + i];
        }
        return v;
    };
    // thread -> elements.  [k][row] storage: k pair kp = tid / 32 (rows k0 + 2 kp, + 1), 4 columns at 4 (tid % 32).
    //                      [row][k] storage: e = 0, 1: row (e * 256 + tid) / 4, 4 k at 4 ((e * 256 + tid) % 4).
    const int kp = tid >> 5, c4 = (tid & 31) * 4;
    // interior tiles: per-thread pointers that walk along K (set behind `kbeg` below), so that an iteration's loads
    // cost no 64-bit address arithmetic (in-kernel stamps: 359 of an iteration's 3900 cycles went into forming them)
    const float* wa[2]; const float* wb[2];
    const unsigned short* wpa = nullptr; const unsigned short* wpb = nullptr;
    size_t stepA = 0, stepB = 0;
    auto walk_init = [&](int k0) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int idx = e * 256 + tid;
            wa[e] = AKM ? A + (size_t)(k0 + 2 * kp + e) * lda + m0 + c4 : A + (size_t)(m0 + idx / 4) * lda + k0 + (idx % 4) * 4;
            wb[e] = BKM ? B + (size_t)(k0 + 2 * kp + e) * ldb + n0 + c4 : B + (size_t)(n0 + idx / 4) * ldb + k0 + (idx % 4) * 4;
        }
        if (PA) wpa = (const unsigned short*)A + (AKM ? (size_t)(k0 + (tid >> 4)) * lda + m0 + (tid & 15) * 8
                                                      : (size_t)(m0 + (tid >> 1)) * lda + k0 + (tid & 1) * 8);
        if (PB) wpb = (const unsigned short*)B + (BKM ? (size_t)(k0 + (tid >> 4)) * ldb + n0 + (tid & 15) * 8
                                                      : (size_t)(n0 + (tid >> 1)) * ldb + k0 + (tid & 1) * 8);
        stepA = AKM ? (size_t)BK * lda : BK;
        stepB = BKM ? (size_t)BK * ldb : BK;
    };
    auto load_tiles_fast = [&](f32x4* ra, f32x4* rb) {      // loads the tile the pointers stand on, then steps them
        if (PA) {
#pragma unroll
            for (int p = 0; p < NP; ++p) qa[p] = *(const u32x4s*)(wpa + (size_t)p * plA);
            wpa += stepA;
        }
        if (PB) {
#pragma unroll
            for (int p = 0; p < NP; ++p) qb[p] = *(const u32x4s*)(wpb + (size_t)p * plB);
            wpb += stepB;
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            if (!PA) { ra[e] = *(const f32x4*)wa[e]; wa[e] += stepA; }
            if (!PB) { rb[e] = *(const f32x4*)wb[e]; wb[e] += stepB; }
        }
    };
    auto load_tiles = [&](int k0) {
        if (PA) load_pre(qa, A, lda, plA, AKM, k0, m0, M, K, false);
        if (PB) load_pre(qb, B, ldb, plB, BKM, k0, n0, N, K, false);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int idx = e * 256 + tid;
            if (!PA) {
                if (AKM) ra[e] = ld4(A, lda, k0 + 2 * kp + e, m0 + c4, K, M, vecA);
                else     ra[e] = ld4(A, lda, m0 + idx / 4, k0 + (idx % 4) * 4, M, K, vecA);
            }
            if (!PB) {
                if (BKM) rb[e] = ld4(B, ldb, k0 + 2 * kp + e, n0 + c4, K, N, vecB);
                else     rb[e] = ld4(B, ldb, n0 + idx / 4, k0 + (idx % 4) * 4, N, K, vecB);
            }
        }
    };
    auto store_op = [&](unsigned char (*img)[SPL_IMG], const f32x4* r, bool kmaj, float sg) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            unsigned w0[3], w1[3];
            split_pair<NP>(r[e][0] * sg, r[e][1] * sg, w0);
            split_pair<NP>(r[e][2] * sg, r[e][3] * sg, w1);
            const int idx = e * 256 + tid;
            // [k][row] storage: 4 rows c4.. of k = 2 kp + e;   [row][k] storage: row idx / 4, k = 4 (idx % 4) ..
            const size_t off = kmaj ? (size_t)(2 * kp + e) * SPL_KROW + c4 * 2
                                    : (size_t)(((idx % 4) * 4) >> 3) * SPL_KHALF + (idx / 4) * 16 + (((idx % 4) * 4) & 7) * 2;
#pragma unroll
            for (int p = 0; p < NP; ++p) *(u32x2*)(img[p] + off) = u32x2{w0[p], w1[p]};
        }
    };
    // K-tile t enters with the sign (-1)^t on its A operand and is joined with the same sign: see mfma_tile
    auto sgn = [](int t) { return (TERMS > 1 && (t & 1)) ? -1.0f : 1.0f; };
    auto store_tiles = [&](int t) {
        if (PA) store_pre(sm[t & 1][0], qa, AKM, TERMS > 1 && (t & 1));
        else store_op(sm[t & 1][0], ra, AKM, sgn(t));
        if (PB) store_pre(sm[t & 1][1], qb, BKM, false);
        else store_op(sm[t & 1][1], rb, BKM, 1.0f);
    };
    const int lr = lane & 31, lk = lane >> 5;
    // transposed read: lane 4q + p of a 16-lane group addresses k row q, rows 4p .. 4p + 3 of the group's 16 rows
    // and receives, for row (lane & 15) of the group, the 4 k of the block (cdna_hip_programming.md T10)
    const int tr_off = (8 * lk + ((lane >> 2) & 3)) * SPL_KROW + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    auto frag = [&](const unsigned char* img, int row0, bool kmaj) -> bf16x8 {
        if (kmaj) {
            typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
            const unsigned char* p = img + tr_off + row0 * 2;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)p);
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p + 4 * SPL_KROW));
            return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
        return *(const bf16x8*)(img + (size_t)lk * SPL_KHALF + (row0 + lr) * 16);
    };
#define SPL_MFMA(acc_, a_, b_) acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_, b_, acc_, 0, 0, 0)
    auto mfma_tile = [&](int tile) {
        const int buf = tile & 1;
        const float sg = sgn(tile);
        bf16x8 a[2][NP], b[2][NP];
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#if defined(SPL_ABL) && SPL_ABL == 4
                if (p > 0) { a[i][p] = a[i][0]; b[i][p] = b[i][0]; continue; }
#endif
                a[i][p] = frag(sm[buf][0][p], wm * 64 + i * 32, AKM);
                b[i][p] = frag(sm[buf][1][p], wn * 64 + i * 32, BKM);
            }
        if (TERMS == 1) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) SPL_MFMA(acc[i][j], a[i][0], b[j][0]);
        } else {
            // The bf16 MFMA truncates its fp32 accumulation toward -inf (measured: mean error -0.5 ulp of the accumulator
            // per instruction, where v_mfma_f32_32x32x2_f32 rounds to nearest); a bias of one sign survives an LSTM's
            // cell integration far better than rounding noise.  So the TERMS products of a K-tile form a chain from a
            // zero accumulator and join the running sum with a round-to-nearest v_fma_f32, and K-tiles alternate in
            // sign: odd tiles multiply -A (the split of -x is exactly minus the split of x) and are subtracted, so the
            // truncation pushes the sum down on even tiles and up on odd ones and cancels in expectation.  (Accumulating
            // straight into the running sum and flipping ITS sign every 4 tiles instead saves the 64 joins per tile: rms
            // error 1.7e-7 instead of 8.6e-8, bias -3e-9, 2.5 % faster: not taken.)
            constexpr int pa[8] = {0, 0, 1, 1, 0, 2, 1, 2}, pb[8] = {0, 1, 0, 1, 2, 0, 2, 1};
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f32x16 tmp;
#pragma unroll
                    for (int e = 0; e < 16; ++e) tmp[e] = 0.0f;
#pragma unroll
#if defined(SPL_ABL) && SPL_ABL == 2
                    for (int t = TERMS - 1; t >= 0; --t) SPL_MFMA(acc[i][j], a[i][pa[t] < NP ? pa[t] : 0], b[j][pb[t] < NP ? pb[t] : 0]);
                    continue;
#endif
#if defined(SPL_ABL) && SPL_ABL == 3
                    for (int t = 2; t >= 0; --t) SPL_MFMA(tmp, a[i][t], b[j][t]);
#else
                    for (int t = TERMS - 1; t >= 0; --t) SPL_MFMA(tmp, a[i][pa[t] < NP ? pa[t] : 0], b[j][pb[t] < NP ? pb[t] : 0]);
#endif
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = __builtin_fmaf(tmp[e], sg, acc[i][j][e]);
                }
        }
    };
#undef SPL_MFMA

    const int kbeg = blockIdx.y * kchunk;
    const int kend = (kbeg + kchunk < K) ? kbeg + kchunk : K;
    const int Ksave = K;
    K = kend;
    const int nk = (kend - kbeg + BK - 1) / BK;
    const int nk_full = (kend - kbeg) / BK;
    const bool interior = vecA && vecB && (m0 + BM <= M) && (n0 + BN <= N) && nk_full > 0;
    if (interior) {
        walk_init(kbeg);
        load_tiles_fast(ra, rb);
        store_tiles(0);
        __syncthreads();
        // one iteration per trip, not unrolled: an unrolled pair made the register allocator move the 64 accumulator
        // registers between the halves (v_mov behind the MFMA latency).  Loads two tiles ahead and an MFMA / split-VALU
        // interleave (sched_group_barrier) were measured and changed nothing: 162 vs 170 TFLOP/s
#ifdef GEMM_STAMPS      // diagnostic build only (tools/ubench/gemm_stamps.hip): where an iteration's cycles go
        unsigned long long st_[5] = {0, 0, 0, 0, 0};
#define GST_(i) do { unsigned long long n_ = __builtin_amdgcn_s_memtime(); st_[i] += n_ - sp_; sp_ = n_; } while (0)
        unsigned long long sp_ = __builtin_amdgcn_s_memtime();
#else
#define GST_(i) do { } while (0)
#endif
#pragma unroll 1
        for (int kt = 0; kt + 1 < nk_full; ++kt) {
            load_tiles_fast(ra, rb);
            __builtin_amdgcn_sched_barrier(0);
            GST_(0);
            mfma_tile(kt);
            __builtin_amdgcn_sched_barrier(0);
            GST_(1);
#ifdef GEMM_STAMPS
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            GST_(2);
#endif
            store_tiles(kt + 1);
            GST_(3);
            __syncthreads();
            GST_(4);
        }
#ifdef GEMM_STAMPS
        if (lane == 0)
            for (int i = 0; i < 5; ++i) atomicAdd(&gemm_stamp_acc[i], st_[i]);
        if (tid == 0) atomicAdd(&gemm_stamp_acc[5], (unsigned long long)(nk_full - 1));
#endif
#undef GST_
        if (nk > nk_full) {                                 // K tail: one guarded tile
            load_tiles(kbeg + nk_full * BK);
            mfma_tile(nk_full - 1);
            store_tiles(nk_full);
            __syncthreads();
            mfma_tile(nk_full);
        } else {
            mfma_tile(nk_full - 1);
        }
    } else {
        load_tiles(kbeg);
        store_tiles(0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) load_tiles(kbeg + (kt + 1) * BK);
            mfma_tile(kt);
            if (kt + 1 < nk) store_tiles(kt + 1);
            __syncthreads();
        }
    }
    K = Ksave;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + (lane & 31);
            if (col >= N) continue;
            const float bv = (bias && !part) ? bias[col] : 0.0f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (row >= M) continue;
                const float s = acc[i][j][e];
                if (part) { part[((size_t)blockIdx.y * M + row) * N + col] = s; continue; }
                float* p = C + (size_t)row * ldc + col;
                float v = alpha * s + bv;
                if (beta != 0.0f) v += beta * (*p);
                *p = v;
            }
        }
}

// The same products as gemm_split_kernel from a workgroup of EIGHT waves on the same 128 x 128 x 16 tiles (wave = 64 x 32 of
// the output): half the staging work and half the accumulators per wave, 128 registers per wave, so two workgroups per
// CU are four waves per SIMD instead of two, and a wave's load / split / barrier phases have three other waves' matrix
// phases to hide under.  Products, their order and the K-tile sign alternation are the same: results are bit-identical
// to gemm_split_kernel's.  No pre-split operands.
template <bool TA, bool TB, int TERMS>
__global__ __launch_bounds__(512, 4) void gemm_split8_kernel(int M, int N, int K, float alpha,
                                                          const float* __restrict__ A, int lda,
                                                          const float* __restrict__ B, int ldb,
                                                          float beta, float* __restrict__ C, int ldc,
                                                          const float* __restrict__ bias, int ntn,
                                                          int kchunk, float* __restrict__ part, size_t plA, size_t plB) {
    constexpr int NP = TERMS == 1 ? 1 : TERMS == 3 ? 2 : 3;
    constexpr bool PA = false, PB = false;
    constexpr bool AKM = TA, BKM = !TB;        // operand stored [k][row] (row contiguous)
    __shared__ __attribute__((aligned(16))) unsigned char sm[2][2][NP][SPL_IMG];   // [buf][A|B][part]

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
    const int wm = wave >> 2, wn = wave & 3;

    f32x16 acc[2][1];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 1; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    f32x4 ra[1], rb[1];
    u32x4s qa[NP], qb[NP];                                 // pre-split operands: 8 bf16 per thread and part
    const bool vecA = ((((uintptr_t)A) & 15) == 0) && ((lda & (PA ? 7 : 3)) == 0);
    const bool vecB = ((((uintptr_t)B) & 15) == 0) && ((ldb & (PB ? 7 : 3)) == 0);
    // pre-split operand, stored [k][row] (kmaj): thread -> k = tid / 16, rows 8 (tid % 16) ..;  stored [row][k]: row =
    // tid / 2, k = 8 (tid % 2) ..  Columns between the matrix and its padded width hold zeros.
    auto load_pre = [&](u32x4s* q, const float* X, int ld, size_t pl, bool kmaj, int k0, int r0, int nrows, int kdim,
                        bool fast) {
        const unsigned short* P = (const unsigned short*)X;
        const int rr = kmaj ? k0 + (tid >> 4) : r0 + (tid >> 1);
        const int cc = kmaj ? r0 + (tid & 15) * 8 : k0 + (tid & 1) * 8;
        const bool ok = fast || (kmaj ? (rr < kdim && cc < ld) : (rr < nrows && cc < ld && cc < kdim));
#pragma unroll
        for (int p = 0; p < NP; ++p)
            q[p] = ok ? *(const u32x4s*)(P + (size_t)p * pl + (size_t)rr * ld + cc) : u32x4s{0, 0, 0, 0};
    };
    auto store_pre = [&](unsigned char (*img)[SPL_IMG], const u32x4s* q, bool kmaj, bool neg) {
        const size_t off = kmaj ? (size_t)(tid >> 4) * SPL_KROW + (tid & 15) * 16
                                : (size_t)(tid & 1) * SPL_KHALF + (tid >> 1) * 16;
        const unsigned flip = neg ? 0x80008000u : 0u;
#pragma unroll
        for (int p = 0; p < NP; ++p) *(u32x4s*)(img[p] + off) = q[p] ^ flip;
    };
    auto ld4 = [&](const float* X, int ld, int r, int c, int R, int Cc, bool vec) -> f32x4 {
        if (vec && r < R && c + 3 < Cc) return *(const f32x4*)(X + (size_t)r * ld + c);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r < R) {
#pragma unroll
            for (int i = 0; i < 4; ++i) if (c + i < Cc) v[i] = X[(size_t)r * ld + c + i];
        }
        return v;
    };
    // thread -> elements.  [k][row] storage: k row kp = tid / 32, 4 columns at 4 (tid % 32).
    //                      [row][k] storage: row tid / 4, 4 k at 4 (tid % 4).   (512 threads: one 16-byte piece each)
    const int kp = tid >> 5, c4 = (tid & 31) * 4;
    // interior tiles: per-thread pointers that walk along K (set behind `kbeg` below), so that an iteration's loads
    // cost no 64-bit address arithmetic (in-kernel stamps: 359 of an iteration's 3900 cycles went into forming them)
    const float* wa[2]; const float* wb[2];
    const unsigned short* wpa = nullptr; const unsigned short* wpb = nullptr;
    size_t stepA = 0, stepB = 0;
    auto walk_init = [&](int k0) {
#pragma unroll
        for (int e = 0; e < 1; ++e) {
            const int idx = tid;
            wa[e] = AKM ? A + (size_t)(k0 + kp) * lda + m0 + c4 : A + (size_t)(m0 + idx / 4) * lda + k0 + (idx % 4) * 4;
            wb[e] = BKM ? B + (size_t)(k0 + kp) * ldb + n0 + c4 : B + (size_t)(n0 + idx / 4) * ldb + k0 + (idx % 4) * 4;
        }
        if (PA) wpa = (const unsigned short*)A + (AKM ? (size_t)(k0 + (tid >> 4)) * lda + m0 + (tid & 15) * 8
                                                      : (size_t)(m0 + (tid >> 1)) * lda + k0 + (tid & 1) * 8);
        if (PB) wpb = (const unsigned short*)B + (BKM ? (size_t)(k0 + (tid >> 4)) * ldb + n0 + (tid & 15) * 8
                                                      : (size_t)(n0 + (tid >> 1)) * ldb + k0 + (tid & 1) * 8);
        stepA = AKM ? (size_t)BK * lda : BK;
        stepB = BKM ? (size_t)BK * ldb : BK;
    };
    auto load_tiles_fast = [&](f32x4* ra, f32x4* rb) {      // loads the tile the pointers stand on, then steps them
        if (PA) {
#pragma unroll
            for (int p = 0; p < NP; ++p) qa[p] = *(const u32x4s*)(wpa + (size_t)p * plA);
            wpa += stepA;
        }
        if (PB) {
#pragma unroll
            for (int p = 0; p < NP; ++p) qb[p] = *(const u32x4s*)(wpb + (size_t)p * plB);
            wpb += stepB;
        }
#pragma unroll
        for (int e = 0; e < 1; ++e) {
            if (!PA) { ra[e] = *(const f32x4*)wa[e]; wa[e] += stepA; }
            if (!PB) { rb[e] = *(const f32x4*)wb[e]; wb[e] += stepB; }
        }
    };
    auto load_tiles = [&](int k0) {
        if (PA) load_pre(qa, A, lda, plA, AKM, k0, m0, M, K, false);
        if (PB) load_pre(qb, B, ldb, plB, BKM, k0, n0, N, K, false);
#pragma unroll
        for (int e = 0; e < 1; ++e) {
            const int idx = tid;
            if (!PA) {
                if (AKM) ra[e] = ld4(A, lda, k0 + kp, m0 + c4, K, M, vecA);
                else     ra[e] = ld4(A, lda, m0 + idx / 4, k0 + (idx % 4) * 4, M, K, vecA);
            }
            if (!PB) {
                if (BKM) rb[e] = ld4(B, ldb, k0 + kp, n0 + c4, K, N, vecB);
                else     rb[e] = ld4(B, ldb, n0 + idx / 4, k0 + (idx % 4) * 4, N, K, vecB);
            }
        }
    };
    auto store_op = [&](unsigned char (*img)[SPL_IMG], const f32x4* r, bool kmaj, float sg) {
#pragma unroll
        for (int e = 0; e < 1; ++e) {
            unsigned w0[3], w1[3];
            split_pair<NP>(r[e][0] * sg, r[e][1] * sg, w0);
            split_pair<NP>(r[e][2] * sg, r[e][3] * sg, w1);
            const int idx = tid;
            // [k][row] storage: 4 rows c4.. of k = 2 kp + e;   [row][k] storage: row idx / 4, k = 4 (idx % 4) ..
            const size_t off = kmaj ? (size_t)(kp) * SPL_KROW + c4 * 2
                                    : (size_t)(((idx % 4) * 4) >> 3) * SPL_KHALF + (idx / 4) * 16 + (((idx % 4) * 4) & 7) * 2;
#pragma unroll
            for (int p = 0; p < NP; ++p) *(u32x2*)(img[p] + off) = u32x2{w0[p], w1[p]};
        }
    };
    // K-tile t enters with the sign (-1)^t on its A operand and is joined with the same sign: see mfma_tile
    auto sgn = [](int t) { return (TERMS > 1 && (t & 1)) ? -1.0f : 1.0f; };
    auto store_tiles = [&](int t) {
        if (PA) store_pre(sm[t & 1][0], qa, AKM, TERMS > 1 && (t & 1));
        else store_op(sm[t & 1][0], ra, AKM, sgn(t));
        if (PB) store_pre(sm[t & 1][1], qb, BKM, false);
        else store_op(sm[t & 1][1], rb, BKM, 1.0f);
    };
    const int lr = lane & 31, lk = lane >> 5;
    // transposed read: lane 4q + p of a 16-lane group addresses k row q, rows 4p .. 4p + 3 of the group's 16 rows
    // and receives, for row (lane & 15) of the group, the 4 k of the block (cdna_hip_programming.md T10)
    const int tr_off = (8 * lk + ((lane >> 2) & 3)) * SPL_KROW + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    auto frag = [&](const unsigned char* img, int row0, bool kmaj) -> bf16x8 {
        if (kmaj) {
            typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
            const unsigned char* p = img + tr_off + row0 * 2;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)p);
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p + 4 * SPL_KROW));
            return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
        return *(const bf16x8*)(img + (size_t)lk * SPL_KHALF + (row0 + lr) * 16);
    };
#define SPL_MFMA(acc_, a_, b_) acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_, b_, acc_, 0, 0, 0)
    auto mfma_tile = [&](int tile) {
        const int buf = tile & 1;
        const float sg = sgn(tile);
        bf16x8 a[2][NP], b[1][NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
#if defined(G8_ABL) && G8_ABL == 6      /* timing ablation: a third of the fragment reads, no loads, no stores */
            if (p > 0) { a[0][p] = a[0][0]; a[1][p] = a[1][0]; b[0][p] = b[0][0]; continue; }
#endif
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i][p] = frag(sm[buf][0][p], wm * 64 + i * 32, AKM);
            b[0][p] = frag(sm[buf][1][p], wn * 32, BKM);
        }
        if (TERMS == 1) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 1; ++j) SPL_MFMA(acc[i][j], a[i][0], b[j][0]);
        } else {
            // The bf16 MFMA truncates its fp32 accumulation toward -inf (measured: mean error -0.5 ulp of the accumulator
            // per instruction, where v_mfma_f32_32x32x2_f32 rounds to nearest); a bias of one sign survives an LSTM's
            // cell integration far better than rounding noise.  So the TERMS products of a K-tile form a chain from a
            // zero accumulator and join the running sum with a round-to-nearest v_fma_f32, and K-tiles alternate in
            // sign: odd tiles multiply -A (the split of -x is exactly minus the split of x) and are subtracted, so the
            // truncation pushes the sum down on even tiles and up on odd ones and cancels in expectation.  (Accumulating
            // straight into the running sum and flipping ITS sign every 4 tiles instead saves the 64 joins per tile: rms
            // error 1.7e-7 instead of 8.6e-8, bias -3e-9, 2.5 % faster: not taken.)
            constexpr int pa[8] = {0, 0, 1, 1, 0, 2, 1, 2}, pb[8] = {0, 1, 0, 1, 2, 0, 2, 1};
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 1; ++j) {
                    f32x16 tmp;
#pragma unroll
                    for (int e = 0; e < 16; ++e) tmp[e] = 0.0f;
#pragma unroll
#if defined(SPL_ABL) && SPL_ABL == 2
                    for (int t = TERMS - 1; t >= 0; --t) SPL_MFMA(acc[i][j], a[i][pa[t] < NP ? pa[t] : 0], b[j][pb[t] < NP ? pb[t] : 0]);
                    continue;
#endif
#if defined(SPL_ABL) && SPL_ABL == 3
                    for (int t = 2; t >= 0; --t) SPL_MFMA(tmp, a[i][t], b[j][t]);
#else
                    for (int t = TERMS - 1; t >= 0; --t) SPL_MFMA(tmp, a[i][pa[t] < NP ? pa[t] : 0], b[j][pb[t] < NP ? pb[t] : 0]);
#endif
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = __builtin_fmaf(tmp[e], sg, acc[i][j][e]);
                }
        }
    };
#undef SPL_MFMA

    const int kbeg = blockIdx.y * kchunk;
    const int kend = (kbeg + kchunk < K) ? kbeg + kchunk : K;
    const int Ksave = K;
    K = kend;
    const int nk = (kend - kbeg + BK - 1) / BK;
    const int nk_full = (kend - kbeg) / BK;
    const bool interior = vecA && vecB && (m0 + BM <= M) && (n0 + BN <= N) && nk_full > 0;
    if (interior) {
        walk_init(kbeg);
        load_tiles_fast(ra, rb);
        store_tiles(0);
        __syncthreads();
        // one iteration per trip, not unrolled: an unrolled pair made the register allocator move the 64 accumulator
        // registers between the halves (v_mov behind the MFMA latency).  Loads two tiles ahead and an MFMA / split-VALU
        // interleave (sched_group_barrier) were measured and changed nothing: 162 vs 170 TFLOP/s
#pragma unroll 1
        for (int kt = 0; kt + 1 < nk_full; ++kt) {
#if !defined(G8_ABL) || (G8_ABL != 1 && G8_ABL != 4 && G8_ABL != 6)   /* G8_ABL: timing ablations of tools/ubench/gemm_ablate.sh, never a product build */
            load_tiles_fast(ra, rb);
#endif
            __builtin_amdgcn_sched_barrier(0);
#if !defined(G8_ABL) || G8_ABL != 3
            mfma_tile(kt);
#endif
            __builtin_amdgcn_sched_barrier(0);
#if !defined(G8_ABL) || (G8_ABL != 2 && G8_ABL != 4 && G8_ABL != 6)
            store_tiles(kt + 1);
#endif
            __syncthreads();
        }
        if (nk > nk_full) {                                 // K tail: one guarded tile
            load_tiles(kbeg + nk_full * BK);
            mfma_tile(nk_full - 1);
            store_tiles(nk_full);
            __syncthreads();
            mfma_tile(nk_full);
        } else {
            mfma_tile(nk_full - 1);
        }
    } else {
        load_tiles(kbeg);
        store_tiles(0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) load_tiles(kbeg + (kt + 1) * BK);
            mfma_tile(kt);
            if (kt + 1 < nk) store_tiles(kt + 1);
            __syncthreads();
        }
    }
    K = Ksave;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 1; ++j) {
            const int col = n0 + wn * 32 + (lane & 31);
            if (col >= N) continue;
            const float bv = (bias && !part) ? bias[col] : 0.0f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (row >= M) continue;
                const float s = acc[i][j][e];
                if (part) { part[((size_t)blockIdx.y * M + row) * N + col] = s; continue; }
                float* p = C + (size_t)row * ldc + col;
                float v = alpha * s + bv;
                if (beta != 0.0f) v += beta * (*p);
                *p = v;
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
static int gemm_launch(int mode, int transA, int transB, int M, int N, int K, float alpha, const float* A,
                       int lda, const float* B, int ldb, float beta, float* C, int ldc,
                       const float* bias, float* ws, size_t ws_bytes, hipStream_t stream, int preA = 0, int preB = 0,
                       size_t plA = 0, size_t plB = 0) {
    if (M < 0 || N < 0 || K < 0 || !C) return ASR_ERR_ARG;
    if ((preA || preB) && mode != 6) return ASR_ERR_ARG;
    // a pre-split operand: plane 0 and the planes 16-byte aligned, padded rows (asr_split_bf16x3 makes them so)
    if (preA && ((((uintptr_t)A) & 15) || (lda & 7) || (plA & 7))) return ASR_ERR_ARG;
    if (preB && ((((uintptr_t)B) & 15) || (ldb & 7) || (plB & 7))) return ASR_ERR_ARG;
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
    static int wg8 = -1;                  // ASR_GEMM8=0: the four-wave kernel for every product (comparison runs)
    if (wg8 < 0) { const char* e = getenv("ASR_GEMM8"); wg8 = e ? atoi(e) : 1; }
    if (wg8 && !preA && !preB && (mode == 6 || mode == 3 || mode == 1)) {
        dim3 block8(512);
#define SPLIT8_(TA_, TB_, T_)                                                                            \
        hipLaunchKernelGGL((gemm_split8_kernel<TA_, TB_, T_>), grid, block8, 0, stream, M, N, K, alpha, A, lda, B, ldb, \
                           beta, C, ldc, bias, ntn, kchunk, part, plA, plB)
#define LAUNCH8(TA_, TB_)                                                                                \
        do {                                                                                            \
            if (mode == 6) SPLIT8_(TA_, TB_, 6);                                                        \
            else if (mode == 3) SPLIT8_(TA_, TB_, 3);                                                   \
            else SPLIT8_(TA_, TB_, 1);                                                                  \
        } while (0)
        if (!transA && !transB) LAUNCH8(false, false);
        else if (!transA && transB) LAUNCH8(false, true);
        else if (transA && !transB) LAUNCH8(true, false);
        else LAUNCH8(true, true);
#undef LAUNCH8
#undef SPLIT8_
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
#define SPLIT_(TA_, TB_, T_, PA_, PB_)                                                                   \
    hipLaunchKernelGGL((gemm_split_kernel<TA_, TB_, T_, PA_, PB_>), grid, block, 0, stream, M, N, K, alpha, A, lda, B, \
                       ldb, beta, C, ldc, bias, ntn, kchunk, part, plA, plB)
#define LAUNCH(TA_, TB_)                                                                                \
    do {                                                                                                \
        if (mode == 6 && preA && preB) SPLIT_(TA_, TB_, 6, true, true);                                 \
        else if (mode == 6 && preA) SPLIT_(TA_, TB_, 6, true, false);                                   \
        else if (mode == 6 && preB) SPLIT_(TA_, TB_, 6, false, true);                                   \
        else if (mode == 6) SPLIT_(TA_, TB_, 6, false, false);                                          \
        else if (mode == 8) SPLIT_(TA_, TB_, 8, false, false);                                          \
        else if (mode == 3) SPLIT_(TA_, TB_, 3, false, false);                                          \
        else if (mode == 1) SPLIT_(TA_, TB_, 1, false, false);                                          \
        else hipLaunchKernelGGL((gemm_f32_kernel<TA_, TB_>), grid, block, 0, stream, M, N, K, alpha, A, \
                                lda, B, ldb, beta, C, ldc, bias, ntn, kchunk, part);                    \
    } while (0)
    if (!transA && !transB) LAUNCH(false, false);
    else if (!transA && transB) LAUNCH(false, true);
    else if (transA && !transB) LAUNCH(true, false);
    else LAUNCH(true, true);
#undef LAUNCH
#undef SPLIT_
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
    return gemm_launch(0, transA, transB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, ws, ws_bytes, stream);
}
// the optional bf16 mode: fp32 tensors in memory, operands rounded to bf16 on the way into LDS, fp32 accumulate
// (gemm_split_kernel with the leading term only)
extern "C" int asr_gemm_bf16_ws(int transA, int transB, int M, int N, int K, float alpha, const float* A,
                                int lda, const float* B, int ldb, float beta, float* C, int ldc,
                                const float* bias, float* ws, size_t ws_bytes, hipStream_t stream) {
    return gemm_launch(1, transA, transB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, ws, ws_bytes, stream);
}

// fp32-grade products from split bf16 operands (gemm_split_kernel): terms = 8, 6 (3: the two-part reduced-precision
// mode; 1: plain bf16 in the same kernel)
extern "C" int asr_gemm_bf16x3_ws(int terms, int transA, int transB, int M, int N, int K, float alpha, const float* A,
                                  int lda, const float* B, int ldb, float beta, float* C, int ldc,
                                  const float* bias, float* ws, size_t ws_bytes, hipStream_t stream) {
    if (terms != 8 && terms != 6 && terms != 3 && terms != 1) return ASR_ERR_ARG;
    return gemm_launch(terms, transA, transB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, ws,
                       ws_bytes, stream);
}

// ------------------------------------------------------------------------------------------------
// fp32 [rows][cols] (leading dimension ldx) -> three bf16 planes [rows][ldp], ldp = cols rounded up to a multiple of 8,
// the padding columns zero: x = h + m + l with h = rn(x), m = rn(x - h), l = rn(x - h - m) (split_pair above).
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ X, int ldx, int rows, int cols,
                                                     unsigned short* __restrict__ P, int ldp, size_t pl) {
    const int g8 = ldp >> 3;
    const size_t tot = (size_t)rows * g8;
    const bool vec = ((((uintptr_t)X) & 15) == 0) && ((ldx & 3) == 0);
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < tot; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i / g8), c = (int)(i % g8) * 8;
        float v[8];
        const float* src = X + (size_t)r * ldx + c;
        if (vec && c + 7 < cols) {
            const f32x4 a = *(const f32x4*)src, b = *(const f32x4*)(src + 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[k] = a[k]; v[4 + k] = b[k]; }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = c + k < cols ? src[k] : 0.f;
        }
        unsigned w[4][3];
#pragma unroll
        for (int k = 0; k < 4; ++k) split_pair<3>(v[2 * k], v[2 * k + 1], w[k]);
#pragma unroll
        for (int p = 0; p < 3; ++p)
            *(u32x4s*)(P + (size_t)p * pl + (size_t)r * ldp + c) = u32x4s{w[0][p], w[1][p], w[2][p], w[3][p]};
    }
}

extern "C" int asr_split_ld(int cols) { return (cols + 7) & ~7; }

extern "C" int asr_split_bf16x3(const float* X, int ldx, int rows, int cols, void* planes, int ldp,
                                size_t plane_stride, hipStream_t stream) {
    if (rows < 0 || cols < 0 || !planes || ldp < cols || (ldp & 7) || (plane_stride & 7) ||
        plane_stride < (size_t)rows * ldp || (((uintptr_t)planes) & 15) || (rows && cols && !X))
        return ASR_ERR_ARG;
    if (!rows || !cols) return ASR_OK;
    const size_t tot = (size_t)rows * (ldp >> 3);
    const int grid = (int)((tot + 255) / 256 < 16384 ? (tot + 255) / 256 : 16384);
    hipLaunchKernelGGL(split3_kernel, dim3(grid), dim3(256), 0, stream, X, ldx, rows, cols, (unsigned short*)planes,
                       ldp, plane_stride);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}

// asr_gemm_bf16x3_ws (6 products) with either operand, or both, given as the planes asr_split_bf16x3 wrote:
// preA / preB != 0: A / B points at plane 0 of op's STORED matrix (same shape and transposition flags as the fp32
// form), lda / ldb = ldp, planeA / planeB = plane_stride.  Bit-identical to the fp32-operand call.
extern "C" int asr_gemm_presplit_ws(int preA, int preB, int transA, int transB, int M, int N, int K, float alpha,
                                    const void* A, int lda, size_t planeA, const void* B, int ldb, size_t planeB,
                                    float beta, float* C, int ldc, const float* bias, float* ws, size_t ws_bytes,
                                    hipStream_t stream) {
    return gemm_launch(6, transA, transB, M, N, K, alpha, (const float*)A, lda, (const float*)B, ldb, beta, C, ldc,
                       bias, ws, ws_bytes, stream, preA, preB, planeA, planeB);
}

extern "C" int asr_gemm_f32(int transA, int transB, int M, int N, int K, float alpha, const float* A,
                            int lda, const float* B, int ldb, float beta, float* C, int ldc,
                            const float* bias, hipStream_t stream) {
    return asr_gemm_f32_ws(transA, transB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, nullptr, 0,
                           stream);
}
