// K2 (HyperLSTM): persistent recurrent scan, forward and BPTT, for
//   libs/lasagne_libs/hyper_lhuc_layers.py:216-237 (eq.10 hyper cell), :363-412 (step),
//   :414-425 (step_masked), :493-500 (theano.scan over [orig_input, input, mask]).
//
// Host-side preparation (asr_amd/engine_hyper.py):
//   * xproj  = x.W_x (T,B,4N)                                   (:469)
//   * xhproj = x.W_xhat[N:] + bhat (T,B,4H): the x_t half of eq.10's [h_{t-1}, x_t].W_xhat hoisted
//     out of the loop ([h,x].W = h.W[:N] + x.W[N:], exact algebra)
//   * eq.11 and eq.12 folded: d = (hhat.W_hhat + b).W_z = hhat.(W_hhat.W_z) + b.W_z, so the step
//     goes hhat_t -> d_h, d_x, bias in ONE product with M (H x 12N) (+ constant row), removing the
//     z exchange; the factors' gradients are recovered from dM after the scan.
// Group = G = N/4 workgroups; workgroup j owns main units 4j..4j+3 and hyper units j*UH..j*UH+UH-1.
// Per step two exchanges: hhat_t (after the hyper cell) and h_t (after the main cell).
#include "scan_common.h"
#include <stdlib.h>

struct HyperDir {
    int reverse;
    const float* xproj;      // (T,B,4N)
    const float* xhproj;     // (T,B,4H)
    const float* wpack;      // asr_hyper_pack_w (mode 0 fwd / 1 bwd)
    const float* cvec;       // (12N) = row H of the folded matrix: [c_h | c_x | c_b], interleaved
    float* r_out; int ld_r;  // (T,B,ld_r) h_t
    float* hh_out;           // (T,B,H)  hhat_t
    float* gates;            // (T,B,4N) post-activation
    float* cst;              // (T,B,N)
    float* hgates;           // (T,B,4H)
    float* hcst;             // (T,B,H)
    float* aD;               // (T,B,3,4N): a = h_{t-1}.W_h, d_h, d_x
    const float* dr_ext; int ld_dr;
    float* dXp;              // (T,B,4N) dgates * d_x           -> dW_x, dx
    float* dXh;              // (T,B,4H) clipped hyper gate grads -> dW_xhat, dbhat, dW_hhat, dx
    float* dA;               // (T,B,4N) dgates * d_h           -> dW_h
    float* dD;               // (T,B,3,4N) [dgates*a, dgates*xproj, dgates] -> dM, dcvec
    float* dcv;              // (ceil(B/8),12N) or null: column sums of dD over time and batch, per batch chunk
    float* dbh;              // (ceil(B/8),4H)  or null: column sums of dXh
    float* carry;            // (B, 2N+2H) or null: BPTT state handed from one time chunk's launch to the next
};

struct HyperArgs {
    int T, B, N, H, G, UH, KS, KSP, KT, HS, HSP, HT;
    int s0, s1;                  // steps [s0, s1) of the T-step recurrence run by this launch (in processing order)
    int R, MT, KQ, NBLK, NBW;    // backward W_h product: see hyper_dims
    int ndir, chunk0, rpc;       // rpc = batch rows per chunk/group (8 or 16)
    float clip;
    float clip_out;              // bound of the extra grad_clip on outcell (ScalingHyperLSTMLayer), 0 = none
    int lhuc;                    // HyperLHUCLSTMLayer: 0 off, else 1 + reparam (relu, exp, identity, 2sigmoid)
    const float* mask; int msb, mst;
    HyperDir d[2];
    unsigned* sync;
    float* rex;      // [ng][2][KSP workgroups][16 b][4 units]   h_t   (all-gather)
    float* hex;      // [ng][2][HSP groups of 4 hyper units][16 b][4]  hhat_t (all-gather)
    float* dax;      // [ng][2][G src][64 lanes][4]: da of the source's 16 gate columns (all-gather)
    float* px;       // [ng][2][G src][4 waves][16 units][16 b]: partial W_h.da tiles (R-way reduce)
    float* dhx;      // [ng][2][G dst][G src][UH*16]
    float* dgx;      // [ng][2][H][16][4]
    unsigned long long* dbg;   // STAMP_BYTES at the workspace tail (diagnostic builds)
};

static inline void hyper_dims(int N, int H, HyperArgs* A) {
    A->N = N; A->H = H;
    A->G = N / 4;
    A->KS = N / 4; A->KSP = ((A->KS + 15) / 16) * 16; A->KT = (N + 15) / 16;
    A->HS = H / 4; A->HSP = ((A->HS + 15) / 16) * 16; A->HT = (H + 15) / 16;
    int uh = 1;
    while (uh * A->G < H) uh *= 2;
    A->UH = uh;
    // Backward dh_{t-1} = W_h . da as a 2-D decomposition: R consecutive workgroups form a unit block
    // (4R units); member kr multiplies the block's rows with ITS 1/R of the gate columns (all-gathered
    // da of NBLK source workgroups) and the R partial tiles are summed by the unit owners.  R = the
    // largest divisor of G that is <= 8 (8 = one K range per XCD under round-robin dispatch).
    int r = 1;
    for (int c = 2; c <= 8; ++c) if (A->G % c == 0) r = c;
    A->R = r;
    A->MT = (4 * r + 15) / 16;              // 16-unit tiles per block (1 or 2)
    A->KQ = 4 / A->MT;                      // K slices per workgroup = waves per tile
    A->NBLK = A->G / r;                     // source workgroups (16-column blocks) per K range
    A->NBW = (A->NBLK + A->KQ - 1) / A->KQ; // blocks per wave
}
static inline size_t hyper_pack_floats_wg(const HyperArgs& A, int mode) {
    const int CR = 4 * A.UH;
    const int S4P = ((A.H + 15) / 16) * 16;
    if (mode == 0) return (size_t)A.KSP * 64 + (size_t)(A.KSP + A.HSP) * 4 * CR + (size_t)3 * A.HSP * 64;
    return (size_t)A.NBW * 1024 + (size_t)A.HT * 3 * 256 + (size_t)S4P * 4 * (4 + A.UH);
}

// ------------------------------------------------------------------------------------------------
// packing: one thread block per workgroup slice
__global__ void hyper_pack_kernel(const float* __restrict__ Wh, const float* __restrict__ Whx,
                                  const float* __restrict__ Whh, const float* __restrict__ M, HyperArgs A,
                                  int mode, size_t per_wg, float* __restrict__ out) {
    const int j = blockIdx.x;
    const int N = A.N, H = A.H, UH = A.UH, CR = 4 * UH;
    float* o = out + (size_t)j * per_wg;
    if (mode == 0) {
        const size_t n1 = (size_t)A.KSP * 64;
        for (size_t i = threadIdx.x; i < n1; i += blockDim.x) {
            const int e = i & 3, l = (i >> 2) & 63, s4 = (int)(i >> 8);
            const int row = 4 * (4 * s4 + (l >> 4)) + e, col = j * 16 + (l & 15);   // k = source workgroup, step e = its unit
            o[i] = row < N ? Wh[(size_t)row * 4 * N + col] : 0.f;
        }
        float* og = o + n1;
        const size_t n2 = (size_t)(A.KSP + A.HSP) * 4 * CR;
        for (size_t i = threadIdx.x; i < n2; i += blockDim.x) {
            const int e = i & 3;
            const int ic = (int)((i >> 2) % CR);
            const int qq = (int)((i >> 2) / CR) & 3;
            const int s4g = (int)((i >> 2) / CR) >> 2;           // over KSP/4 + HSP/4 groups
            const int hu = j * UH + ic / 4;
            const int col = hu * 4 + (ic & 3);
            float v = 0.f;
            if (hu < H) {
                if (s4g < A.KSP / 4) {
                    const int row = 4 * (4 * s4g + qq) + e;
                    if (row < N) v = Whx[(size_t)row * 4 * H + col];
                } else {
                    const int row = 4 * (4 * (s4g - A.KSP / 4) + qq) + e;
                    if (row < H) v = Whh[(size_t)row * 4 * H + col];
                }
            }
            og[i] = v;
        }
        float* od = og + n2;
        const size_t n3 = (size_t)3 * A.HSP * 64;
        for (size_t i = threadIdx.x; i < n3; i += blockDim.x) {
            const int e = i & 3, l = (i >> 2) & 63;
            const int s4 = (int)((i >> 8) % (A.HSP / 4)), tau = (int)((i >> 8) / (A.HSP / 4));
            const int row = 4 * (4 * s4 + (l >> 4)) + e;
            od[i] = row < H ? M[(size_t)row * 12 * N + (size_t)tau * 4 * N + j * 16 + (l & 15)] : 0.f;
        }
    } else {
        // W_h rows of the unit block against the own K range, as B operands (k = column, n = unit) of
        // 16x16x4 MFMAs: [wave][block of the wave][lane][e], wave = kq*MT + mt
        const size_t n1 = (size_t)A.NBW * 1024;
        const int ub = j / A.R, kr = j % A.R;
        for (size_t i = threadIdx.x; i < n1; i += blockDim.x) {
            const int e = i & 3, l = (i >> 2) & 63, ib = (int)((i >> 8) % A.NBW), w = (int)((i >> 8) / A.NBW);
            const int mt = w % A.MT, kq = w / A.MT;
            const int bi = kq * A.NBW + ib;                       // block inside the K range
            const int ul = mt * 16 + (l & 15);                    // unit inside the unit block
            const int row = ub * 4 * A.R + ul, col = (kr * A.NBLK + bi) * 16 + 4 * e + (l >> 4);
            o[i] = (bi < A.NBLK && ul < 4 * A.R) ? Wh[(size_t)row * 4 * N + col] : 0.f;
        }
        float* om = o + n1;
        const size_t n2 = (size_t)A.HT * 3 * 256;
        for (size_t i = threadIdx.x; i < n2; i += blockDim.x) {
            const int e = i & 3, l = (i >> 2) & 63;
            const int tau = (int)((i >> 8) % 3), ht = (int)((i >> 8) / 3);
            const int hu = ht * 16 + (l & 15);
            om[i] = hu < H ? M[(size_t)hu * 12 * N + (size_t)tau * 4 * N + j * 16 + 4 * e + (l >> 4)] : 0.f;
        }
        float* orw = om + n2;
        const int S4P = ((H + 15) / 16) * 16;
        const int RR = 4 + UH;
        const size_t n3 = (size_t)S4P * 4 * RR;
        for (size_t i = threadIdx.x; i < n3; i += blockDim.x) {
            const int e = i & 3, ir = (int)((i >> 2) % RR), qq = (int)(((i >> 2) / RR) & 3);
            const int s4 = (int)(((i >> 2) / RR) >> 2);
            const int col = 4 * (4 * s4 + qq) + e;              // hyper unit 4*s4+qq (MFMA k), gate e (MFMA step)
            float v = 0.f;
            if (col < 4 * H) {
                if (ir < 4) v = Whx[(size_t)(j * 4 + ir) * 4 * H + col];
                else if (ir - 4 < UH && j * UH + ir - 4 < H) v = Whh[(size_t)(j * UH + ir - 4) * 4 * H + col];
            }
            orw[i] = v;
        }
    }
}

extern "C" size_t asr_hyper_pack_floats(int N, int H, int mode) {
    HyperArgs A;
    hyper_dims(N, H, &A);
    return hyper_pack_floats_wg(A, mode) * A.G;
}

// Wh (N,4N), Whx (N,4H) = rows [0,N) of W_xhat, Whh (H,4H), M (H,12N): all gate-interleaved.
extern "C" int asr_hyper_pack_w(const float* Wh, const float* Whx, const float* Whh, const float* M, int N,
                                int H, int mode, float* out, hipStream_t stream) {
    if (!Wh || !Whx || !Whh || !M || !out || N <= 0 || H <= 0 || (N & 3) || (H & 3) || H > N) return ASR_ERR_ARG;
    HyperArgs A;
    hyper_dims(N, H, &A);
    if (A.UH > 4) return ASR_ERR_ARG;
    hipLaunchKernelGGL(hyper_pack_kernel, dim3(A.G), dim3(256), 0, stream, Wh, Whx, Whh, M, A, mode,
                       hyper_pack_floats_wg(A, mode), out);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}

// reparam_* of libs/lasagne_libs/hyper_lhuc_layers.py:17-27 and their derivatives (kind = HyperArgs.lhuc)
__device__ __forceinline__ float lhuc_rp(int kind, float v) {
    return kind == 1 ? fmaxf(v, 0.f) : kind == 2 ? expf(v) : kind == 3 ? v : 2.f * sigmoidf_(v);
}
__device__ __forceinline__ float lhuc_drp(int kind, float v) {
    if (kind == 1) return v > 0.f ? 1.f : 0.f;
    if (kind == 2) return expf(v);
    if (kind == 3) return 1.f;
    const float sg = sigmoidf_(v);
    return 2.f * sg * (1.f - sg);
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// NC = 16-row batch chunks carried by one workgroup (1 or 2; the code is written for 4 as well, see hyper_launch): a workgroup is NC teams of 4 waves that run the
// SAME program on consecutive chunks against the SAME resident weights (one copy in LDS) and share every
// publish / wait of the step (one arrival per workgroup and exchange, payload NC times as large).  The matrix pipe
// is 13-19 % busy and arithmetic ~ waits at NC = 1, so a second team's MFMAs and LDS / L2
// latencies fall into the first team's waits.  Per-row results are bit-identical to NC = 1: a team's arithmetic
// does not depend on NC.
template <int NC>
__global__ __launch_bounds__(256 * NC) void hyper_scan_fwd_kernel(HyperArgs A, size_t per_wg) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __builtin_amdgcn_s_setprio(3);       // latency-bound chain: win issue arbitration against co-resident GEMM waves
    const int N = A.N, H = A.H, UH = A.UH, CR = 4 * UH, KS = A.KS, KSP = A.KSP, HS = A.HS, HSP = A.HSP;
    const int T = A.T, B = A.B, G = A.G;
    float* Wm = smem;                                           // [KSP/4][64][4]
    float* Wg = Wm + (size_t)KSP * 64;                          // [(KSP+HSP)/4][4][CR][4]
    float* Wd = Wg + (size_t)(KSP + HSP) * 4 * CR;              // [3][HSP/4][64][4]
    // wave-uniform, and the compiler must KNOW it: the exchange buffers' descriptors are built from pointers that depend
    // on it, and a descriptor from a value taken to be divergent costs a waterfall loop (4 v_readfirstlane, 2 v_cmp, a
    // saveexec and a branch) around every buffer load and store of the step
    const int team = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
    float* red = Wd + (size_t)3 * HSP * 64 + (size_t)team * 3 * 4 * 256;   // per team [3 tiles][4 waves][64][4]
    int* flag = (int*)(Wd + (size_t)3 * HSP * 64 + (size_t)NC * 3 * 4 * 256);

    const int j = blockIdx.x, dir = blockIdx.y, chunk = blockIdx.z * NC + team + A.chunk0;
    const int grp = chunk * A.ndir + dir;                       // exchange buffers: one set per chunk and direction
    const int sgrp = (blockIdx.z * NC + A.chunk0) * A.ndir + dir;   // arrival counters: one set per workgroup set
    const HyperDir& D = A.d[dir];
    // Roles (K quarter, epilogue duties) are rotated per team so that the teams' epilogue waves (role 0: main cell,
    // role 1: hyper cell) sit on different SIMDs: hardware wave w runs on SIMD w % 4
    const int lane = threadIdx.x & 63, wave = (int)(((threadIdx.x >> 6) + team * (4 / NC)) & 3);
    const int tid = wave * 64 + lane;
    const int q = lane >> 4, bl = lane & 15;
    const int b = chunk * A.rpc + bl;
    const bool bvalid = (bl < A.rpc) && (b < B);

    GroupSync gs;
    group_sync_init(gs, A.sync + (size_t)sgrp * SYNC_STRIDE, G, j, flag);
    {
        const f32x4* src = (const f32x4*)(D.wpack + (size_t)j * per_wg);
        f32x4* dst = (f32x4*)smem;
        for (size_t i = threadIdx.x; i < per_wg / 4; i += 256 * NC) dst[i] = src[i];
    }
    float* rex = A.rex + (size_t)grp * 2 * KSP * 64;
    float* hex = A.hex + (size_t)grp * 2 * HSP * 64;
    const __amdgpu_buffer_rsrc_t rex_rs = make_rsrc_(rex, (unsigned)(2 * KSP * 64 * 4));
    const __amdgpu_buffer_rsrc_t hex_rs = make_rsrc_(hex, (unsigned)(2 * HSP * 64 * 4));
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    // Exchange layouts [block of 4 units][batch][unit in block]: MFMA k index q <-> block 4*s4+q, the 4
    // MFMA steps of a group <-> the 4 units of the block, so a lane's B operands of one group are ONE
    // 16-byte sc1 load (16-B accesses run at ~2x the per-byte rate of 4-B ones on this path)
    const unsigned ex_voff = (unsigned)((q * 64 + bl * 4) * 4);

    const int unit = j * 4 + q;                  // wave 0: main unit of this lane
    const int col4 = unit * 4;
    const int hu = j * UH + q;                   // wave 1: hyper unit of this lane (valid if q < UH)
    const bool hvalid = (q < UH) && (hu < H);
    float c = 0.f, r = 0.f, hc = 0.f, hh = 0.f;
    f32x4 cv_h = {0, 0, 0, 0}, cv_x = {0, 0, 0, 0}, cv_b = {0, 0, 0, 0};
    if (wave == 0) {
        cv_h = *(const f32x4*)(D.cvec + col4);
        cv_x = *(const f32x4*)(D.cvec + 4 * N + col4);
        cv_b = *(const f32x4*)(D.cvec + 8 * N + col4);
    }
    const int s0 = A.s0, s1 = A.s1;
    // zero both parities of both exchange buffers, padding included (state 0 at step 0; padded blocks
    // meet zero weights, but 0 x stale NaN would still poison the MFMA): the group's G workgroups
    // share the work; the first wait below orders it before any reader and any later writer
    for (int i = j * 256 + tid; i < 2 * KSP * 16; i += G * 256) st_sc1_f4(rex_rs, (unsigned)i * 16u, f32x4{0, 0, 0, 0});
    for (int i = j * 256 + tid; i < 2 * HSP * 16; i += G * 256) st_sc1_f4(hex_rs, (unsigned)i * 16u, f32x4{0, 0, 0, 0});
    group_publish(gs);
    if (s0 > 0) {
        // a later time chunk: the carried state is what the previous launch saved for step s0 - 1 (the
        // saved rows hold the masked, i.e. carried, values); every workgroup re-publishes its part of
        // h_{s0-1}, hhat_{s0-1} into the parity step s0 reads, behind the zeroing above
        if (!group_wait(gs)) return;
        const int tp = D.reverse ? T - s0 : s0 - 1;
        const size_t rowp = (size_t)tp * B + b;
        const int par0 = s0 & 1;
        if (wave == 0) {
            if (bvalid) { c = D.cst[rowp * N + unit]; r = D.r_out[rowp * D.ld_r + unit]; }
            const float r1 = __shfl(r, bl + 16), r2 = __shfl(r, bl + 32), r3 = __shfl(r, bl + 48);
            if (q == 0) st_sc1_f4(rex_rs, (unsigned)((((size_t)par0 * KSP + j) * 64 + bl * 4) * 4), f32x4{r, r1, r2, r3});
        }
        if (wave == 1 && hvalid) {
            if (bvalid) { hc = D.hcst[rowp * H + hu]; hh = D.hh_out[rowp * H + hu]; }
            st_sc1(hex + ((size_t)par0 * HSP * 64) + ((size_t)(hu >> 2) * 16 + bl) * 4 + (hu & 3), hh);
        }
        group_publish(gs);
    }

    const int ks4 = KSP / 16, hs4 = HSP / 16;    // b128 groups per wave
    // exchange-buffer operands requested together before the first MFMA that needs them: 8 blocks of h (4 with
    // four teams per workgroup: 128 registers per wave there)
    constexpr int BVC = NC == 4 ? 4 : 8;
    // inputs of the step, loaded one step ahead through walking per-lane pointers; branch-free (every
    // wave issues every load, non-owners read a dummy line) so that no join sits behind the loads
    const bool w0 = (wv == 0), w1 = (wv == 1);
    const int bq = bvalid ? b : (B - 1), huq = hvalid ? hu : 0;
    const float* const dummy = D.wpack;
    const int dt = D.reverse ? -1 : 1;
    const int t00 = D.reverse ? T - 1 - s0 : s0;              // time index of the first step of this launch
    const size_t r00 = (size_t)t00 * B + bq;
    const float* p_m = (w0 || w1) ? A.mask + (size_t)t00 * A.mst + (size_t)bq * A.msb : dummy;
    const float* p_x = w0 ? D.xproj + r00 * 4 * N + col4 : dummy;
    const float* p_xh = w1 ? D.xhproj + r00 * 4 * H + huq * 4 : dummy;
    const int s_m = (w0 || w1) ? dt * A.mst : 0, s_x = w0 ? dt * B * 4 * N : 0, s_xh = w1 ? dt * B * 4 * H : 0;
    float n_m = *p_m;
    f32x4 n_x4 = *(const f32x4*)p_x, n_xh4 = *(const f32x4*)p_xh;
    STAMP_DECL
    for (int step = s0; step < s1; ++step) {
        const int t = D.reverse ? T - 1 - step : step;
        const int par = step & 1;
        const size_t row = (size_t)t * B + b;
        const float m = bvalid ? n_m : 0.f;
        const f32x4 x4 = n_x4, xh4 = n_xh4;
        {
            const bool go = step + 1 < s1;
            p_m += go ? s_m : 0; p_x += go ? s_x : 0; p_xh += go ? s_xh : 0;
            n_m = *p_m; n_x4 = *(const f32x4*)p_x; n_xh4 = *(const f32x4*)p_xh;
        }
        if (!group_wait(gs)) break;
        STAMP(0);

        // ---- phase 1: hyper gate pre-activations first (they gate the first exchange); the main
        // tile a = h.W_h is not needed before phase 2, so when the wave's whole K share of h fits one
        // register chunk (N <= 512) its MFMAs are deferred until after the hhat publish and run
        // under the exchange latency.  All B operands of a chunk are requested before the first
        // MFMA so the exchange-buffer latency is paid once per chunk, not once per k-step.
        f32x4 am0 = {0, 0, 0, 0}, am1 = {0, 0, 0, 0}, ag0 = {0, 0, 0, 0}, ag1 = {0, 0, 0, 0};
        const int ic = (bl < CR ? bl : CR - 1);
        const bool defer = (ks4 <= BVC);
        f32x4 bh[4], bv[BVC];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            bh[i] = ldb_sc1_f4(hex_rs, ex_voff, (unsigned)(par * HSP * 256) + (unsigned)(wv * hs4 + i) * 1024u);
#pragma unroll
        for (int i = 0; i < BVC; ++i)
            bv[i] = ldb_sc1_f4(rex_rs, ex_voff, (unsigned)(par * KSP * 256) + (unsigned)(wv * ks4 + i) * 1024u);
#pragma unroll
        for (int g4 = 0; g4 < BVC; ++g4) {
            if (g4 < ks4) {
                const int s4 = wave * ks4 + g4;
                const f32x4 gv = *(const f32x4*)(Wg + (((size_t)s4 * 4 + q) * CR + ic) * 4);
                ag0 = mfma4(gv[0], bv[g4][0], ag0); ag1 = mfma4(gv[1], bv[g4][1], ag1);
                ag0 = mfma4(gv[2], bv[g4][2], ag0); ag1 = mfma4(gv[3], bv[g4][3], ag1);
                if (!defer) {
                    const f32x4 av = *(const f32x4*)(Wm + ((size_t)s4 * 64 + lane) * 4);
                    am0 = mfma4(av[0], bv[g4][0], am0); am1 = mfma4(av[1], bv[g4][1], am1);
                    am0 = mfma4(av[2], bv[g4][2], am0); am1 = mfma4(av[3], bv[g4][3], am1);
                }
            }
        }
        for (int c0 = BVC; c0 < ks4; c0 += BVC) {       // N > 512: further chunks, main + hyper together
            f32x4 bw[BVC];
#pragma unroll
            for (int i = 0; i < BVC; ++i)
                bw[i] = ldb_sc1_f4(rex_rs, ex_voff, (unsigned)(par * KSP * 256) + (unsigned)(wv * ks4 + c0 + i) * 1024u);
#pragma unroll
            for (int g4 = 0; g4 < BVC; ++g4) {
                if (c0 + g4 < ks4) {
                    const int s4 = wave * ks4 + c0 + g4;
                    const f32x4 av = *(const f32x4*)(Wm + ((size_t)s4 * 64 + lane) * 4);
                    const f32x4 gv = *(const f32x4*)(Wg + (((size_t)s4 * 4 + q) * CR + ic) * 4);
                    am0 = mfma4(av[0], bw[g4][0], am0); ag0 = mfma4(gv[0], bw[g4][0], ag0);
                    am1 = mfma4(av[1], bw[g4][1], am1); ag1 = mfma4(gv[1], bw[g4][1], ag1);
                    am0 = mfma4(av[2], bw[g4][2], am0); ag0 = mfma4(gv[2], bw[g4][2], ag0);
                    am1 = mfma4(av[3], bw[g4][3], am1); ag1 = mfma4(gv[3], bw[g4][3], ag1);
                }
            }
        }
        for (int h0 = 0; h0 < hs4; h0 += 4) {
            if (h0 > 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    bh[i] = ldb_sc1_f4(hex_rs, ex_voff, (unsigned)(par * HSP * 256) + (unsigned)(wv * hs4 + h0 + i) * 1024u);
            }
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                if (h0 + g4 < hs4) {
                    const int s4 = wave * hs4 + h0 + g4;
                    const f32x4 gv = *(const f32x4*)(Wg + (((size_t)(KSP / 4 + s4) * 4 + q) * CR + ic) * 4);
                    ag0 = mfma4(gv[0], bh[g4][0], ag0); ag1 = mfma4(gv[1], bh[g4][1], ag1);
                    ag0 = mfma4(gv[2], bh[g4][2], ag0); ag1 = mfma4(gv[3], bh[g4][3], ag1);
                }
            }
        }
        if (!defer) *(f32x4*)(red + ((size_t)(0 * 4 + wave) * 64 + lane) * 4) = am0 + am1;
        *(f32x4*)(red + ((size_t)(1 * 4 + wave) * 64 + lane) * 4) = ag0 + ag1;
        STAMP(1);
        __syncthreads();
        f32x4 a4 = {0, 0, 0, 0};
        f32x4 hsave = {0, 0, 0, 0};
        if (wave == 0 && !defer) {
#pragma unroll
            for (int w = 0; w < 4; ++w) a4 += *(const f32x4*)(red + ((size_t)(0 * 4 + w) * 64 + lane) * 4);
        }
        if (wave == 1) {          // hyper cell, eq.10 (:226-235)
            f32x4 hg = xh4;
#pragma unroll
            for (int w = 0; w < 4; ++w) hg += *(const f32x4*)(red + ((size_t)(1 * 4 + w) * 64 + lane) * 4);
            const float hi = sigmoidf_(hg[0]), hf = sigmoidf_(hg[1]), hgc = tanhf_(hg[2]), ho = sigmoidf_(hg[3]);
            const float hc_new = hf * hc + hi * hgc;
            const float hh_new = ho * tanhf_(hc_new);
            if (m != 0.f && hvalid) { hc = hc_new; hh = hh_new; }
            if (hvalid) st_sc1(hex + ((size_t)(par ^ 1) * HSP * 64) + ((size_t)(hu >> 2) * 16 + bl) * 4 + (hu & 3), hh);
            hsave = f32x4{hi, hf, hgc, ho};
        }
        STAMP(2);
        group_publish(gs);
        STAMP(3);
        if (defer) {                              // main tile under the exchange latency
#pragma unroll
            for (int g4 = 0; g4 < BVC; ++g4) {
                if (g4 < ks4) {
                    const f32x4 av = *(const f32x4*)(Wm + ((size_t)(wave * ks4 + g4) * 64 + lane) * 4);
                    am0 = mfma4(av[0], bv[g4][0], am0); am1 = mfma4(av[1], bv[g4][1], am1);
                    am0 = mfma4(av[2], bv[g4][2], am0); am1 = mfma4(av[3], bv[g4][3], am1);
                }
            }
            // slot 0 is free here: its last reader (wave 0, previous step's main epilogue) has passed
            // the publish barrier above
            *(f32x4*)(red + ((size_t)(0 * 4 + wave) * 64 + lane) * 4) = am0 + am1;
        }
        if (wave == 1 && bvalid && hvalid) {      // bulk saves after the publish: off the critical path
            D.hh_out[row * H + hu] = hh;
            if (D.hgates) {
                *(f32x4*)(D.hgates + row * 4 * H + hu * 4) = hsave;
                D.hcst[row * H + hu] = hc;
            }
        }
        if (!group_wait(gs)) break;
        STAMP(4);
        if (wave == 0 && defer) {
#pragma unroll
            for (int w = 0; w < 4; ++w) a4 += *(const f32x4*)(red + ((size_t)(0 * 4 + w) * 64 + lane) * 4);
        }

        // ---- phase 2: d_h, d_x, bias = hhat_t . M  (3 tiles), then the main cell
        f32x4 ad[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        for (int h0 = 0; h0 < hs4; h0 += 4) {
            f32x4 bh[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                bh[i] = ldb_sc1_f4(hex_rs, ex_voff, (unsigned)((par ^ 1) * HSP * 256) + (unsigned)(wv * hs4 + h0 + i) * 1024u);
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                if (h0 + g4 < hs4) {
                    const int s4 = wave * hs4 + h0 + g4;
#pragma unroll
                    for (int tau = 0; tau < 3; ++tau) {
                        const f32x4 dv = *(const f32x4*)(Wd + (((size_t)tau * (HSP / 4) + s4) * 64 + lane) * 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) ad[tau] = mfma4(dv[e], bh[g4][e], ad[tau]);
                    }
                }
            }
        }
        __syncthreads();                          // wave 0 has taken a4 out of slot 0
#pragma unroll
        for (int tau = 0; tau < 3; ++tau) *(f32x4*)(red + ((size_t)(tau * 4 + wave) * 64 + lane) * 4) = ad[tau];
        STAMP(5);
        __syncthreads();
        f32x4 dh4 = cv_h, dx4 = cv_x, bb4 = cv_b, gsave = {0, 0, 0, 0};
        if (wave == 0) {
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                dh4 += *(const f32x4*)(red + ((size_t)(0 * 4 + w) * 64 + lane) * 4);
                dx4 += *(const f32x4*)(red + ((size_t)(1 * 4 + w) * 64 + lane) * 4);
                bb4 += *(const f32x4*)(red + ((size_t)(2 * 4 + w) * 64 + lane) * 4);
            }
            // HyperLHUC: M carries d_h = d_x = 1 and, in gate column 0 of the bias tile, the LHUC
            // pre-scale d = (hhat.W_hhat_h + b).W_hz of the unit; the gates are the plain h.W_h + xproj
            const f32x4 pre = A.lhuc ? a4 * dh4 + x4 * dx4 : a4 * dh4 + (x4 * dx4 + bb4);       // :393-395 | :593
            const float ig = sigmoidf_(pre[0]), fg = sigmoidf_(pre[1]), gg = tanhf_(pre[2]), og = sigmoidf_(pre[3]);
            const float c_new = fg * c + ig * gg;
            float r_new = og * tanhf_(c_new);
            if (A.lhuc) r_new *= lhuc_rp(A.lhuc, bb4[0]);        // hyper_lhuc_layers.py:612
            if (m != 0.f) { c = c_new; r = r_new; }
            {
                const float r1 = __shfl(r, bl + 16), r2 = __shfl(r, bl + 32), r3 = __shfl(r, bl + 48);
                if (q == 0)
                    st_sc1_f4(rex_rs, (unsigned)((((size_t)(par ^ 1) * KSP + j) * 64 + bl * 4) * 4), f32x4{r, r1, r2, r3});
            }
            gsave = f32x4{ig, fg, gg, og};
        }
        STAMP(6);
        group_publish(gs);
        STAMP(7);
        if (wave == 0 && bvalid) {
            D.r_out[row * D.ld_r + unit] = r;
            if (D.gates) {
                D.cst[row * N + unit] = c;
                *(f32x4*)(D.gates + row * 4 * N + col4) = gsave;
                float* ad_ = D.aD + row * 12 * N;
                *(f32x4*)(ad_ + col4) = a4;
                *(f32x4*)(ad_ + 4 * N + col4) = A.lhuc ? bb4 : dh4;   // LHUC: d_h == 1, keep the pre-scale
                *(f32x4*)(ad_ + 8 * N + col4) = dx4;
            }
        }
    }
    STAMP_DUMP(A.dbg, (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
#define HYPER_BWD_STAGE (256 + 768 + 1024 + 256 + 1024)   // floats of per-team LDS staging in the backward kernel
#define DAB 8      // da blocks per wave put in flight early (covers NBW <= 8, e.g. G = 128 with R = 8)
template <int NC>
__global__ __launch_bounds__(256 * NC) void hyper_scan_bwd_kernel(HyperArgs A, size_t per_wg) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __builtin_amdgcn_s_setprio(3);
    const int N = A.N, H = A.H, UH = A.UH, HT = A.HT;
    const int T = A.T, B = A.B, G = A.G;
    const int S4P = ((H + 15) / 16) * 16;
    const int UHB = UH * 16;
    constexpr int DABN = NC == 4 ? 4 : DAB;     // operand blocks in flight per wave (128 registers per wave at NC = 4)
    float* Wt = smem;                                   // [4 waves][NBW][64][4]
    float* Mb = Wt + (size_t)A.NBW * 1024;              // [HT][3][64][4]
    float* Rw = Mb + (size_t)HT * 3 * 256;              // [S4P/4][4][4+UH][4]
    const int team = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);   // which of the NC chunks this 4-wave team carries (see forward)
    float* stage = Rw + (size_t)S4P * 4 * (4 + UH);     // per team: HYPER_BWD_STAGE floats
    float* dal = stage + (size_t)team * HYPER_BWD_STAGE;          // [16 cols][16 b]            da
    float* dDl = dal + 256;                             // [3][16 cols][16 b]         dD
    float* rsum = dDl + 768;                            // [16 pieces][4 units][16 b] W_h . da partial gather
    float* hsum = rsum + 1024;                          // [256]                      dhx gather partials
    float* red = hsum + 256;                            // [4 waves][64][4]           Rw product
    int* flag = (int*)(stage + (size_t)NC * HYPER_BWD_STAGE);

    const int j = blockIdx.x, dir = blockIdx.y, chunk = blockIdx.z * NC + team + A.chunk0;
    const int grp = chunk * A.ndir + dir;
    const int sgrp = (blockIdx.z * NC + A.chunk0) * A.ndir + dir;
    const HyperDir& D = A.d[dir];
    // Roles (K quarter, epilogue duties) are rotated per team so that the teams' epilogue waves (role 0: main cell,
    // role 1: hyper cell) sit on different SIMDs: hardware wave w runs on SIMD w % 4
    const int lane = threadIdx.x & 63, wave = (int)(((threadIdx.x >> 6) + team * (4 / NC)) & 3);
    const int tid = wave * 64 + lane;
    const int q = lane >> 4, bl = lane & 15;
    const int b = chunk * A.rpc + bl;
    const bool bvalid = (bl < A.rpc) && (b < B);

    GroupSync gs;
    group_sync_init(gs, A.sync + (size_t)sgrp * SYNC_STRIDE, G, j, flag);
    {
        const f32x4* src = (const f32x4*)(D.wpack + (size_t)j * per_wg);
        f32x4* dst = (f32x4*)smem;
        for (size_t i = threadIdx.x; i < per_wg / 4; i += 256 * NC) dst[i] = src[i];
    }
    float* dax = A.dax + (size_t)grp * 2 * G * 256;
    float* px = A.px + (size_t)grp * 2 * G * 1024;
    float* dhx = A.dhx + (size_t)grp * 2 * G * G * UHB;
    float* dgx = A.dgx + (size_t)grp * 2 * H * 64;
    const __amdgpu_buffer_rsrc_t dax_rs = make_rsrc_(dax, (unsigned)((size_t)2 * G * 256 * 4));
    const __amdgpu_buffer_rsrc_t px_rs = make_rsrc_(px, (unsigned)((size_t)2 * G * 1024 * 4));
    const __amdgpu_buffer_rsrc_t dgx_rs = make_rsrc_(dgx, (unsigned)((size_t)2 * H * 64 * 4));
    const __amdgpu_buffer_rsrc_t dhx_rs = make_rsrc_(dhx, (unsigned)((size_t)2 * G * G * UHB * 4));
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    // W_h product roles: this wave's tile / K slice, first source block of its slice
    const int R = A.R, MT = A.MT, KQ = A.KQ, NBW = A.NBW;
    const int ub = j / R, kr = j % R;
    const int w_mt = wv % MT, w_kq = wv / MT;
    const int w_bi0 = w_kq * NBW;                        // first block of this wave inside the K range
    const int w_blk0 = kr * A.NBLK + w_bi0;              // ... as a source workgroup index
    const float* Ww = Wt + (size_t)wv * NBW * 256;

    const int unit = j * 4 + q, col4 = unit * 4;
    const int hu = j * UH + q;
    const bool hvalid = (q < UH) && (hu < H);
    float dc = 0.f, dh_pass = 0.f, dh_in = 0.f;         // wave 0 lanes
    float dhc = 0.f, dhh_pass = 0.f, dhh_rec = 0.f;     // wave 1 lanes
    // running column sums (d cvec, d bhat): formed here for free instead of re-reading dD / dXh
    f32x4 cs_h = {0, 0, 0, 0}, cs_x = {0, 0, 0, 0}, cs_g = {0, 0, 0, 0}, cs_hg = {0, 0, 0, 0};
    const int s0 = A.s0, s1 = A.s1;
    if (s0 > 0) {
        // a later time chunk of the BPTT: the gradient state and the running column sums continue from
        // where the previous launch (steps < s0) left them
        const float* cy = D.carry + (size_t)(bvalid ? b : 0) * (2 * N + 2 * H);
        if (wave == 0 && bvalid) { dc = cy[unit]; dh_in = cy[N + unit]; }
        if (wave == 1 && bvalid && hvalid) { dhc = cy[2 * N + hu]; dhh_rec = cy[2 * N + H + hu]; }
        if (D.dcv && wave == 0 && bl == 0) {
            const float* o_ = D.dcv + (size_t)chunk * 12 * N;
            cs_h = *(const f32x4*)(o_ + col4); cs_x = *(const f32x4*)(o_ + 4 * N + col4); cs_g = *(const f32x4*)(o_ + 8 * N + col4);
        }
        if (D.dbh && wave == 1 && bl == 0 && hvalid) cs_hg = *(const f32x4*)(D.dbh + (size_t)chunk * 4 * H + hu * 4);
    }
    __syncthreads();

    // saved tensors of the step about to be processed, prefetched one step ahead
    f32x4 n_g4 = {0, 0, 0, 0}, n_a4 = {0, 0, 0, 0}, n_dh4 = {0, 0, 0, 0}, n_dx4 = {0, 0, 0, 0}, n_x4 = {0, 0, 0, 0};
    f32x4 n_hg4 = {0, 0, 0, 0};
    float n_m = 0.f, n_dre = 0.f;
    // Branch-free on purpose.  Any control-flow join behind these loads (`st > 0 ? load : 0`, a
    // wave-role `if`, `if (step + 1 < T)`) makes hipcc place the phi copies of the loaded registers
    // right behind the loads, i.e. an s_waitcnt vmcnt(0..1) that puts the whole memory latency
    // (~3.4k cycles per step by the in-kernel stamps) on the critical path of every step.  So: every
    // wave issues every load, every step; lanes / waves that do not own a tensor read a dummy line
    // (address select, no branch), invalid batch rows read clamped in-range rows, and the selects
    // on the VALUES happen one iteration later, where they are consumed.
    const int bq = bvalid ? b : (B - 1);
    const int huq = hvalid ? hu : 0;
    const bool w0 = (wv == 0), w1 = (wv == 1);
    const float* const dummy = D.wpack;
    // Per-lane pointers that walk the saved tensors one time step per iteration (bumped by a signed
    // per-lane stride; 0 for the lanes that read the dummy line): the in-loop prefetch is a dozen loads
    // and pointer adds, with no kernel-argument reloads or 64-bit multiplies on the step's critical path
    const int dt = D.reverse ? 1 : -1;
    const int t00 = D.reverse ? s0 : T - 1 - s0;                          // time index of this launch's first step
    const size_t r00 = (size_t)t00 * B + bq;                              // its row
    const bool w01 = w0 || w1;
    const float* p_m = w01 ? A.mask + (size_t)t00 * A.mst + (size_t)bq * A.msb : dummy;
    const float* p_g = w0 ? D.gates + r00 * 4 * N + col4 : dummy;
    const float* p_x = w0 ? D.xproj + r00 * 4 * N + col4 : dummy;
    const float* p_a = w0 ? D.aD + r00 * 12 * N + col4 : dummy;
    const float* p_dr = w0 ? D.dr_ext + r00 * D.ld_dr + unit : dummy;
    const float* p_hg = w1 ? D.hgates + r00 * 4 * H + huq * 4 : dummy;
    const int s_m = w01 ? dt * A.mst : 0;
    const int s_g = w0 ? dt * B * 4 * N : 0, s_a = w0 ? dt * B * 12 * N : 0, s_c = w0 ? dt * B * N : 0;
    const int s_dr = w0 ? dt * B * D.ld_dr : 0;
    const int s_hg = w1 ? dt * B * 4 * H : 0, s_hc = w1 ? dt * B * H : 0;
    const int a_1 = w0 ? 4 * N : 0, a_2 = w0 ? 8 * N : 0;
    // The previous cell state of backward step s is the cell row of step s+1, so the two cell tensors
    // are read ONCE, two steps ahead (c0 = this step, c1 = next step = this step's c_prev, n_c2 in
    // flight); a second load of the same row behind a select made hipcc merge the two loads' results at
    // a join, i.e. wait for the fresh load (see above).
    auto row_of = [&](int k) { k = k < T ? k : T - 1; return (size_t)(D.reverse ? k : T - 1 - k) * B + bq; };
    const float* p_c2 = w0 ? D.cst + row_of(s0 + 2) * N + unit : dummy;
    const float* p_hc2 = w1 ? D.hcst + row_of(s0 + 2) * H + huq : dummy;
    float c0 = *(w0 ? D.cst + row_of(s0) * N + unit : dummy), c1 = *(w0 ? D.cst + row_of(s0 + 1) * N + unit : dummy);
    float hc0 = *(w1 ? D.hcst + row_of(s0) * H + huq : dummy), hc1 = *(w1 ? D.hcst + row_of(s0 + 1) * H + huq : dummy);
    float n_c2 = 0.f, n_hc2 = 0.f;
    // loads the step the pointers stand on
    auto prefetch = [&]() {
        n_m = *p_m;
        n_g4 = *(const f32x4*)p_g;
        n_a4 = *(const f32x4*)p_a;
        n_dh4 = *(const f32x4*)(p_a + a_1);
        n_dx4 = *(const f32x4*)(p_a + a_2);
        n_x4 = *(const f32x4*)p_x;
        n_dre = *p_dr;
        n_hg4 = *(const f32x4*)p_hg;
        n_c2 = *p_c2;
        n_hc2 = *p_hc2;
    };
    auto advance = [&](bool go) {
        p_m += go ? s_m : 0; p_g += go ? s_g : 0; p_x += go ? s_g : 0; p_a += go ? s_a : 0;
        p_dr += go ? s_dr : 0; p_hg += go ? s_hg : 0;
    };
    prefetch();
    STAMP_DECL
    for (int step = s0; step < s1; ++step) {
        const int st = T - 1 - step;
        const int t = D.reverse ? T - 1 - st : st;
        const int par = step & 1;
        const size_t row = (size_t)t * B + b;
        const float m = bvalid ? n_m : 0.f;
        const bool on = (m != 0.f);
        const f32x4 g4 = n_g4, a4 = n_a4, dh4 = n_dh4, dx4 = n_dx4, x4 = n_x4, hg4 = n_hg4;
        const float c_new = c0, c_prev = st > 0 ? c1 : 0.f, dre = bvalid ? n_dre : 0.f;
        const float hc_new = hc0, hc_prev = st > 0 ? hc1 : 0.f;
        advance(step + 1 < T);
        prefetch();                                        // step+1's tensors, step+2's cell rows
        p_c2 += (step + 3 < T) ? s_c : 0; p_hc2 += (step + 3 < T) ? s_hc : 0;
        STAMP(9);

        // ---- S1: main cell backward (wave 0)
        f32x4 dg4 = {0, 0, 0, 0}, da4 = {0, 0, 0, 0}, dd_h = {0, 0, 0, 0}, dd_x = {0, 0, 0, 0}, dd_b = {0, 0, 0, 0};
        if (wave == 0) {
            const float dh_t = dre + dh_pass + dh_in;       // dh_in: recurrent terms formed last step
            // HyperLHUC: h = r0 * reparam(d); slot 1 of aD holds d (gate column 0), d_h == 1
            float sc = 1.f, dsc = 0.f, ds_pre = 0.f;
            f32x4 dh4e = dh4;
            if (A.lhuc) { sc = lhuc_rp(A.lhuc, dh4[0]); dsc = lhuc_drp(A.lhuc, dh4[0]); dh4e = f32x4{1.f, 1.f, 1.f, 1.f}; }
            if (on) {
                const float ig = g4[0], fg = g4[1], gg = g4[2], og = g4[3];
                const float tc = tanhf_(c_new);
                const float dr0 = dh_t * sc;
                ds_pre = dh_t * (og * tc) * dsc;
                const float do_ = dr0 * tc;
                const float dct = dc + clipf_(dr0 * og * (1.f - tc * tc), A.clip_out);
                dg4[0] = clipf_(dct * gg * ig * (1.f - ig), A.clip);
                dg4[1] = clipf_(dct * c_prev * fg * (1.f - fg), A.clip);
                dg4[2] = clipf_(dct * ig * (1.f - gg * gg), A.clip);
                dg4[3] = clipf_(do_ * og * (1.f - og), A.clip);
                dc = dct * fg;
                dh_pass = 0.f;
            } else {
                dh_pass = dh_t;
            }
            da4 = dg4 * dh4e;
            dd_h = dg4 * a4; dd_x = dg4 * x4;
            dd_b = A.lhuc ? f32x4{ds_pre, 0.f, 0.f, 0.f} : dg4;
            cs_h += dd_h; cs_x += dd_x; cs_g += dd_b;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                dal[(q * 4 + k) * 16 + bl] = da4[k];
                dDl[0 * 256 + (q * 4 + k) * 16 + bl] = dd_h[k];
                dDl[1 * 256 + (q * 4 + k) * 16 + bl] = dd_x[k];
                dDl[2 * 256 + (q * 4 + k) * 16 + bl] = dd_b[k];
            }
            // all-gather payload: da of the own 16 columns in the operand order of the consumers'
            // 4x4x1 MFMAs (lane = ks*16 + b holds columns 4e+ks, e = 0..3): a 4x4 transpose through the
            // LDS tile just written (one wave, LDS executes a wave's accesses in order), then ONE
            // 16-byte write-through store per lane = 1 KB per workgroup and step
            if (step + 1 < T) {
                __atomic_signal_fence(__ATOMIC_SEQ_CST);
                f32x4 dap;
#pragma unroll
                for (int e = 0; e < 4; ++e) dap[e] = dal[((e * 4 + q) * 16) + bl];
                st_sc1_f4(dax_rs, (unsigned)((((size_t)par * G + j) * 256 + lane * 4) * 4), dap);
            }
        }
        STAMP(0);
        __syncthreads();
        // ---- S2: partial dhhat_t over the own 48 columns of M (-> dhx, reduce-scatter).  dD^T is the A
        // operand (rows = batch) and M^T the B operand (columns = hyper units), so a lane's accumulator
        // holds 4 consecutive batch rows of one hyper unit: one 16-byte write-through store per tile
        for (int ht = wave; ht < HT; ht += 4) {
            f32x4 acc = {0, 0, 0, 0};
#pragma unroll
            for (int tau = 0; tau < 3; ++tau) {
                const f32x4 mv = *(const f32x4*)(Mb + (((size_t)ht * 3 + tau) * 64 + lane) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = mfma4(dDl[tau * 256 + (4 * e + q) * 16 + bl], mv[e], acc);
            }
            const int hq = ht * 16 + bl;                      // hyper unit of this lane's accumulator
            if (hq < H)
                st_sc1_f4(dhx_rs, (unsigned)(((((size_t)par * G + hq / UH) * G + j) * UHB + (hq % UH) * 16 + q * 4) * 4), acc);
        }
        STAMP(1);
        group_publish(gs);
        STAMP(2);
        if (wave == 0 && bvalid) {        // bulk outputs after the publish: off the critical path
            *(f32x4*)(D.dA + row * 4 * N + col4) = da4;
            *(f32x4*)(D.dXp + row * 4 * N + col4) = dg4 * dx4;
            float* dd_ = D.dD + row * 12 * N;
            *(f32x4*)(dd_ + col4) = dd_h;
            *(f32x4*)(dd_ + 4 * N + col4) = dd_x;
            *(f32x4*)(dd_ + 8 * N + col4) = dd_b;
        }
        if (!group_wait(gs)) break;
        STAMP(3);
        // ---- S3: gather dhx for own hyper units.  Behind its loads, the first DAB blocks per wave of
        // the all-gathered da are put in flight already (consumed after publish2): their latency
        // overlaps S3, the hyper cell and the second exchange
        f32x4 dav[DABN];
        const unsigned da_sbase = (unsigned)(((size_t)par * G + w_blk0) * 1024);
        {
            const int e = tid % UHB, s0 = tid / UHB, sstep = 256 / UHB;
            float acc = 0.f;
            const unsigned sbase = (unsigned)(((size_t)par * G + j) * G * UHB * 4);
            const unsigned voff = (unsigned)((s0 * UHB + e) * 4);
            for (int c0 = 0; c0 < G; c0 += 8 * sstep) {       // src = s0 + c0 + i*sstep (beyond G*UHB of
                float v[8];                                   // this block only for the last dst: range-checked)
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    v[i] = ldb_sc1(dhx_rs, voff, sbase + (unsigned)((c0 + i * sstep) * UHB * 4));
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (s0 + c0 + i * sstep >= G) v[i] = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) acc += v[i];
            }
            STAMP(15);
#pragma unroll
            for (int i = 0; i < DABN; ++i)
                dav[i] = ldb_sc1_f4(dax_rs, (unsigned)(lane * 16), da_sbase + (unsigned)(i * 1024));
            hsum[tid] = acc;
        }
        STAMP(4);
        __syncthreads();
        // ---- S4: hyper cell backward (wave 1)
        f32x4 dhg = {0, 0, 0, 0};
        if (wave == 1 && hvalid) {
            float dsum = 0.f;
            const int sstep = 256 / UHB;
            for (int k = 0; k < sstep; ++k) dsum += hsum[k * UHB + q * 16 + bl];
            const float dhh_t = dhh_pass + dhh_rec;          // gradient wrt the state hhat_t
            if (on) {
                const float dhh_new = dhh_t + dsum;          // + via d = hhat_new.M of this step
                const float hi = hg4[0], hf = hg4[1], hgc = hg4[2], ho = hg4[3];
                const float thc = tanhf_(hc_new);
                const float dho = dhh_new * thc;
                const float dhct = dhc + clipf_(dhh_new * ho * (1.f - thc * thc), A.clip_out);
                dhg[0] = clipf_(dhct * hgc * hi * (1.f - hi), A.clip);
                dhg[1] = clipf_(dhct * hc_prev * hf * (1.f - hf), A.clip);
                dhg[2] = clipf_(dhct * hi * (1.f - hgc * hgc), A.clip);
                dhg[3] = clipf_(dho * ho * (1.f - ho), A.clip);
                dhc = dhct * hf;
                dhh_pass = 0.f;
            } else {
                dhh_pass = dhh_t;
            }
            cs_hg += dhg;
            // all-gather payload: dgx[par][hu][b][gate], one 16-byte write-through store
            st_sc1_f4(dgx_rs, (unsigned)((((size_t)par * H + hu) * 16 + bl) * 16), dhg);
        }
        STAMP(5);
        if (step + 1 == T) {                            // nothing consumes the last step's feedback
            if (wave == 1 && hvalid && bvalid) *(f32x4*)(D.dXh + row * 4 * H + hu * 4) = dhg;
            break;
        }
        // ---- partial dh_{t-1}: rows of the unit block x own K range of the all-gathered da (published
        // with exchange 1).  da^T is the A operand (rows = batch), W_h^T the B operand (columns = units):
        // a lane's accumulator = 4 consecutive batch rows of one unit -> one 16-byte store per lane
        {
            f32x4 acc = {0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < DABN; ++i) {
                const bool ok = (i < NBW) && (w_bi0 + i < A.NBLK);
                const f32x4 wt = *(const f32x4*)(Ww + ((size_t)(i < NBW ? i : 0) * 64 + lane) * 4);
                const f32x4 dv = ok ? dav[i] : f32x4{0, 0, 0, 0};
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = mfma4(dv[e], wt[e], acc);
            }
            for (int i0 = DABN; i0 < NBW; i0 += DABN) {
                f32x4 v[DABN];
#pragma unroll
                for (int i = 0; i < DABN; ++i)
                    v[i] = ldb_sc1_f4(dax_rs, (unsigned)(lane * 16), da_sbase + (unsigned)((i0 + i) * 1024));
#pragma unroll
                for (int i = 0; i < DABN; ++i) {
                    const bool ok = (i0 + i < NBW) && (w_bi0 + i0 + i < A.NBLK);
                    const f32x4 wt = *(const f32x4*)(Ww + ((size_t)(i0 + i < NBW ? i0 + i : 0) * 64 + lane) * 4);
                    const f32x4 dv = ok ? v[i] : f32x4{0, 0, 0, 0};
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc = mfma4(dv[e], wt[e], acc);
                }
            }
            STAMP(12);
            // tile [unit bl][batch q*4..q*4+3] of this wave
            st_sc1_f4(px_rs, (unsigned)(((((size_t)par * G + j) * 4 + wv) * 256 + bl * 16 + q * 4) * 4), acc);
        }
        STAMP(10);
        group_publish(gs);
        STAMP(6);
        if (wave == 1 && hvalid && bvalid) *(f32x4*)(D.dXh + row * 4 * H + hu * 4) = dhg;
        STAMP(11);
        if (!group_wait(gs)) break;
        STAMP(7);
        // ---- S5: own rows of [W_xhat[:N] ; W_hhat] times the gathered hyper gate grads; gather of the
        // R x KQ partial W_h.da pieces (4 own units x 16 batch = 256 B each) of the own unit block
        // (the partial pieces are requested first and parked in LDS only after the Rw product below: a store behind the
        // load would make hipcc wait for it before the dgx loads are even issued: two memory latencies in a row)
        const int pc5 = tid >> 4, f45 = tid & 15;             // piece = (member s, K slice kq2)
        f32x4 pv5;
        {
            const int s_ = pc5 / KQ, kq2 = pc5 % KQ;
            const int ul = kr * 4;                            // own units inside the block
            const int tile = kq2 * MT + ul / 16;
            pv5 = ldb_sc1_f4(px_rs, (unsigned)((((ul & 15) * 16) + f45 * 4) * 4),
                             (unsigned)(((((size_t)par * G + ub * R + (s_ < R ? s_ : 0)) * 4 + tile) * 256) * 4));
        }
        {
            f32x4 acc = {0, 0, 0, 0};
            const int RR = 4 + UH;
            const int ir = (bl < RR ? bl : RR - 1);
            const int n4 = S4P / 16;
            for (int c0 = 0; c0 < n4; c0 += DABN) {
                f32x4 bv[DABN];
#pragma unroll
                for (int i = 0; i < DABN; ++i)      // MFMA k = hyper unit 4*s4+q, step e = gate: dgx[par][unit][bl][0..3]
                    bv[i] = ldb_sc1_f4(dgx_rs, (unsigned)((q * 64 + bl * 4) * 4),
                                       (unsigned)(par * H * 256) + (unsigned)(wv * n4 + c0 + i) * 1024u);
#pragma unroll
                for (int g4i = 0; g4i < DABN; ++g4i) {
                    if (c0 + g4i < n4) {
                        const int s4 = wave * n4 + c0 + g4i;
                        const f32x4 rv = *(const f32x4*)(Rw + (((size_t)s4 * 4 + q) * RR + ir) * 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc = mfma4(rv[e], bv[g4i][e], acc);
                    }
                }
            }
            STAMP(13);
            *(f32x4*)(red + ((size_t)wave * 64 + lane) * 4) = acc;
        }
        if (pc5 / KQ >= R) pv5 = f32x4{0, 0, 0, 0};
        *(f32x4*)(rsum + ((size_t)pc5 * 16 + f45) * 4) = pv5;
        __syncthreads();
        STAMP(14);
        // output rows i = q*4+reg: i<4 -> main unit i (lanes q==0, reg=i); i=4+uh -> hyper (q==1, reg=uh)
        if (wave == 0) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) v += red[((size_t)w * 64 + (0 * 16 + bl)) * 4 + q];
#pragma unroll
            for (int pc = 0; pc < 16; ++pc) v += rsum[pc * 64 + q * 16 + bl];        // R x KQ partial pieces
            dh_in = v;
        }
        if (wave == 1) {
            float v = 0.f;
            if (q < UH) {
#pragma unroll
                for (int w = 0; w < 4; ++w) v += red[((size_t)w * 64 + (1 * 16 + bl)) * 4 + q];
            }
            dhh_rec = v;
        }
        c0 = c1; c1 = n_c2; hc0 = hc1; hc1 = n_hc2;
        STAMP(8);
        __syncthreads();
    }
    if (s1 < T && D.carry && bvalid) {                  // hand the BPTT state to the next time chunk's launch
        float* cy = D.carry + (size_t)b * (2 * N + 2 * H);
        if (wave == 0) { cy[unit] = dc; cy[N + unit] = dh_pass + dh_in; }
        if (wave == 1 && hvalid) { cy[2 * N + hu] = dhc; cy[2 * N + H + hu] = dhh_pass + dhh_rec; }
    }
    if (D.dcv && wave == 0) {                           // sum over the 16 batch lanes, lane bl == 0 writes
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                cs_h[e] += __shfl_xor(cs_h[e], o); cs_x[e] += __shfl_xor(cs_x[e], o); cs_g[e] += __shfl_xor(cs_g[e], o);
            }
        }
        if (bl == 0) {
            float* o_ = D.dcv + (size_t)chunk * 12 * N;
            *(f32x4*)(o_ + col4) = cs_h; *(f32x4*)(o_ + 4 * N + col4) = cs_x; *(f32x4*)(o_ + 8 * N + col4) = cs_g;
        }
    }
    if (D.dbh && wave == 1) {
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) cs_hg[e] += __shfl_xor(cs_hg[e], o);
        }
        if (bl == 0 && hvalid) *(f32x4*)(D.dbh + (size_t)chunk * 4 * H + hu * 4) = cs_hg;
    }
    STAMP_DUMP(A.dbg, (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
}

// ------------------------------------------------------------------------------------------------
// cap on the 16-row chunks one workgroup carries (scan_common.h: scan_nc_cap): 0 = ASR_SCAN_NC or the default
int asr_scan_nc_cap_ = 0;
extern "C" int asr_scan_set_chunks_per_wg(int nc) {
    if (nc != 0 && nc != 1 && nc != 2 && nc != 4) return ASR_ERR_ARG;
    asr_scan_nc_cap_ = nc;
    return ASR_OK;
}

struct asr_hyper_scan_desc {
    int T, B, N, H, ndir;
    int step0, step1;            // steps [step0, step1) of the recurrence, in processing order; step1 == 0: all T
    float grad_clip;
    int clip_outcell;
    int lhuc_reparam;            // 0: HyperLSTM / scaling; 1 relu, 2 exp, 3 identity, 4 2sigmoid: HyperLHUCLSTMLayer
    const float* mask; int mask_stride_b, mask_stride_t;
    HyperDir dir[2];
};

extern "C" size_t asr_hyper_scan_ws_bytes(int T, int B, int N, int H, int ndir, int backward) {
    (void)T;
    HyperArgs A;
    hyper_dims(N, H, &A);
    const size_t ng = (size_t)ndir * ((B + 7) / 8);     // sized for the 8-row chunking
    size_t bytes = ng * SYNC_STRIDE * 4;
    bytes += ng * 2 * A.KSP * 64 * 4 + ng * 2 * A.HSP * 64 * 4;
    if (backward) {
        bytes += ng * 2 * (size_t)A.G * 256 * 4;
        bytes += ng * 2 * (size_t)A.G * 1024 * 4;
        bytes += ng * 2 * (size_t)A.G * A.G * A.UH * 16 * 4;
        bytes += ng * 2 * (size_t)H * 64 * 4;
    }
    return bytes + 1024 + STAMP_BYTES;
}

static int hyper_launch(const asr_hyper_scan_desc* d, void* ws, size_t ws_bytes, int backward,
                        hipStream_t stream) {
    if (!d || !ws || d->T < 0 || d->B <= 0 || d->N <= 0 || d->H <= 0 || (d->N & 3) || (d->H & 3) ||
        d->H > d->N || d->ndir < 1 || d->ndir > 2 || !d->mask)
        return ASR_ERR_ARG;
    HyperArgs A;
    hyper_dims(d->N, d->H, &A);
    if (A.UH > 4) return ASR_ERR_ARG;
    for (int k = 0; k < d->ndir; ++k) {
        const HyperDir& D = d->dir[k];
        if (!D.wpack || !D.xproj) return ASR_ERR_ARG;
        if (!backward && (!D.xhproj || !D.cvec || !D.r_out || !D.hh_out)) return ASR_ERR_ARG;
        if (backward && (!D.gates || !D.cst || !D.hgates || !D.hcst || !D.aD || !D.dr_ext || !D.dXp ||
                         !D.dXh || !D.dA || !D.dD))
            return ASR_ERR_ARG;
    }
    if (ws_bytes < asr_hyper_scan_ws_bytes(d->T, d->B, d->N, d->H, d->ndir, backward)) return ASR_ERR_WS;
    if (d->T == 0) return ASR_OK;
    A.s0 = d->step0; A.s1 = d->step1 > 0 ? d->step1 : d->T;
    if (A.s0 < 0 || A.s0 >= A.s1 || A.s1 > d->T) return ASR_ERR_ARG;
    for (int k = 0; k < d->ndir; ++k) {                  // a time chunk other than the first continues saved state
        const HyperDir& D = d->dir[k];
        if (!backward && A.s0 > 0 && (!D.cst || !D.hcst)) return ASR_ERR_ARG;
        if (backward && (A.s0 > 0 || A.s1 < d->T) && !D.carry) return ASR_ERR_ARG;
    }
    if (A.G * d->ndir > 256) return ASR_ERR_LIMIT;       // co-residency bound of the persistent grid
    A.T = d->T; A.B = d->B; A.ndir = d->ndir; A.clip = d->grad_clip;
    A.clip_out = d->clip_outcell ? d->grad_clip : 0.f;
    if (d->lhuc_reparam < 0 || d->lhuc_reparam > 4) return ASR_ERR_ARG;
    A.lhuc = d->lhuc_reparam;
    A.mask = d->mask; A.msb = d->mask_stride_b; A.mst = d->mask_stride_t;
    A.d[0] = d->dir[0]; A.d[1] = d->dir[d->ndir > 1 ? 1 : 0];
    const size_t per_wg = hyper_pack_floats_wg(A, backward);
    // LDS of a workgroup that carries nc chunks: the resident weights once, the staging buffers per team
    auto lds_of = [&](int nc) -> size_t {
        return !backward ? (per_wg + (size_t)nc * 3 * 4 * 256) * 4 + 64 : (per_wg + (size_t)nc * HYPER_BWD_STAGE) * 4 + 64;
    };
    if (lds_of(1) > 160 * 1024) return ASR_ERR_LIMIT;
    // Knobs, read once per process (experiments only): ASR_SCAN_RPC = 8 cuts the batch into 8-row chunks (two
    // independent recurrences sharing a CU: measured SLOWER, 482 vs 325 ms/step on cfg3, the exchange is bound by
    // memory-side traffic, which doubles); ASR_SCAN_NC / asr_scan_set_chunks_per_wg cap the chunks per workgroup (1, 2 or 4).
    static const int env_rpc = [] { const char* e = getenv("ASR_SCAN_RPC"); return e && atoi(e) == 8 ? 8 : 16; }();
    const int env_nc = scan_nc_cap();
    A.rpc = env_rpc;
    const int nchunk = (d->B + A.rpc - 1) / A.rpc;
    const size_t ng = (size_t)d->ndir * ((d->B + 7) / 8);
    char* p = (char*)ws;
    A.sync = (unsigned*)p;  p += ng * SYNC_STRIDE * 4;
    A.rex = (float*)p;      p += ng * 2 * A.KSP * 64 * 4;
    A.hex = (float*)p;      p += ng * 2 * A.HSP * 64 * 4;
    A.dax = A.px = A.dhx = A.dgx = nullptr;
    if (backward) {
        A.dax = (float*)p;  p += ng * 2 * (size_t)A.G * 256 * 4;
        A.px = (float*)p;   p += ng * 2 * (size_t)A.G * 1024 * 4;
        A.dhx = (float*)p;  p += ng * 2 * (size_t)A.G * A.G * A.UH * 16 * 4;
        A.dgx = (float*)p;
    }
    if (hipMemsetAsync(A.sync, 0, ng * SYNC_STRIDE * 4, stream) != hipSuccess) return ASR_ERR_LAUNCH;
    A.dbg = (unsigned long long*)((char*)ws + asr_hyper_scan_ws_bytes(d->T, d->B, d->N, d->H, d->ndir, backward) - STAMP_BYTES);
    const int per_z = A.G * d->ndir;                     // workgroups of one z slice (all groups of NC chunks)
    for (int c0 = 0; c0 < nchunk;) {
        const int rem = nchunk - c0;
        // chunks per workgroup: 1 or 2, capped by ASR_SCAN_NC, by the rpc = 8 experiment and by the LDS
        // (hyper scans: at most 2.  Four teams = 1024 threads leave 128 registers per wave; both kernels then spill
        // 80-150 registers into scratch and ran slower than two launches of two teams: backward 93.8 vs 74.7 ms per
        // 64-row layer, forward 56.3 vs 48.0, profiles/r03_batch_sweep.jsonl)
        // Teams only for chunks that do not fit the launch as workgroups of their own (one per CU)
        const int zcap = 256 / per_z > 0 ? 256 / per_z : 1;
        int nc = (rem > zcap && env_nc >= 2) ? 2 : 1;
        if (A.rpc != 16) nc = 1;
        while (nc > 1 && lds_of(nc) > 160 * 1024) nc >>= 1;
        size_t lds = lds_of(nc);
        // Two workgroups fit one CU when each needs <= 80 KB of LDS (only taken with teams switched off); one
        // workgroup per CU whenever the grid fits the chip that way (the sc1 hand-off is measured for one workgroup
        // per CU: MI355X_MICROARCH.md, table under 'Valid forms'): then ask for more than half of a CU's 160 KB
        int zmax = ((lds <= 80 * 1024 && nc == 1) ? 512 : 256) / per_z;
        if (zmax < 1) zmax = 1;
        int nz = nc == 2 ? rem / 2 : rem;
        if (nz > zmax) nz = zmax;
        if ((size_t)per_z * nz <= 256 && lds < 82 * 1024) lds = 82 * 1024;
        A.chunk0 = c0;
        dim3 grid(A.G, d->ndir, nz), block(256 * nc);
        if (!scan_grid_fits((size_t)per_z * nz, lds)) return ASR_ERR_LIMIT;
        const void* fn;
        if (!backward) fn = nc == 2 ? (const void*)hyper_scan_fwd_kernel<2> : (const void*)hyper_scan_fwd_kernel<1>;
        else fn = nc == 2 ? (const void*)hyper_scan_bwd_kernel<2> : (const void*)hyper_scan_bwd_kernel<1>;
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return ASR_ERR_LAUNCH;
        void* kargs[] = {(void*)&A, (void*)&per_wg};
        if (hipLaunchKernel(fn, grid, block, kargs, lds, stream) != hipSuccess) return ASR_ERR_LAUNCH;
        ASR_CHECK_LAUNCH();
        c0 += nz * nc;
    }
    return ASR_OK;
}

extern "C" int asr_hyper_scan_fwd(const asr_hyper_scan_desc* d, void* ws, size_t ws_bytes, hipStream_t stream) {
    return hyper_launch(d, ws, ws_bytes, 0, stream);
}
extern "C" int asr_hyper_scan_bwd(const asr_hyper_scan_desc* d, void* ws, size_t ws_bytes, hipStream_t stream) {
    return hyper_launch(d, ws, ws_bytes, 1, stream);
}
