// K2 (LSTM family): persistent recurrent scan, forward and BPTT, for
//   libs/lhuclstm_layers.py:73-130 (LSTMLayer / TBPTTLSTMLayer / LHUC base recurrence),
//   libs/lasagne_libs/recurrent_layers.py:437-498 (ProjectLSTMLayer), :707-763
//   (LayerNormProjectLSTMLayer), :1237-1289 (InputCondLayerNormProjectLSTMLayer).
//
// What the kernel sees (prepared by the host side, see asr_amd/engine.py):
//   * xin (T,B,4N): everything on the input side of the gate pre-activation, already including the
//     bias and (for LN variants) the input layer norm  -- not recurrent, so hoisted out of the loop.
//   * the recurrent matrix with the output projection folded in, W' = W_p . W_h (N x 4N), so the
//     recurrent state is r_t = o * tanh(outcell) (N wide) for every variant and the step needs ONE
//     all-gather of r; the projected output h_t = r_t . W_p is one large MFMA GEMM after the scan.
//   * gate-interleaved columns: col = unit*4 + gate (gate order i,f,c,o).
// Each workgroup owns UT tiles of 4 units, keeps its W' slice resident in LDS as MFMA A-fragments
// for all T steps, computes  a^T[16 cols x 16 batch] = W'^T . r^T  with v_mfma_f32_16x16x4_f32
// (4 waves split K), and fuses LN statistics, gates, cell update, masking and the saves for BPTT
// into the step.  One launch covers both directions (and several 16-row batch chunks).
#include "scan_common.h"
#include <stdlib.h>

struct LstmDir {
    int reverse;
    const float* xin;        // (T,B,4N)
    const float* wpack;      // fwd: asr_lstm_pack_w(...,0)   bwd: asr_lstm_pack_w(...,1)
    const float* alpha_hid;  // (4N) or (B,4N) when ln_kind==2
    const float* beta_hid;   // (4N) or null
    const float* alpha_c;    // (N)
    const float* beta_c;     // (N)
    const float* c0;         // (B,N) or null
    const float* r0;         // (B,N) or null
    const float* a0;         // (B,4N) or null: the recurrent pre-activation h_{-1}.W_h of the FIRST step, used in place of
                             // r0.W' there.  A stateful projected layer carries hid_init = h of the previous window,
                             // projected with the W_p of THAT window (libs/lhuclstm_layers.py:313-319); after an optimiser
                             // step r0.(W_p.W_h) with the new W_p is not h_init.W_h any more (forward scans only)
    float* r_out;            // (T,B,ld_r) state r after each step (time order)
    int ld_r;
    float* gates;            // (T,B,4N) post-activation i,f,g,o       (null: inference)
    float* cst;              // (T,B,N)  cell state after each step   (required)
    float* ahat;             // (T,B,4N) normalised recurrent pre-activation (LN)
    float* rs_a;             // (T,B)
    float* chat;             // (T,B,N)  normalised cell (cell LN)
    float* rs_c;             // (T,B)
    // backward only
    const float* dr_ext;     // (T,B,ld_dr) dLoss/d r_t from the layers above
    int ld_dr;
    float* dG;               // (T,B,4N) clipped gradient wrt the gate pre-activations
    float* dA;               // (T,B,4N) gradient wrt a_t = r_{t-1}.W'   (== dG when no LN)
    float* dOut;             // (T,B,N)  clipped gradient wrt outcell (cell LN only)
    // options of the recurrent_layers.py classes (peepholes, learn_init, only_return_hidden=False; :52-57).  Any of
    // them set selects the OPT instantiation of the generic kernels, one 16-row chunk per workgroup.
    const float* peep;       // (N,4) per unit: w_cell_to_ingate, w_cell_to_forgetgate, 0, w_cell_to_outgate; or null
    int a0_until_valid;      // a0 stands in for r.W' until the row's first unmasked step, not only at step 0 (a learned
                             // hid_init of a projected layer is carried through leading masked steps, :246-248,:239)
    const float* dc_ext;     // backward in:  (T,B,ld_dc) dLoss/d cell_t from the layers above (cell_out returned, :275-288)
    int ld_dc;
    float* dc0;              // backward out: (B,N) gradient wrt the initial cell row
    float* dr0;              // backward out: (B,N) gradient that reached r_{-1} THROUGH masked steps; the caller adds the
                             //               first scan step's da.W'^T (that product is not formed inside the launch)
    float* da0;              // backward out: (B,4N) da of each row's first unmasked step (zeros for an empty row)
    float* dpeep;            // backward out: (B,N,4) per-row sums over time of dg_i c_{t-1}, dg_f c_{t-1}, 0, dg_o c_t
};
static inline bool lstm_dir_has_options(const LstmDir& D) {
    return D.peep || D.a0_until_valid || D.dc_ext || D.dc0 || D.dr0 || D.da0 || D.dpeep;
}

struct LstmScanArgs {
    int T, B, N, UT, G, KS, KSP, KT;
    int ndir, nchunk, chunk0;
    int ln_kind;             // 0 none, 1 static alpha[,beta], 2 per-sample alpha
    unsigned long long* dbg; // STAMP_BYTES at the workspace tail (diagnostic builds)
    int per_gate;            // LN statistics per gate (gate_ln / sample_gate_ln) instead of per row
    int cell_ln;
    float clip;
    const float* mask; int msb, mst;
    LstmDir d[2];
    unsigned* sync;          // [ngroups][SYNC_STRIDE]
    float* rex;              // [ngroups][2][KSP tiles][16 b][4 units]
    float* sta;              // [ngroups][2][KS][4 gates][32]  (gate 0 only unless per_gate)
    float* stc;              // [ngroups][2][KS][32]
    float* dex;              // [ngroups][2][KS dst][G src][64]      (backward)
};

// ------------------------------------------------------------------------------------------------
// weight packing.  W (N x 4N) row-major, gate-interleaved columns.
//  mode 0 (forward A-fragments):  out[tile][s4][l][e] = W[4*(4*s4+(l>>4)) + e][tile*16 + (l&15)]
//          (MFMA k index l>>4 <-> source tile 4*s4+k, the 4 steps e of a group <-> the tile's 4 units)
//  mode 1 (backward A-fragments): out[tile][kt][l][e] = W[kt*16 + (l&15)][tile*16 + 4*e + (l>>4)]
//  mode 2 (backward, 2-D decomposition of lstm_scan_bwd_ln2d_kernel; N % 128 == 0):
//          out[j = ub*8+kr][wave][i][t][l][e] = W[ub*32 + t*16 + (l&15)][16*(kr*KS/8 + wave*TW + i) + 4*(l>>4) + e]
// rows/steps beyond N are zero.
__global__ void lstm_pack_kernel(const float* __restrict__ W, int N, int KS, int KSP, int KT, int mode,
                                 float* __restrict__ out) {
    const int TW = KS / 32;                             // mode 2: source tiles per wave
    const int per_tile = mode == 0 ? KSP * 64 : mode == 1 ? KT * 256 : TW * 2048;
    const size_t tot = (size_t)KS * per_tile;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < tot;
         i += (size_t)gridDim.x * blockDim.x) {
        const int tile = (int)(i / per_tile);
        const int r = (int)(i % per_tile);
        const int e = r & 3, l = (r >> 2) & 63, blk = r >> 8;
        int row, col;
        if (mode == 0) { row = 4 * (4 * blk + (l >> 4)) + e; col = tile * 16 + (l & 15); }
        else if (mode == 1) { row = blk * 16 + (l & 15);     col = tile * 16 + 4 * e + (l >> 4); }
        else {              // workgroup `tile` = (ub, kr); blk = (wave * TW + i) * 2 + t
            const int ub = tile / 8, kr = tile % 8, t = blk & 1, wi = blk >> 1;
            row = ub * 32 + t * 16 + (l & 15);
            col = 16 * (kr * (KS / 8) + wi) + 4 * (l >> 4) + e;
        }
        out[i] = row < N ? W[(size_t)row * 4 * N + col] : 0.0f;
    }
}

static inline void lstm_dims(int N, int* KS, int* KSP, int* KT) {
    *KS = N / 4;
    *KSP = ((*KS + 15) / 16) * 16;
    *KT = (N + 15) / 16;
}

extern "C" size_t asr_lstm_pack_floats(int N, int mode) {
    int KS, KSP, KT;
    lstm_dims(N, &KS, &KSP, &KT);
    return mode == 0 ? (size_t)KS * KSP * 64 : mode == 1 ? (size_t)KS * KT * 256 : (size_t)KS * (KS / 32) * 2048;
}

// Which packing the backward scan of a layer expects (1 or 2): 2 = the 2-D decomposition, taken for the LN
// variants (ln_kind 1..4) whose grid is one workgroup per 4 units (N/4 * ndir <= 256) with N a multiple of 128,
// N <= 512.  The caller packs W with this mode and the launcher selects the kernel by the same rule.
extern "C" int asr_lstm_bwd_pack_mode(int N, int ndir, int ln_kind) {
    if (ln_kind >= 1 && ln_kind <= 4 && N > 0 && N % 128 == 0 && N <= 512 && ndir >= 1 && (N / 4) * ndir <= 256)
        return 2;
    return 1;
}

extern "C" int asr_lstm_pack_w(const float* W, int N, int mode, float* out, hipStream_t stream) {
    if (!W || !out || N <= 0 || (N & 3) || mode < 0 || mode > 2 || (mode == 2 && N % 128)) return ASR_ERR_ARG;
    int KS, KSP, KT;
    lstm_dims(N, &KS, &KSP, &KT);
    size_t tot = asr_lstm_pack_floats(N, mode);
    int grid = (int)((tot + 255) / 256 < 2048 ? (tot + 255) / 256 : 2048);
    hipLaunchKernelGGL(lstm_pack_kernel, dim3(grid), dim3(256), 0, stream, W, N, KS, KSP, KT, mode, out);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}

// ------------------------------------------------------------------------------------------------
// Combine per-tile (mean, M2) statistics (equal counts cnt per tile) into the row mean / rstd over
// all KS tiles.  Called by a whole epilogue wave; lane (q,bl) returns the stats of batch row bl.
// The per-tile pairs are read with sc1 BUFFER loads, all issued before the first use and with no branch
// around any of them (tiles beyond KS fall outside the descriptor's range and read as 0): the conditional
// 8-byte atomic loads this replaces were waited for one by one, ~20k cycles per call at KS = 128.
// ALL FOUR WAVES of the workgroup take part (call it outside any `wave < UT` branch): wave w, lane (q, bl)
// sums tiles 16 i + 4 w + q of batch row bl, the four partial sums meet in LDS (`scr`, 256 floats, free at
// the call sites; the caller's next barrier precedes its next write).
template <int COMBINE_CHUNK, typename F>
__device__ __forceinline__ void for_tile_pairs(const float* st, int KS, int tstride, int first, int stride, int bl,
                                               int sub, F&& f) {
    // `st` and the descriptor are workgroup-uniform (a per-wave base would make hipcc wrap every load in a
    // readfirstlane loop); `sub` = float offset of this wave's 32-float block inside a tile's tstride floats
    const __amdgpu_buffer_rsrc_t rs = make_rsrc_(st, (unsigned)(KS * tstride * 4));
    // 16-byte loads (the pairs of batch rows bl & ~1 and bl | 1), the lane keeps its own half: the sc1 hand-off
    // behind replicated arrival counters is measured for 4- and 16-byte loads, not 8-byte ones
    // (MI355X_MICROARCH.md, table under 'Valid forms')
    const unsigned voff = (unsigned)((first * tstride + sub + (bl & ~1) * 2) * 4);
    const bool hi = bl & 1;
    for (int s0 = 0; s0 < KS; s0 += stride * COMBINE_CHUNK) {
        f32x4 v[COMBINE_CHUNK];
#pragma unroll
        for (int i = 0; i < COMBINE_CHUNK; ++i) v[i] = ldb_sc1_f4(rs, voff, (unsigned)((s0 + stride * i) * tstride * 4));
#pragma unroll
        for (int i = 0; i < COMBINE_CHUNK; ++i) f(hi ? v[i][2] : v[i][0], hi ? v[i][3] : v[i][1]);
    }
}
__device__ __forceinline__ void stats_finish(float sm, float sm2, float sM2, int KS, float cnt, float& mean,
                                             float& rstd) {
    mean = sm / KS;
    float between = sm2 - KS * mean * mean;
    if (between < 0.f) between = 0.f;
    const float var = (sM2 + cnt * between) / (cnt * KS);
    rstd = 1.0f / sqrtf(var + ASR_EPS);
}
// row statistics over all KS tiles (equal counts cnt per tile); every lane (q, bl) of every wave gets row bl's
__device__ __forceinline__ void combine_stats(const float* st, int KS, int tid, float cnt, float& mean,
                                              float& rstd, float* scr, int tstride = 32) {
    const int w = tid >> 6, q = (tid >> 4) & 3, bl = tid & 15;
    float sm = 0.f, sm2 = 0.f, sM2 = 0.f;
    for_tile_pairs<8>(st, KS, tstride, 4 * w + q, 16, bl, 0, [&](float mj, float M2j) { sm += mj; sm2 += mj * mj; sM2 += M2j; });
    sm = sum_q(sm); sm2 = sum_q(sm2); sM2 = sum_q(sM2);
    if (q == 0) { scr[(w * 16 + bl) * 4] = sm; scr[(w * 16 + bl) * 4 + 1] = sm2; scr[(w * 16 + bl) * 4 + 2] = sM2; }
    __syncthreads();
    sm = 0.f; sm2 = 0.f; sM2 = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) { sm += scr[(k * 16 + bl) * 4]; sm2 += scr[(k * 16 + bl) * 4 + 1]; sM2 += scr[(k * 16 + bl) * 4 + 2]; }
    stats_finish(sm, sm2, sM2, KS, cnt, mean, rstd);
}
// per-gate statistics (gate g's pairs at st + 32 g, tile stride 128): wave g combines gate g
__device__ __forceinline__ void combine_stats4(const float* st, int KS, int tid, float cnt, f32x4& mean4,
                                               f32x4& rstd4, float* scr) {
    const int w = tid >> 6, q = (tid >> 4) & 3, bl = tid & 15;
    float sm = 0.f, sm2 = 0.f, sM2 = 0.f;
    for_tile_pairs<32>(st, KS, 128, q, 4, bl, w * 32, [&](float mj, float M2j) { sm += mj; sm2 += mj * mj; sM2 += M2j; });
    sm = sum_q(sm); sm2 = sum_q(sm2); sM2 = sum_q(sM2);
    float mean, rstd;
    stats_finish(sm, sm2, sM2, KS, cnt, mean, rstd);
    if (q == 0) { scr[(w * 16 + bl) * 2] = mean; scr[(w * 16 + bl) * 2 + 1] = rstd; }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 4; ++g) { mean4[g] = scr[(g * 16 + bl) * 2]; rstd4[g] = scr[(g * 16 + bl) * 2 + 1]; }
}
// plain sums (backward LN: sum dxh, sum dxh*xhat) over all tiles
__device__ __forceinline__ void combine_sums(const float* st, int KS, int tid, float& s1, float& s2, float* scr,
                                             int tstride = 32) {
    const int w = tid >> 6, q = (tid >> 4) & 3, bl = tid & 15;
    float a = 0.f, b = 0.f;
    for_tile_pairs<8>(st, KS, tstride, 4 * w + q, 16, bl, 0, [&](float x, float y) { a += x; b += y; });
    a = sum_q(a); b = sum_q(b);
    if (q == 0) { scr[(w * 16 + bl) * 2] = a; scr[(w * 16 + bl) * 2 + 1] = b; }
    __syncthreads();
    s1 = 0.f; s2 = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) { s1 += scr[(k * 16 + bl) * 2]; s2 += scr[(k * 16 + bl) * 2 + 1]; }
}
__device__ __forceinline__ void combine_sums4(const float* st, int KS, int tid, f32x4& s14, f32x4& s24, float* scr) {
    const int w = tid >> 6, q = (tid >> 4) & 3, bl = tid & 15;
    float a = 0.f, b = 0.f;
    for_tile_pairs<32>(st, KS, 128, q, 4, bl, w * 32, [&](float x, float y) { a += x; b += y; });
    a = sum_q(a); b = sum_q(b);
    if (q == 0) { scr[(w * 16 + bl) * 2] = a; scr[(w * 16 + bl) * 2 + 1] = b; }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 4; ++g) { s14[g] = scr[(g * 16 + bl) * 2]; s24[g] = scr[(g * 16 + bl) * 2 + 1]; }
}

// NC (all four scan kernels below) = 16-row batch chunks carried by one workgroup: NC teams of 4 waves run the same
// program on consecutive chunks against the same resident weights and share every publish / wait of the step (see
// hyper_scan_fwd_kernel, scan_hyper.hip).  Roles (K quarter, epilogue duty) are rotated per team so that the teams'
// epilogue waves sit on different SIMDs.  Per-row results do not depend on NC.
#define TEAM_IDS_(NC_)                                                                                   \
    const int team = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);   /* uniform, and known to be: scan_hyper.hip */ \
    const int j = blockIdx.x, dir = blockIdx.y, chunk = blockIdx.z * NC_ + team + A.chunk0;              \
    const int grp = chunk * A.ndir + dir;                                 /* exchange buffers: per chunk */ \
    const int sgrp = (blockIdx.z * NC_ + A.chunk0) * A.ndir + dir;        /* arrival counters: per workgroup set */ \
    const LstmDir& D = A.d[dir];                                                                         \
    const int lane = threadIdx.x & 63, wave = (int)(((threadIdx.x >> 6) + team * (4 / NC_)) & 3);        \
    const int tid = wave * 64 + lane;

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// STREAM (layers too wide for resident weights: 16 UT N 4 bytes per workgroup above the CU's 160 KB, i.e. N > ~1280): the
// A fragments are read from the packed image in global memory every step (L2 / Infinity Cache) instead of from LDS.
// Same arithmetic in the same order; the step then streams N * 64 UT bytes per workgroup and is bandwidth-bound.
template <int UT, int NC, bool OPT = false, bool STREAM = false>
__global__ __launch_bounds__(256 * NC) void lstm_scan_fwd_kernel(LstmScanArgs A) {
    __builtin_amdgcn_s_setprio(3);       // latency-bound chain: win issue arbitration against co-resident GEMM waves
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Wl = smem;                                   // [UT][KSP/4][64][4]
    const size_t wfl = STREAM ? 0 : (size_t)UT * A.KSP * 64;                       // floats of resident weights
    float* red = smem + wfl + (threadIdx.x >> 8) * UT * 1024;                      // per team [UT][4 waves][64][4]
    int* flag = (int*)(smem + wfl + NC * UT * 1024);

    TEAM_IDS_(NC)
    const int q = lane >> 4, bl = lane & 15;
    const int b0 = chunk * 16;
    const int b = b0 + bl;
    const bool bvalid = b < A.B;
    const int N = A.N, KS = A.KS, KSP = A.KSP, T = A.T, B = A.B;

    GroupSync gs;
    group_sync_init(gs, A.sync + (size_t)sgrp * SYNC_STRIDE, A.G, j, flag);

    // resident weights
    const float* const Wg = D.wpack + (size_t)j * UT * KSP * 64;
    if constexpr (!STREAM) {
        const f32x4* src = (const f32x4*)Wg;
        f32x4* dst = (f32x4*)Wl;
        for (int i = threadIdx.x; i < UT * KSP * 16; i += 256 * NC) dst[i] = src[i];
    }
    float* rex = A.rex + (size_t)grp * 2 * KSP * 64;
    float* sta = A.sta ? A.sta + (size_t)grp * 2 * KS * 128 : nullptr;
    float* stc = A.stc ? A.stc + (size_t)grp * 2 * KS * 32 : nullptr;
    const __amdgpu_buffer_rsrc_t rex_rs = make_rsrc_(rex, (unsigned)(2 * KSP * 64 * 4));
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    // exchange layout [tile][batch][unit]: a lane's B operands of one group of 4 MFMA steps are ONE
    // 16-byte sc1 load, a tile's r is one 16-byte sc1 store per batch row
    const unsigned ex_voff = (unsigned)((q * 64 + bl * 4) * 4);

    // epilogue-wave state: wave ut < UT owns tile (j*UT+ut); lane (q,bl) owns unit q of it for row bl
    const int tile = j * UT + (wave < UT ? wave : 0);
    const int unit = tile * 4 + q;
    const int col4 = unit * 4;
    float c = 0.f, r = 0.f;
    if (wave < UT) {
        if (bvalid && D.c0) c = D.c0[(size_t)b * N + unit];
        if (bvalid && D.r0) r = D.r0[(size_t)b * N + unit];
        const float r1 = __shfl(r, bl + 16), r2 = __shfl(r, bl + 32), r3 = __shfl(r, bl + 48);
        if (q == 0) st_sc1_f4(rex_rs, (unsigned)(((size_t)tile * 64 + bl * 4) * 4), f32x4{r, r1, r2, r3});   // parity 0 holds r_{-1}
    }
    if (j == 0)                                          // padded tiles (zero weights) must not hold NaNs
        for (int i = tid; i < 2 * (KSP - KS) * 16; i += 256) {
            const int pr = i / ((KSP - KS) * 16), o = i % ((KSP - KS) * 16);
            st_sc1_f4(rex_rs, (unsigned)((((size_t)pr * KSP + KS) * 16 + o) * 16), f32x4{0, 0, 0, 0});
        }
    float ac_u = 1.f, bc_u = 0.f;
    if (A.cell_ln) { ac_u = D.alpha_c[unit]; bc_u = D.beta_c[unit]; }
    // LN gain / shift of this lane's 4 gate columns: loop-invariant, read once (per sample when ln_kind == 2)
    f32x4 al4 = {1.f, 1.f, 1.f, 1.f}, be4 = {0.f, 0.f, 0.f, 0.f};
    if (A.ln_kind && wave < UT) {
        al4 = *(const f32x4*)(D.alpha_hid + (A.ln_kind == 2 ? (size_t)(bvalid ? b : 0) * 4 * N : 0) + col4);
        if (D.beta_hid) be4 = *(const f32x4*)(D.beta_hid + col4);
    }
    f32x4 pe4 = {0.f, 0.f, 0.f, 0.f}, a0v = {0.f, 0.f, 0.f, 0.f};
    bool virgin = true;                                  // OPT: no unmasked step of this row yet
    if constexpr (OPT) {
        if (wave < UT && D.peep) pe4 = *(const f32x4*)(D.peep + col4);
        if (wave < UT && D.a0) a0v = *(const f32x4*)(D.a0 + (size_t)(bvalid ? b : (B - 1)) * 4 * N + col4);
    }
    group_publish(gs);

    const int s4_per_wave = KSP / 16;                    // groups of 4 MFMA k-steps per wave
    // inputs of the step, loaded one step ahead through walking per-lane pointers; branch-free (every
    // wave issues the loads, non-owners read a dummy line), because hipcc waits for a load at the first
    // control-flow join behind it
    const bool own = wv < UT;
    const int bq = bvalid ? b : (B - 1);
    const float* const dummy = D.wpack;
    const int dt = D.reverse ? -1 : 1;
    const int t0 = D.reverse ? T - 1 : 0;
    const float* p_x = own ? D.xin + ((size_t)t0 * B + bq) * 4 * N + col4 : dummy;
    const float* p_m = own ? A.mask + (size_t)bq * A.msb + (size_t)t0 * A.mst : dummy;
    const int s_x = own ? dt * B * 4 * N : 0, s_m = own ? dt * A.mst : 0;
    f32x4 n_x4 = *(const f32x4*)p_x;
    float n_m = *p_m;
    // the saves walk per-lane pointers too: addresses formed from the descriptor inside the loop reload its
    // fields through scalar loads that are waited for one by one
    const size_t rw0 = (size_t)t0 * B + bq;
    const bool sv_ln = A.ln_kind && D.ahat, sv_cl = A.cell_ln && D.chat;
    const int rsw = A.per_gate ? 4 : 1;
    float* q_r = D.r_out + rw0 * D.ld_r + unit;
    float* q_c = D.cst + rw0 * N + unit;
    float* q_g = D.gates ? D.gates + rw0 * 4 * N + col4 : nullptr;
    float* q_ah = sv_ln ? D.ahat + rw0 * 4 * N + col4 : nullptr;
    float* q_ra = sv_ln ? D.rs_a + rw0 * rsw : nullptr;
    float* q_ch = sv_cl ? D.chat + rw0 * N + unit : nullptr;
    float* q_rc = sv_cl ? D.rs_c + rw0 : nullptr;
    const int z_r = dt * B * D.ld_r, z_c = dt * B * N, z_g = q_g ? dt * B * 4 * N : 0, z_ah = sv_ln ? dt * B * 4 * N : 0;
    const int z_ra = sv_ln ? dt * B * rsw : 0, z_ch = sv_cl ? dt * B * N : 0, z_rc = sv_cl ? dt * B : 0;
    const bool lead = tile == 0 && q == 0;
    STAMP_DECL
    for (int step = 0; step < T; ++step) {
        const int t = D.reverse ? T - 1 - step : step;
        const int par = step & 1;
        const size_t row = (size_t)t * B + b;
        const f32x4 x4 = n_x4;
        const float m = (own && bvalid) ? n_m : 0.f;
        {
            const bool go = step + 1 < T;
            p_x += go ? s_x : 0; p_m += go ? s_m : 0;
            n_x4 = *(const f32x4*)p_x; n_m = *p_m;
        }
        if (!group_wait(gs)) break;
        STAMP(0);

        // a^T tile partials over this wave's K quarter; B operand straight from the exchange buffer
        f32x4 acc0[UT], acc1[UT];
#pragma unroll
        for (int u = 0; u < UT; ++u) { acc0[u] = f32x4{0, 0, 0, 0}; acc1[u] = f32x4{0, 0, 0, 0}; }
        // request a whole chunk of B operands before its first MFMA: the exchange-buffer latency is
        // paid once per chunk instead of once per k-step
        for (int c0 = 0; c0 < s4_per_wave; c0 += 8) {
            f32x4 bv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)    // tiles >= KS carry zero weights (and zeroed payload)
                bv[i] = ldb_sc1_f4(rex_rs, ex_voff, (unsigned)(par * KSP * 256) + (unsigned)(wv * s4_per_wave + c0 + i) * 1024u);
            f32x4 avs[STREAM ? 8 : 1][UT];                 // STREAM: the chunk's A fragments, requested together as well
            if constexpr (STREAM) {
#pragma unroll
                for (int g4 = 0; g4 < 8; ++g4) {
                    const int s4 = wave * s4_per_wave + (c0 + g4 < s4_per_wave ? c0 + g4 : s4_per_wave - 1);
#pragma unroll
                    for (int u = 0; u < UT; ++u)
                        avs[g4][u] = *(const f32x4*)(Wg + (((size_t)u * (KSP / 4) + s4) * 64 + lane) * 4);
                }
            }
#pragma unroll
            for (int g4 = 0; g4 < 8; ++g4) {
                if (c0 + g4 < s4_per_wave) {
                    const int s4 = wave * s4_per_wave + c0 + g4;
#pragma unroll
                    for (int u = 0; u < UT; ++u) {
                        f32x4 av;
                        if constexpr (STREAM) av = avs[g4][u];
                        else av = *(const f32x4*)(Wl + (((size_t)u * (KSP / 4) + s4) * 64 + lane) * 4);
                        acc0[u] = mfma4(av[0], bv[g4][0], acc0[u]);
                        acc1[u] = mfma4(av[1], bv[g4][1], acc1[u]);
                        acc0[u] = mfma4(av[2], bv[g4][2], acc0[u]);
                        acc1[u] = mfma4(av[3], bv[g4][3], acc1[u]);
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UT; ++u)
            *(f32x4*)(red + ((size_t)(u * 4 + wave) * 64 + lane) * 4) = acc0[u] + acc1[u];
        __syncthreads();
        STAMP(1);

        f32x4 a4 = {0, 0, 0, 0};
        if (wave < UT) {
#pragma unroll
            for (int w = 0; w < 4; ++w) a4 += *(const f32x4*)(red + ((size_t)(wave * 4 + w) * 64 + lane) * 4);
            if constexpr (OPT) {
                if (D.a0 && (step == 0 || (D.a0_until_valid && virgin))) a4 = a0v;
            } else {
                if (step == 0 && D.a0) a4 = *(const f32x4*)(D.a0 + (size_t)bq * 4 * N + col4);
            }
        }
        float rs_a = 0.f;
        f32x4 rs_a4 = {0, 0, 0, 0};
        f32x4 ah4 = {0, 0, 0, 0};
        if (A.ln_kind && !A.per_gate) {
            if (wave < UT) {       // per-tile mean / M2 over its 16 columns, per batch row
                const float mj = sum_q(a4[0] + a4[1] + a4[2] + a4[3]) * (1.0f / 16.0f);
                f32x4 dlt = a4 - mj;
                const float M2 = sum_q(dlt[0] * dlt[0] + dlt[1] * dlt[1] + dlt[2] * dlt[2] + dlt[3] * dlt[3]);
                if (q == 0) st_sc1_f2(sta + ((size_t)par * KS + tile) * 128 + bl * 2, mj, M2);
            }
            group_publish(gs);
            if (!group_wait(gs)) break;
            STAMP(2);
            float mean;
            combine_stats(sta + (size_t)par * KS * 128, KS, tid, 16.0f, mean, rs_a, red, 128);
            ah4 = (a4 - mean) * rs_a;
        } else if (A.ln_kind) {    // gate_ln: each gate's N columns are normalised on their own;
            if (wave < UT) {       // per tile and gate: mean / M2 over the tile's 4 units
                f32x4 mj4, M24;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    mj4[g] = sum_q(a4[g]) * 0.25f;
                    const float dl = a4[g] - mj4[g];
                    M24[g] = sum_q(dl * dl);
                }
                const float mjq = q == 0 ? mj4[0] : q == 1 ? mj4[1] : q == 2 ? mj4[2] : mj4[3];
                const float M2q = q == 0 ? M24[0] : q == 1 ? M24[1] : q == 2 ? M24[2] : M24[3];
                st_sc1_f2(sta + (((size_t)par * KS + tile) * 4 + q) * 32 + bl * 2, mjq, M2q);   // lane q: gate q
            }
            group_publish(gs);
            if (!group_wait(gs)) break;
            {
                f32x4 mean4;
                combine_stats4(sta + (size_t)par * KS * 128, KS, tid, 4.0f, mean4, rs_a4, red);
                ah4 = (a4 - mean4) * rs_a4;
            }
        }
        float chat_v = 0.f, rs_c = 0.f, c_new = 0.f, ig = 0.f, fg = 0.f, gg = 0.f, og = 0.f;
        if (wave < UT) {
            f32x4 pre;
            if (A.ln_kind) {
                pre = x4 + al4 * ah4 + be4;
            } else {
                pre = x4 + a4;
            }
            if constexpr (OPT) {     // peepholes: in / forget gate see c_{t-1}, the out gate the new cell (:190-192,:212-213)
                ig = sigmoidf_(pre[0] + c * pe4[0]); fg = sigmoidf_(pre[1] + c * pe4[1]); gg = tanhf_(pre[2]);
                c_new = fg * c + ig * gg;
                og = sigmoidf_(pre[3] + c_new * pe4[3]);
            } else {
                ig = sigmoidf_(pre[0]); fg = sigmoidf_(pre[1]); gg = tanhf_(pre[2]); og = sigmoidf_(pre[3]);
                c_new = fg * c + ig * gg;
            }
        }
        float outcell = c_new;
        STAMP(3);
        if (A.cell_ln) {
            if (wave < UT) {       // per-tile stats over its 4 units
                const float mj = sum_q(c_new) * 0.25f;
                const float dl = c_new - mj;
                const float M2 = sum_q(dl * dl);
                if (q == 0) st_sc1_f2(stc + ((size_t)par * KS + tile) * 32 + bl * 2, mj, M2);
            }
            group_publish(gs);
            if (!group_wait(gs)) break;
            STAMP(4);
            {
                float mean;
                combine_stats(stc + (size_t)par * KS * 32, KS, tid, 4.0f, mean, rs_c, red);
                chat_v = (c_new - mean) * rs_c;
                outcell = ac_u * chat_v + bc_u;
            }
        }
        if (wave < UT) {
            const float r_new = og * tanhf_(outcell);
            if (m != 0.f) { c = c_new; r = r_new; virgin = false; }
            {
                const float r1 = __shfl(r, bl + 16), r2 = __shfl(r, bl + 32), r3 = __shfl(r, bl + 48);
                if (q == 0)
                    st_sc1_f4(rex_rs, (unsigned)((((size_t)(par ^ 1) * KSP + tile) * 64 + bl * 4) * 4), f32x4{r, r1, r2, r3});
            }
        }
        STAMP(5);
        group_publish(gs);
        STAMP(6);
        if (wave < UT) {                                 // bulk saves behind the publish: off the critical path
            if (bvalid) {
                *q_r = r;
                *q_c = c;
                if (q_g) *(f32x4*)q_g = f32x4{ig, fg, gg, og};
                if (q_ah) {
                    *(f32x4*)q_ah = ah4;
                    if (lead) {
                        if (A.per_gate) *(f32x4*)q_ra = rs_a4;
                        else *q_ra = rs_a;
                    }
                }
                if (q_ch) {
                    *q_ch = chat_v;
                    if (lead) *q_rc = rs_c;
                }
            }
            q_r += z_r; q_c += z_c; q_g += z_g; q_ah += z_ah; q_ra += z_ra; q_ch += z_ch; q_rc += z_rc;
        }
    }
    STAMP_DUMP(A.dbg, (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
}

// ------------------------------------------------------------------------------------------------
// forward, cell-LN variants with one exchange fewer per step (UT = 1, N <= 512).  Instead of r_t the
// workgroups all-gather (c_new, o) of their units; every workgroup then forms the cell-LN statistics of its
// 16 batch rows over all N units itself and r_t = o * tanh(alpha_c * chat + beta_c) for all units, in the
// registers that feed the MFMA B operand (each lane keeps r_{t-1} of its 32 (unit, row) pairs for the masked
// carry).  The owner's saves that need the statistics (r_out, chat, rs_c of step t) happen at the top of
// step t+1; one extra wait after the loop flushes the last step.  Same arithmetic as the generic kernel
// except the order of the sums inside the cell-LN statistics.
// ------------------------------------------------------------------------------------------------
// 32 tanh per lane and step sit on the critical path of the kernel below; the saved r_out uses the same form,
// so state and saves agree bit for bit.  Since round 3 this is the library's accurate tanhf_ (common.h); the
// round-2 form (hardware reciprocal of __expf) is kept for the ASR_ACT=0 comparison build only.
__device__ __forceinline__ float tanh_rcp_(float x) {
#if ASR_ACT == 0
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * x) + 1.0f);
#else
    return tanhf_(x);
#endif
}
template <int NC>
__global__ __launch_bounds__(256 * NC) void lstm_scan_fwd_cln_kernel(LstmScanArgs A) {
    __builtin_amdgcn_s_setprio(3);       // latency-bound chain: win issue arbitration against co-resident GEMM waves
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Wl = smem;                                   // [KSP/4][64][4]
    float* red = Wl + (size_t)A.KSP * 64 + (threadIdx.x >> 8) * 1024;     // per team [4 waves][64][4]
    int* flag = (int*)(Wl + (size_t)A.KSP * 64 + NC * 1024);
    float* scr = (float*)flag + 16 + (threadIdx.x >> 8) * 256;            // per team [2][4 waves][16 rows] statistics partials

    TEAM_IDS_(NC)
    const int q = lane >> 4, bl = lane & 15;
    const int b = chunk * 16 + bl;
    const bool bvalid = b < A.B;
    const int N = A.N, KS = A.KS, KSP = A.KSP, T = A.T, B = A.B;

    GroupSync gs;
    group_sync_init(gs, A.sync + (size_t)sgrp * SYNC_STRIDE, A.G, j, flag);
    {
        const f32x4* src = (const f32x4*)(D.wpack + (size_t)j * KSP * 64);
        f32x4* dst = (f32x4*)Wl;
        for (int i = threadIdx.x; i < KSP * 16; i += 256 * NC) dst[i] = src[i];
    }
    // exchange layout [parity][tile][c | o][batch][unit]
    float* rex = A.rex + (size_t)grp * 4 * KSP * 64;
    float* sta = A.sta ? A.sta + (size_t)grp * 2 * KS * 128 : nullptr;
    const __amdgpu_buffer_rsrc_t rex_rs = make_rsrc_(rex, (unsigned)(4 * KSP * 64 * 4));
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const unsigned ex_voff = (unsigned)((q * 128 + bl * 4) * 4);
    const int s4pw = KSP / 16;                           // <= 8: groups of 4 MFMA k-steps per wave

    const bool own = wv == 0;
    const int tile = j, unit = tile * 4 + q, col4 = unit * 4;
    float c = 0.f, r = 0.f;
    if (own) {
        if (bvalid && D.c0) c = D.c0[(size_t)b * N + unit];
        if (bvalid && D.r0) r = D.r0[(size_t)b * N + unit];
        const float r1 = __shfl(r, bl + 16), r2 = __shfl(r, bl + 32), r3 = __shfl(r, bl + 48);
        if (q == 0) st_sc1_f4(rex_rs, (unsigned)(((size_t)tile * 128 + bl * 4) * 4), f32x4{r, r1, r2, r3});   // r_{-1} in the c slot
    }
    const float ac_u = D.alpha_c[unit], bc_u = D.beta_c[unit];
    f32x4 al4 = {1.f, 1.f, 1.f, 1.f}, be4 = {0.f, 0.f, 0.f, 0.f};
    if (A.ln_kind && own) {
        al4 = *(const f32x4*)(D.alpha_hid + (A.ln_kind == 2 ? (size_t)(bvalid ? b : 0) * 4 * N : 0) + col4);
        if (D.beta_hid) be4 = *(const f32x4*)(D.beta_hid + col4);
    }
    // cell-LN gain / shift of the 32 units this lane feeds to the MFMA: unit 4 * (4 * (wv * s4pw + i) + q) + e
    f32x4 acv[8], bcv[8], rp[8];
    bool tv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int tl = 4 * (wv * s4pw + i) + q;
        tv[i] = i < s4pw && tl < KS;
        const int tq = tv[i] ? tl : 0;
        acv[i] = *(const f32x4*)(D.alpha_c + tq * 4);
        bcv[i] = *(const f32x4*)(D.beta_c + tq * 4);
        rp[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    group_publish(gs);

    const int bq = bvalid ? b : (B - 1);
    const float* const dummy = D.wpack;
    const int dt = D.reverse ? -1 : 1;
    const int t0 = D.reverse ? T - 1 : 0;
    const float* p_x = own ? D.xin + ((size_t)t0 * B + bq) * 4 * N + col4 : dummy;
    const float* p_m = A.mask + (size_t)bq * A.msb + (size_t)t0 * A.mst;      // every lane: row bl's mask
    const int s_x = own ? dt * B * 4 * N : 0, s_m = dt * A.mst;
    f32x4 n_x4 = *(const f32x4*)p_x;
    float n_m = *p_m;
    const size_t rw0 = (size_t)t0 * B + bq;
    const bool sv_ln = A.ln_kind && D.ahat, sv_cl = D.chat != nullptr;
    const int rsw = A.per_gate ? 4 : 1;
    float* q_r = D.r_out + rw0 * D.ld_r + unit;
    float* q_c = D.cst + rw0 * N + unit;
    float* q_g = D.gates ? D.gates + rw0 * 4 * N + col4 : nullptr;
    float* q_ah = sv_ln ? D.ahat + rw0 * 4 * N + col4 : nullptr;
    float* q_ra = sv_ln ? D.rs_a + rw0 * rsw : nullptr;
    float* q_ch = sv_cl ? D.chat + rw0 * N + unit : nullptr;
    float* q_rc = sv_cl ? D.rs_c + rw0 : nullptr;
    const int z_r = dt * B * D.ld_r, z_c = dt * B * N, z_g = q_g ? dt * B * 4 * N : 0, z_ah = sv_ln ? dt * B * 4 * N : 0;
    const int z_ra = sv_ln ? dt * B * rsw : 0, z_ch = sv_cl ? dt * B * N : 0, z_rc = sv_cl ? dt * B : 0;
    const bool lead = tile == 0 && q == 0;
    float m_prev = 0.f, c_sv = 0.f, o_sv = 0.f;         // the step whose r is still to be formed
    for (int step = 0; step <= T; ++step) {
        const int par = step & 1;
        const f32x4 x4 = n_x4;
        const float m = bvalid ? n_m : 0.f;
        {
            const bool go = step + 1 < T;
            p_x += go ? s_x : 0; p_m += go ? s_m : 0;
            n_x4 = *(const f32x4*)p_x; n_m = *p_m;
        }
        if (!group_wait(gs)) break;
        f32x4 cv[8], ov[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const unsigned so = (unsigned)(par * KSP * 512) + (unsigned)(wv * s4pw + i) * 2048u;
            cv[i] = ldb_sc1_f4(rex_rs, ex_voff, so);
            ov[i] = ldb_sc1_f4(rex_rs, ex_voff + 256u, so);
        }
        if (step == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) rp[i] = tv[i] ? cv[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
            float s1 = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) s1 += tv[i] ? (cv[i][0] + cv[i][1]) + (cv[i][2] + cv[i][3]) : 0.f;
            s1 = sum_q(s1);
            if (q == 0) scr[wave * 16 + bl] = s1;
            __syncthreads();
            const float mean = ((scr[bl] + scr[16 + bl]) + (scr[32 + bl] + scr[48 + bl])) / N;
            float s2 = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const f32x4 dl = cv[i] - mean;
                s2 += tv[i] ? (dl[0] * dl[0] + dl[1] * dl[1]) + (dl[2] * dl[2] + dl[3] * dl[3]) : 0.f;
            }
            s2 = sum_q(s2);
            if (q == 0) scr[64 + wave * 16 + bl] = s2;
            __syncthreads();
            const float rstd = 1.0f / sqrtf(((scr[64 + bl] + scr[80 + bl]) + (scr[96 + bl] + scr[112 + bl])) / N + ASR_EPS);
            const bool onp = m_prev != 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float rn = ov[i][e] * tanh_rcp_(acv[i][e] * ((cv[i][e] - mean) * rstd) + bcv[i][e]);
                    rp[i][e] = (onp && tv[i]) ? rn : rp[i][e];
                }
            }
            if (own) {                                   // the saves of the step before that needed the statistics
                const float chat_v = (c_sv - mean) * rstd;
                const float r_new = o_sv * tanh_rcp_(ac_u * chat_v + bc_u);
                if (onp) r = r_new;
                if (bvalid) {
                    *q_r = r;
                    if (q_ch) {
                        *q_ch = chat_v;
                        if (lead) *q_rc = rstd;
                    }
                }
                q_r += z_r; q_ch += z_ch; q_rc += z_rc;
            }
        }
        if (step == T) break;

        f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
        for (int g4 = 0; g4 < 8; ++g4) {
            if (g4 < s4pw) {
                const f32x4 av = *(const f32x4*)(Wl + (((size_t)(wave * s4pw + g4)) * 64 + lane) * 4);
                acc0 = mfma4(av[0], rp[g4][0], acc0);
                acc1 = mfma4(av[1], rp[g4][1], acc1);
                acc0 = mfma4(av[2], rp[g4][2], acc0);
                acc1 = mfma4(av[3], rp[g4][3], acc1);
            }
        }
        *(f32x4*)(red + ((size_t)wave * 64 + lane) * 4) = acc0 + acc1;
        __syncthreads();
        f32x4 a4 = {0, 0, 0, 0};
        if (own) {
#pragma unroll
            for (int w = 0; w < 4; ++w) a4 += *(const f32x4*)(red + ((size_t)w * 64 + lane) * 4);
        }
        float rs_a = 0.f;
        f32x4 rs_a4 = {0, 0, 0, 0}, ah4 = {0, 0, 0, 0}, sv_g4 = {0, 0, 0, 0};
        if (A.ln_kind && !A.per_gate) {
            if (own) {
                const float mj = sum_q(a4[0] + a4[1] + a4[2] + a4[3]) * (1.0f / 16.0f);
                f32x4 dlt = a4 - mj;
                const float M2 = sum_q(dlt[0] * dlt[0] + dlt[1] * dlt[1] + dlt[2] * dlt[2] + dlt[3] * dlt[3]);
                if (q == 0) st_sc1_f2(sta + ((size_t)par * KS + tile) * 128 + bl * 2, mj, M2);
            }
            group_publish(gs);
            if (!group_wait(gs)) break;
            float mean;
            combine_stats(sta + (size_t)par * KS * 128, KS, tid, 16.0f, mean, rs_a, red, 128);
            ah4 = (a4 - mean) * rs_a;
        } else if (A.ln_kind) {
            if (own) {
                f32x4 mj4, M24;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    mj4[g] = sum_q(a4[g]) * 0.25f;
                    const float dl = a4[g] - mj4[g];
                    M24[g] = sum_q(dl * dl);
                }
                const float mjq = q == 0 ? mj4[0] : q == 1 ? mj4[1] : q == 2 ? mj4[2] : mj4[3];
                const float M2q = q == 0 ? M24[0] : q == 1 ? M24[1] : q == 2 ? M24[2] : M24[3];
                st_sc1_f2(sta + (((size_t)par * KS + tile) * 4 + q) * 32 + bl * 2, mjq, M2q);
            }
            group_publish(gs);
            if (!group_wait(gs)) break;
            f32x4 mean4;
            combine_stats4(sta + (size_t)par * KS * 128, KS, tid, 4.0f, mean4, rs_a4, red);
            ah4 = (a4 - mean4) * rs_a4;
        }
        if (own) {
            const f32x4 pre = A.ln_kind ? x4 + al4 * ah4 + be4 : x4 + a4;
            const float ig = sigmoidf_(pre[0]), fg = sigmoidf_(pre[1]), gg = tanhf_(pre[2]), og = sigmoidf_(pre[3]);
            const float c_new = fg * c + ig * gg;
            if (m != 0.f) c = c_new;
            c_sv = c_new; o_sv = og;
            {
                const float c1 = __shfl(c_new, bl + 16), c2 = __shfl(c_new, bl + 32), c3 = __shfl(c_new, bl + 48);
                const float o1 = __shfl(og, bl + 16), o2 = __shfl(og, bl + 32), o3 = __shfl(og, bl + 48);
                if (q == 0) {
                    const unsigned base = (unsigned)((((size_t)(par ^ 1) * KSP + tile) * 128 + bl * 4) * 4);
                    st_sc1_f4(rex_rs, base, f32x4{c_new, c1, c2, c3});
                    st_sc1_f4(rex_rs, base + 256u, f32x4{og, o1, o2, o3});
                }
            }
            sv_g4 = f32x4{ig, fg, gg, og};
        }
        m_prev = m;
        group_publish(gs);
        if (own) {                                       // bulk saves behind the publish: off the critical path
            if (bvalid) {
                *q_c = c;
                if (q_g) *(f32x4*)q_g = sv_g4;
                if (q_ah) {
                    *(f32x4*)q_ah = ah4;
                    if (lead) {
                        if (A.per_gate) *(f32x4*)q_ra = rs_a4;
                        else *q_ra = rs_a;
                    }
                }
            }
            q_c += z_c; q_g += z_g; q_ah += z_ah; q_ra += z_ra;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward (BPTT).  Walks the forward iteration order in reverse.  Per step and workgroup:
//   1. sum the partial dr_{t} contributions every workgroup published in the previous step
//   2. cell / gate gradients for its units (+ LN backward with two small stat exchanges)
//   3. partial dr_{t-1}[b, all k] = da[b, own cols] . W'[k, own cols]^T  -> publish per destination
// ------------------------------------------------------------------------------------------------
template <int UT, int NC, bool OPT = false, bool STREAM = false>
__global__ __launch_bounds__(256 * NC) void lstm_scan_bwd_kernel(LstmScanArgs A) {
    __builtin_amdgcn_s_setprio(3);       // latency-bound chain: win issue arbitration against co-resident GEMM waves
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Wb = smem;                                   // [UT][KT][64][4]
    const size_t wfl = STREAM ? 0 : (size_t)UT * A.KT * 256;
    float* dal = smem + wfl + (threadIdx.x >> 8) * UT * 1280;                      // per team [UT][16 cols][16 b]
    float* rsum = dal + UT * 256;                       // [UT][16 sg][16 b][4]
    int* flag = (int*)(smem + wfl + NC * UT * 1280);

    TEAM_IDS_(NC)
    const int q = lane >> 4, bl = lane & 15;
    const int b = chunk * 16 + bl;
    const bool bvalid = b < A.B;
    const int N = A.N, KS = A.KS, KT = A.KT, T = A.T, B = A.B, G = A.G;

    GroupSync gs;
    group_sync_init(gs, A.sync + (size_t)sgrp * SYNC_STRIDE, A.G, j, flag);
    const float* const Wg = D.wpack + (size_t)j * UT * KT * 256;
    if constexpr (!STREAM) {
        const f32x4* src = (const f32x4*)Wg;
        f32x4* dst = (f32x4*)Wb;
        for (int i = threadIdx.x; i < UT * KT * 64; i += 256 * NC) dst[i] = src[i];
    }
    float* dex = A.dex + (size_t)grp * 2 * KS * G * 64;
    float* sta = A.sta ? A.sta + (size_t)grp * 2 * KS * 128 : nullptr;
    float* stc = A.stc ? A.stc + (size_t)grp * 2 * KS * 32 : nullptr;
    const __amdgpu_buffer_rsrc_t dex_rs = make_rsrc_(dex, (unsigned)((size_t)2 * KS * G * 64 * 4));

    const int tile = j * UT + (wave < UT ? wave : 0);
    const int unit = tile * 4 + q;
    const int col4 = unit * 4;
    float ac_u = 1.f, bc_u = 0.f;
    if (A.cell_ln) { ac_u = D.alpha_c[unit]; bc_u = D.beta_c[unit]; }
    float dc = 0.f, dr_pass = 0.f;
    f32x4 al4 = {1.f, 1.f, 1.f, 1.f};                  // LN gain of this lane's 4 gate columns (loop-invariant)
    if (A.ln_kind && wave < UT)
        al4 = *(const f32x4*)(D.alpha_hid + (A.ln_kind == 2 ? (size_t)(bvalid ? b : 0) * 4 * N : 0) + col4);
    __syncthreads();

    // Saved tensors of a step are loaded one step ahead through walking per-lane pointers, branch-free
    // (every wave issues every load; non-owners, absent tensors and invalid batch rows read a dummy
    // line / a clamped row): hipcc waits for a load at the first control-flow join behind it, which put
    // the HBM latency of these loads in front of every step.  The cell rows are read once, two steps
    // ahead (c_prev of backward step s is the cell row of step s+1).
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const bool own = wv < UT;
    const bool lnp = own && A.ln_kind, lnpg = lnp && A.per_gate, lnrow = lnp && !A.per_gate, cln = own && A.cell_ln;
    const int bq = bvalid ? b : (B - 1);
    const float* const dummy = D.wpack;
    const int dt = D.reverse ? 1 : -1;
    auto row_of = [&](int k) { k = k < T ? k : T - 1; return (size_t)(D.reverse ? k : T - 1 - k) * B + bq; };
    const size_t r00 = row_of(0);
    const float* p_m = own ? A.mask + (size_t)bq * A.msb + (size_t)(D.reverse ? 0 : T - 1) * A.mst : dummy;
    const float* p_g = own ? D.gates + r00 * 4 * N + col4 : dummy;
    const float* p_dr = own ? D.dr_ext + r00 * D.ld_dr + unit : dummy;
    const float* p_ah = lnp ? D.ahat + r00 * 4 * N + col4 : dummy;
    const float* p_ra4 = lnpg ? D.rs_a + r00 * 4 : dummy;
    const float* p_ra = lnrow ? D.rs_a + r00 : dummy;
    const float* p_ch = cln ? D.chat + r00 * N + unit : dummy;
    const float* p_rc = cln ? D.rs_c + r00 : dummy;
    const float* p_c2 = own ? D.cst + row_of(2) * N + unit : dummy;
    const int s_m = own ? dt * A.mst : 0, s_g = own ? dt * B * 4 * N : 0, s_dr = own ? dt * B * D.ld_dr : 0;
    const int s_ah = lnp ? dt * B * 4 * N : 0, s_ra4 = lnpg ? dt * B * 4 : 0, s_ra = lnrow ? dt * B : 0;
    const int s_ch = cln ? dt * B * N : 0, s_rc = cln ? dt * B : 0, s_c = own ? dt * B * N : 0;
    float c0v = *(own ? D.cst + row_of(0) * N + unit : dummy), c1v = *(own ? D.cst + row_of(1) * N + unit : dummy);
    const float c_init = *((own && D.c0) ? D.c0 + (size_t)bq * N + unit : dummy);
    const bool has_c0 = D.c0 != nullptr;
    // OPT: external cell gradient, peephole vectors, and what the learned initial rows / peepholes need at the end
    const float* p_dce = (OPT && own && D.dc_ext) ? D.dc_ext + r00 * D.ld_dc + unit : dummy;
    const int s_dce = (OPT && own && D.dc_ext) ? dt * B * D.ld_dc : 0;
    const bool has_dce = OPT && D.dc_ext;
    f32x4 pe4 = {0.f, 0.f, 0.f, 0.f}, pw = {0.f, 0.f, 0.f, 0.f}, da_first = {0.f, 0.f, 0.f, 0.f};
    if constexpr (OPT) { if (own && D.peep) pe4 = *(const f32x4*)(D.peep + col4); }
    float n_m, n_dre, n_ra, n_ch, n_rc, n_c2, n_dce = 0.f;
    f32x4 n_g4, n_ah4, n_ra4;
    auto prefetch = [&]() {
        if constexpr (OPT) n_dce = *p_dce;
        n_m = *p_m; n_g4 = *(const f32x4*)p_g; n_dre = *p_dr;
        n_ah4 = *(const f32x4*)p_ah; n_ra4 = *(const f32x4*)p_ra4; n_ra = *p_ra;
        n_ch = *p_ch; n_rc = *p_rc; n_c2 = *p_c2;
    };
    prefetch();
    // gradient saves through walking per-lane pointers as well (see the forward kernel)
    float* q_dg = D.dG + r00 * 4 * N + col4;
    float* q_da = (A.ln_kind && D.dA) ? D.dA + r00 * 4 * N + col4 : nullptr;
    float* q_do = (A.cell_ln && D.dOut) ? D.dOut + r00 * N + unit : nullptr;
    const int z_dg = dt * B * 4 * N, z_da = q_da ? dt * B * 4 * N : 0, z_do = q_do ? dt * B * N : 0;
    STAMP_DECL

    for (int step = 0; step < T; ++step) {
        const int st = T - 1 - step;                     // forward iteration index
        const int t = D.reverse ? T - 1 - st : st;
        const int par = step & 1;
        const size_t row = (size_t)t * B + b;
        const bool live = own && bvalid;
        const float m = live ? n_m : 0.f;
        const f32x4 g4 = n_g4, ah4 = n_ah4, rs_a4 = n_ra4;
        const float c_new = c0v, c_prev = st > 0 ? c1v : (has_c0 ? c_init : 0.f);
        const float dre = live ? n_dre : 0.f, rs_a = n_ra, chat_v = n_ch, rs_c = n_rc;
        if constexpr (OPT) { if (live && has_dce) dc += n_dce; }     // dLoss/d cell_t joins the carried cell gradient
        {
            const bool go = step + 1 < T;
            if constexpr (OPT) p_dce += go ? s_dce : 0;
            p_m += go ? s_m : 0; p_g += go ? s_g : 0; p_dr += go ? s_dr : 0; p_ah += go ? s_ah : 0;
            p_ra4 += go ? s_ra4 : 0; p_ra += go ? s_ra : 0; p_ch += go ? s_ch : 0; p_rc += go ? s_rc : 0;
            prefetch();
            p_c2 += (step + 3 < T) ? s_c : 0;
        }
        // 1. gather dr contributions published by the previous backward step
        float dr_sum = 0.f;
        if (step > 0) {
            if (!group_wait(gs)) break;
            STAMP(0);
            const int sg = tid >> 4, f4 = tid & 15;
#pragma unroll
            for (int u = 0; u < UT; ++u) {
                f32x4 acc = {0, 0, 0, 0};
                const size_t base = (((size_t)(par ^ 1) * KS + (j * UT + u)) * G) * 64;
                for (int c0 = 0; c0 < G; c0 += 128) {
                    f32x4 v[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        v[i] = ldb_sc1_f4(dex_rs, (unsigned)((sg * 64 + f4 * 4) * 4),
                                          (unsigned)(base * 4) + (unsigned)((c0 + i * 16) * 256));
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (sg + c0 + i * 16 >= G) v[i] = f32x4{0, 0, 0, 0};
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc += v[i];
                }
                *(f32x4*)(rsum + ((size_t)(u * 16 + sg) * 16 + f4) * 4) = acc;
            }
            __syncthreads();
            if (wave < UT) {
#pragma unroll
                for (int sgi = 0; sgi < 16; ++sgi) dr_sum += rsum[((size_t)(wave * 16 + sgi) * 16 + bl) * 4 + q];
            }
        }
        STAMP(1);
        // 2. cell / gate gradients
        const float dr_t = dre + dr_pass + dr_sum;
        const float ig = g4[0], fg = g4[1], gg = g4[2], og = g4[3];
        const bool on = (m != 0.f);
        float outcell = c_new;
        if (A.cell_ln) outcell = ac_u * chat_v + bc_u;
        const float tc = tanhf_(outcell);
        float do_ = 0.f, dout = 0.f;
        if (on) {
            do_ = dr_t * tc;
            dout = clipf_(dr_t * og * (1.f - tc * tc), A.clip);
        }
        float dcn = dout;
        if (A.cell_ln) {
            const float dxh = dout * ac_u;
            if (wave < UT) {
                const float s1 = sum_q(dxh), s2 = sum_q(dxh * chat_v);
                if (q == 0) st_sc1_f2(stc + ((size_t)par * KS + tile) * 32 + bl * 2, s1, s2);
            }
            group_publish(gs);
            // bulk saves go behind the publish (its vmcnt(0) drain would wait for them), under the exchange
            if (wave < UT && bvalid && q_do) *q_do = dout;
            if (!group_wait(gs)) break;
            STAMP(2);
            {
                float s1, s2;
                combine_sums(stc + (size_t)par * KS * 32, KS, tid, s1, s2, rsum);
                dcn = rs_c * (dxh - s1 / N - chat_v * (s2 / N));
            }
        }
        f32x4 dg4 = {0, 0, 0, 0};
        if (on) {
            float dct = dc + dcn;
            dg4[3] = clipf_(do_ * og * (1.f - og), A.clip);
            if constexpr (OPT) dct += dg4[3] * pe4[3];           // the out gate looked at the new cell
            dg4[0] = clipf_(dct * gg * ig * (1.f - ig), A.clip);
            dg4[1] = clipf_(dct * c_prev * fg * (1.f - fg), A.clip);
            dg4[2] = clipf_(dct * ig * (1.f - gg * gg), A.clip);
            dc = dct * fg;
            if constexpr (OPT) {
                dc += dg4[0] * pe4[0] + dg4[1] * pe4[1];
                pw[0] += dg4[0] * c_prev; pw[1] += dg4[1] * c_prev; pw[3] += dg4[3] * c_new;
            }
            dr_pass = 0.f;
        } else {
            dr_pass = dr_t;
        }
        f32x4 da4 = dg4;
        bool dg_pending = wave < UT && bvalid;
        STAMP(3);
        if (A.ln_kind) {
            f32x4 dah = {0, 0, 0, 0};
            if (wave < UT) {
                dah = dg4 * al4;
                if (!A.per_gate) {
                    const float s1 = sum_q(dah[0] + dah[1] + dah[2] + dah[3]);
                    const float s2 = sum_q(dah[0] * ah4[0] + dah[1] * ah4[1] + dah[2] * ah4[2] + dah[3] * ah4[3]);
                    if (q == 0) st_sc1_f2(sta + ((size_t)par * KS + tile) * 128 + bl * 2, s1, s2);
                } else {
                    f32x4 s14, s24;
#pragma unroll
                    for (int g = 0; g < 4; ++g) { s14[g] = sum_q(dah[g]); s24[g] = sum_q(dah[g] * ah4[g]); }
                    const float s1q = q == 0 ? s14[0] : q == 1 ? s14[1] : q == 2 ? s14[2] : s14[3];
                    const float s2q = q == 0 ? s24[0] : q == 1 ? s24[1] : q == 2 ? s24[2] : s24[3];
                    st_sc1_f2(sta + (((size_t)par * KS + tile) * 4 + q) * 32 + bl * 2, s1q, s2q);
                }
            }
            group_publish(gs);
            if (dg_pending) { *(f32x4*)q_dg = dg4; dg_pending = false; }
            if (!group_wait(gs)) break;
            STAMP(4);
            {
                if (!A.per_gate) {
                    float s1, s2;
                    combine_sums(sta + (size_t)par * KS * 128, KS, tid, s1, s2, rsum, 128);
                    const float m1 = s1 / (4 * N), m2 = s2 / (4 * N);
                    da4 = rs_a * (dah - m1 - ah4 * m2);
                } else {
                    f32x4 s14, s24;
                    combine_sums4(sta + (size_t)par * KS * 128, KS, tid, s14, s24, rsum);
                    da4 = rs_a4 * (dah - s14 * (1.0f / N) - ah4 * (s24 * (1.0f / N)));
                }
            }

        }
        if constexpr (OPT) { if (on) da_first = da4; }           // backward order: the last one kept is the row's first step
        // 3. partial dr_{prev}: B operand needs da[b][col 4e+(l>>4)] -> go through LDS
        if (wave < UT) {
#pragma unroll
            for (int k = 0; k < 4; ++k) dal[(size_t)wave * 256 + (q * 4 + k) * 16 + bl] = da4[k];
        }
        __syncthreads();
        STAMP(5);
        if (step + 1 < T) {
            if constexpr (STREAM) {                      // 4 destination tiles' fragments requested together
                for (int kt0 = wave; kt0 < KT; kt0 += 16) {
                    f32x4 avs[4][UT];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int kt = kt0 + 4 * i < KT ? kt0 + 4 * i : kt0;
#pragma unroll
                        for (int u = 0; u < UT; ++u) avs[i][u] = *(const f32x4*)(Wg + (((size_t)u * KT + kt) * 64 + lane) * 4);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int kt = kt0 + 4 * i;
                        if (kt >= KT) continue;
                        f32x4 acc = {0, 0, 0, 0};
#pragma unroll
                        for (int u = 0; u < UT; ++u)
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                acc = mfma4(avs[i][u][e], dal[(size_t)u * 256 + (4 * e + q) * 16 + bl], acc);
                        const int dst = kt * 4 + q;
                        if (dst < KS)
                            st_sc1_f4(dex_rs, (unsigned)(((((size_t)par * KS + dst) * G + j) * 64 + bl * 4) * 4), acc);
                    }
                }
            } else
            for (int kt = wave; kt < KT; kt += 4) {
                f32x4 acc = {0, 0, 0, 0};
#pragma unroll
                for (int u = 0; u < UT; ++u) {
                    const f32x4 av = *(const f32x4*)(Wb + (((size_t)u * KT + kt) * 64 + lane) * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc = mfma4(av[e], dal[(size_t)u * 256 + (4 * e + q) * 16 + bl], acc);
                }
                const int dst = kt * 4 + q;              // destination tile (units 4*dst .. 4*dst+3)
                if (dst < KS)
                    st_sc1_f4(dex_rs, (unsigned)(((((size_t)par * KS + dst) * G + j) * 64 + bl * 4) * 4), acc);
            }
            STAMP(6);
            group_publish(gs);
            STAMP(7);
        }
        if (dg_pending) *(f32x4*)q_dg = dg4;
        if (wave < UT && bvalid && q_da) *(f32x4*)q_da = da4;
        c0v = c1v; c1v = n_c2;
        q_dg += z_dg; q_da += z_da; q_do += z_do;
    }
    if constexpr (OPT) {
        if (wave < UT && bvalid) {
            if (D.dc0) D.dc0[(size_t)b * N + unit] = dc;
            if (D.dr0) D.dr0[(size_t)b * N + unit] = dr_pass;
            if (D.da0) *(f32x4*)(D.da0 + (size_t)b * 4 * N + col4) = da_first;
            if (D.dpeep) *(f32x4*)(D.dpeep + (size_t)b * 4 * N + col4) = pw;
        }
    }
    STAMP_DUMP(A.dbg, (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
}

// ------------------------------------------------------------------------------------------------
// backward, LN variants (row or per-gate statistics) at UT = 1 with G = N/4 a multiple of 32 and N <= 512:
// the dr_{t-1} = da . W'^T product as a 2-D decomposition instead of the all-to-all of per-destination
// partials (32 KB written and 32 KB read per workgroup and step).  Workgroup j = (unit block ub = j / 8,
// member kr = j % 8):
//   * with the gate-LN statistics exchange every workgroup also publishes dah = dg * alpha of its own 16
//     columns (1 KB); the statistics turn any column's dah into da = rstd * (dah - m1 - ahat * m2), ahat
//     being the saved forward tensor (read straight from HBM, one step ahead);
//   * member kr forms da for ITS eighth of the 4N columns (16 KB of dah gathered, 16 KB of ahat) and
//     multiplies it with the rows of W' of the block's 32 units: a (32 units x 16 batch) partial = 2 KB,
//     published with the step's last exchange;
//   * at the top of the next step a workgroup sums the 8 members' partials of its own 4 units (8 x 4 B per lane).
// MFMA k index q of step s in source tile i <-> column 16 i + 4 q + s, so a lane's operands are the 16-byte
// groups both tensors are stored in.  Same arithmetic as lstm_scan_bwd_kernel up to the order of the sums.
// ------------------------------------------------------------------------------------------------
#define L2D_R 8
template <int PG, int NC>                               // PG: per-gate LN statistics (compile time: the row-LN
__global__ __launch_bounds__(256 * NC) void lstm_scan_bwd_ln2d_kernel(LstmScanArgs A) {   // path keeps scalar statistics)
    __builtin_amdgcn_s_setprio(3);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int TW = A.KS / (4 * L2D_R);                  // source tiles per wave (1..4)
    float* Wb = smem;                                   // [4 waves][TW][2 unit tiles][64][4]
    float* red2 = Wb + (size_t)TW * 2048 + (threadIdx.x >> 8) * 2304;     // per team [4 waves][2][64][4]
    float* scr = red2 + 2048;                           // combine_sums scratch
    int* flag = (int*)(Wb + (size_t)TW * 2048 + NC * 2304);

    TEAM_IDS_(NC)
    const int q = lane >> 4, bl = lane & 15;
    const int b = chunk * 16 + bl;
    const bool bvalid = b < A.B;
    const int N = A.N, KS = A.KS, T = A.T, B = A.B, G = A.G;
    const int ub = j / L2D_R, kr = j % L2D_R;

    GroupSync gs;
    group_sync_init(gs, A.sync + (size_t)sgrp * SYNC_STRIDE, A.G, j, flag);
    {
        const f32x4* src = (const f32x4*)(D.wpack + (size_t)j * TW * 2048);
        f32x4* dst = (f32x4*)Wb;
        for (int i = threadIdx.x; i < TW * 512; i += 256 * NC) dst[i] = src[i];
    }
    // exchange buffers inside the (larger) dex region: dhx [2][KS tiles][16 b][16 cols], px [2][G][2][16 b][16 u]
    float* dhx = A.dex + (size_t)grp * 2 * KS * G * 64;
    float* px = dhx + (size_t)2 * KS * 256;
    float* sta = A.sta + (size_t)grp * 2 * KS * 128;
    float* stc = A.stc ? A.stc + (size_t)grp * 2 * KS * 32 : nullptr;
    const __amdgpu_buffer_rsrc_t dhx_rs = make_rsrc_(dhx, (unsigned)((size_t)2 * KS * 256 * 4));
    const __amdgpu_buffer_rsrc_t px_rs = make_rsrc_(px, (unsigned)((size_t)2 * G * 512 * 4));

    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const bool own = wv == 0;
    const int tile = j, unit = tile * 4 + q, col4 = unit * 4;
    float ac_u = 1.f, bc_u = 0.f;
    if (A.cell_ln) { ac_u = D.alpha_c[unit]; bc_u = D.beta_c[unit]; }
    float dc = 0.f, dr_pass = 0.f;
    f32x4 al4 = {1.f, 1.f, 1.f, 1.f};
    if (own) al4 = *(const f32x4*)(D.alpha_hid + (A.ln_kind == 2 ? (size_t)(bvalid ? b : 0) * 4 * N : 0) + col4);
    __syncthreads();

    const bool cln = own && A.cell_ln;
    const int bq = bvalid ? b : (B - 1);
    const float* const dummy = D.wpack;
    const int dt = D.reverse ? 1 : -1;
    auto row_of = [&](int k) { k = k < T ? k : T - 1; return (size_t)(D.reverse ? k : T - 1 - k) * B + bq; };
    const size_t r00 = row_of(0);
    const float* p_m = own ? A.mask + (size_t)bq * A.msb + (size_t)(D.reverse ? 0 : T - 1) * A.mst : dummy;
    const float* p_g = own ? D.gates + r00 * 4 * N + col4 : dummy;
    const float* p_dr = own ? D.dr_ext + r00 * D.ld_dr + unit : dummy;
    const float* p_ah = own ? D.ahat + r00 * 4 * N + col4 : dummy;
    const int rsw = PG ? 4 : 1;
    const float* p_ra = D.rs_a + r00 * rsw;                             // every lane: rstd of row bl (x4 per gate)
    const float* p_ch = cln ? D.chat + r00 * N + unit : dummy;
    const float* p_rc = cln ? D.rs_c + r00 : dummy;
    const float* p_c2 = own ? D.cst + row_of(2) * N + unit : dummy;
    // ahat of this member's column slice, row bl: source tile kr*(KS/8) + wave*TW + i, columns 4q..4q+3 of it
    const float* p_as = D.ahat + r00 * 4 * N + (size_t)(kr * (KS / L2D_R) + wv * TW) * 16 + 4 * q;
    const int s_m = own ? dt * A.mst : 0, s_g = own ? dt * B * 4 * N : 0, s_dr = own ? dt * B * D.ld_dr : 0;
    const int s_ah = own ? dt * B * 4 * N : 0, s_ra = dt * B * rsw, s_as = dt * B * 4 * N;
    const int s_ch = cln ? dt * B * N : 0, s_rc = cln ? dt * B : 0, s_c = own ? dt * B * N : 0;
    float c0v = *(own ? D.cst + row_of(0) * N + unit : dummy), c1v = *(own ? D.cst + row_of(1) * N + unit : dummy);
    const float c_init = *((own && D.c0) ? D.c0 + (size_t)bq * N + unit : dummy);
    const bool has_c0 = D.c0 != nullptr;
    float n_m, n_dre, n_ch, n_rc, n_c2;
    f32x4 n_g4, n_ah4, n_as[4], n_ra4;
    auto prefetch = [&]() {
        n_m = *p_m; n_g4 = *(const f32x4*)p_g; n_dre = *p_dr;
        n_ah4 = *(const f32x4*)p_ah;
        if (PG) n_ra4 = *(const f32x4*)p_ra;
        else { const float r1_ = *p_ra; n_ra4 = f32x4{r1_, r1_, r1_, r1_}; }
        n_ch = *p_ch; n_rc = *p_rc; n_c2 = *p_c2;
#pragma unroll
        for (int i = 0; i < 4; ++i) n_as[i] = *(const f32x4*)(p_as + (i < TW ? i : 0) * 16);
    };
    prefetch();
    float* q_dg = D.dG + r00 * 4 * N + col4;
    float* q_da = D.dA ? D.dA + r00 * 4 * N + col4 : nullptr;
    float* q_do = (A.cell_ln && D.dOut) ? D.dOut + r00 * N + unit : nullptr;
    const int z_dg = dt * B * 4 * N, z_da = q_da ? dt * B * 4 * N : 0, z_do = q_do ? dt * B * N : 0;
    // own units inside the block: unit tile kr >> 2, units 4*(kr&3) .. +3
    const unsigned px_own = (unsigned)((((kr >> 2) * 16 + bl) * 16 + 4 * (kr & 3) + q) * 4);

    for (int step = 0; step < T; ++step) {
        const int st = T - 1 - step;
        const int par = step & 1;
        const bool live = own && bvalid;
        const float m = live ? n_m : 0.f;
        const f32x4 g4 = n_g4, ah4 = n_ah4;
        f32x4 as4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) as4[i] = n_as[i];
        const float c_new = c0v, c_prev = st > 0 ? c1v : (has_c0 ? c_init : 0.f);
        const float dre = live ? n_dre : 0.f, chat_v = n_ch, rs_c = n_rc;
        const f32x4 rs_a = n_ra4;                        // per gate (row LN: the same value four times)
        {
            const bool go = step + 1 < T;
            p_m += go ? s_m : 0; p_g += go ? s_g : 0; p_dr += go ? s_dr : 0; p_ah += go ? s_ah : 0;
            p_ra += go ? s_ra : 0; p_as += go ? s_as : 0; p_ch += go ? s_ch : 0; p_rc += go ? s_rc : 0;
            prefetch();
            p_c2 += (step + 3 < T) ? s_c : 0;
        }
        // 1. the 8 members' partials of the own 4 units, published by the previous backward step
        float dr_sum = 0.f;
        if (step > 0) {
            if (!group_wait(gs)) break;
            float v[L2D_R];
#pragma unroll
            for (int s_ = 0; s_ < L2D_R; ++s_)
                v[s_] = ldb_sc1(px_rs, px_own, (unsigned)((((size_t)(par ^ 1) * G + ub * L2D_R + s_) * 512) * 4));
#pragma unroll
            for (int s_ = 0; s_ < L2D_R; ++s_) dr_sum += v[s_];
        }
        // 2. cell / gate gradients (as lstm_scan_bwd_kernel)
        const float dr_t = dre + dr_pass + dr_sum;
        const float ig = g4[0], fg = g4[1], gg = g4[2], og = g4[3];
        const bool on = (m != 0.f);
        float outcell = c_new;
        if (A.cell_ln) outcell = ac_u * chat_v + bc_u;
        const float tc = tanhf_(outcell);
        float do_ = 0.f, dout = 0.f;
        if (on) {
            do_ = dr_t * tc;
            dout = clipf_(dr_t * og * (1.f - tc * tc), A.clip);
        }
        float dcn = dout;
        if (A.cell_ln) {
            const float dxh = dout * ac_u;
            if (own) {
                const float s1 = sum_q(dxh), s2 = sum_q(dxh * chat_v);
                if (q == 0) st_sc1_f2(stc + ((size_t)par * KS + tile) * 32 + bl * 2, s1, s2);
            }
            group_publish(gs);
            if (own && bvalid && q_do) *q_do = dout;
            if (!group_wait(gs)) break;
            {
                float s1, s2;
                combine_sums(stc + (size_t)par * KS * 32, KS, tid, s1, s2, scr);
                dcn = rs_c * (dxh - s1 / N - chat_v * (s2 / N));
            }
        }
        f32x4 dg4 = {0, 0, 0, 0};
        if (on) {
            const float dct = dc + dcn;
            dg4[0] = clipf_(dct * gg * ig * (1.f - ig), A.clip);
            dg4[1] = clipf_(dct * c_prev * fg * (1.f - fg), A.clip);
            dg4[2] = clipf_(dct * ig * (1.f - gg * gg), A.clip);
            dg4[3] = clipf_(do_ * og * (1.f - og), A.clip);
            dc = dct * fg;
            dr_pass = 0.f;
        } else {
            dr_pass = dr_t;
        }
        // 3. gate-LN statistics + all-gather of dah
        const f32x4 dah = own ? dg4 * al4 : f32x4{0, 0, 0, 0};
        if (own) {
            if (!PG) {
                const float s1 = sum_q(dah[0] + dah[1] + dah[2] + dah[3]);
                const float s2 = sum_q(dah[0] * ah4[0] + dah[1] * ah4[1] + dah[2] * ah4[2] + dah[3] * ah4[3]);
                if (q == 0) st_sc1_f2(sta + ((size_t)par * KS + tile) * 128 + bl * 2, s1, s2);
            } else {
                f32x4 s14, s24;
#pragma unroll
                for (int g = 0; g < 4; ++g) { s14[g] = sum_q(dah[g]); s24[g] = sum_q(dah[g] * ah4[g]); }
                const float s1q = q == 0 ? s14[0] : q == 1 ? s14[1] : q == 2 ? s14[2] : s14[3];
                const float s2q = q == 0 ? s24[0] : q == 1 ? s24[1] : q == 2 ? s24[2] : s24[3];
                st_sc1_f2(sta + (((size_t)par * KS + tile) * 4 + q) * 32 + bl * 2, s1q, s2q);
            }
            st_sc1_f4(dhx_rs, (unsigned)((((size_t)par * KS + tile) * 256 + bl * 16 + q * 4) * 4), dah);
        }
        group_publish(gs);
        if (own && bvalid) *(f32x4*)q_dg = dg4;
        if (!group_wait(gs)) break;
        // the all-gathered dah of this member's column slice is requested BEFORE the statistics are combined: both are
        // fresh hand-off reads (one fabric round trip each) and do not depend on each other
        f32x4 dv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            dv[i] = ldb_sc1_f4(dhx_rs, (unsigned)((bl * 16 + q * 4) * 4),
                               (unsigned)((((size_t)par * KS + kr * (KS / L2D_R) + wv * TW + (i < TW ? i : 0)) * 256) * 4));
        f32x4 m1, m2;                                    // per gate = per column % 4 = per element of a lane's group
        if (!PG) {
            float s1, s2;
            combine_sums(sta + (size_t)par * KS * 128, KS, tid, s1, s2, scr, 128);
            const float a_ = s1 / (4 * N), b_ = s2 / (4 * N);
            m1 = f32x4{a_, a_, a_, a_}; m2 = f32x4{b_, b_, b_, b_};
        } else {
            f32x4 s14, s24;
            combine_sums4(sta + (size_t)par * KS * 128, KS, tid, s14, s24, scr);
            m1 = s14 * (1.0f / N); m2 = s24 * (1.0f / N);
        }
        // own columns (owner wave).  The value is pinned HERE: left alone, hipcc sinks the whole expression to the store
        // at the end of the step (after the third publish), and that build gave, about once per launch, a dA whose
        // ah4*m2 term was missing in gates 0 and 2 of one unit over all 16 batch rows (the low halves of the two
        // v_pk_fma_f32 ... op_sel:[0,1,0] it used there); every input of the expression was bit-stable.  The instruction
        // form alone is exact (tools/ubench/pk_opsel.hip, 5e10 results), so the cause is not established; with the value
        // computed here tools/soak_determinism.py finds no differing bit.
        f32x4 da4 = rs_a * (dah - m1 - ah4 * m2);
        asm volatile("" : "+v"(da4[0]), "+v"(da4[1]), "+v"(da4[2]), "+v"(da4[3]));
        // 4. partial dr_{t-1} of the block's 32 units over this member's column slice
        if (step + 1 < T) {
            f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (i < TW) {
                    const f32x4 dai = rs_a * (dv[i] - m1 - as4[i] * m2);
                    const f32x4 w0 = *(const f32x4*)(Wb + ((((size_t)wave * TW + i) * 2 + 0) * 64 + lane) * 4);
                    const f32x4 w1 = *(const f32x4*)(Wb + ((((size_t)wave * TW + i) * 2 + 1) * 64 + lane) * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc0 = mfma4(w0[e], dai[e], acc0);
                        acc1 = mfma4(w1[e], dai[e], acc1);
                    }
                }
            }
            *(f32x4*)(red2 + (((size_t)wave * 2 + 0) * 64 + lane) * 4) = acc0;
            *(f32x4*)(red2 + (((size_t)wave * 2 + 1) * 64 + lane) * 4) = acc1;
            __syncthreads();
            if (wave < 2) {                              // unit tile `wave`: lane (q, bl) = units 4q..4q+3, batch bl
                f32x4 v = *(const f32x4*)(red2 + (((size_t)0 * 2 + wave) * 64 + lane) * 4);
#pragma unroll
                for (int w = 1; w < 4; ++w) v += *(const f32x4*)(red2 + (((size_t)w * 2 + wave) * 64 + lane) * 4);
                st_sc1_f4(px_rs, (unsigned)(((((size_t)par * G + j) * 2 + wave) * 256 + bl * 16 + q * 4) * 4), v);
            }
            group_publish(gs);
        }
        if (own && bvalid && q_da) *(f32x4*)q_da = da4;
        c0v = c1v; c1v = n_c2;
        q_dg += z_dg; q_da += z_da; q_do += z_do;
    }
}

// ------------------------------------------------------------------------------------------------
// host entry points
// ------------------------------------------------------------------------------------------------
struct asr_lstm_scan_desc {      // mirrored in include/asr_amd.h
    int T, B, N, ndir, ln_kind, cell_ln;
    float grad_clip;
    const float* mask; int mask_stride_b, mask_stride_t;
    LstmDir dir[2];
};

static int pick_ut(int N, int ngroups_per_step) {
    // smallest UT in {1,2,4} whose grid fits 256 co-resident workgroups
    for (int ut = 1; ut <= 4; ut *= 2) {
        if (N % (4 * ut)) continue;
        if ((N / (4 * ut)) * ngroups_per_step <= 256) return ut;
    }
    return 0;
}

extern "C" size_t asr_lstm_scan_ws_bytes(int T, int B, int N, int ndir, int backward) {
    (void)T;
    int KS, KSP, KT;
    lstm_dims(N, &KS, &KSP, &KT);
    const int nchunk = (B + 15) / 16;
    const size_t ng = (size_t)ndir * nchunk;
    size_t bytes = ng * SYNC_STRIDE * 4;
    bytes += ng * 4 * KSP * 64 * 4;           // rex (r, or (c, o) for lstm_scan_fwd_cln_kernel)
    bytes += 5 * ng * 2 * KS * 32 * 4;        // sta (x4 for per-gate LN), stc
    if (backward) bytes += ng * 2 * (size_t)KS * KS * 64 * 4;   // dex (G <= KS)
    return bytes + 1024 + STAMP_BYTES;
}

static int lstm_scan_launch(const asr_lstm_scan_desc* d, void* ws, size_t ws_bytes, int backward,
                            hipStream_t stream) {
    if (!d || !ws) return ASR_ERR_ARG;
    if (d->T < 0 || d->B <= 0 || d->N <= 0 || (d->N & 3) || d->ndir < 1 || d->ndir > 2 ||
        d->ln_kind < 0 || d->ln_kind > 4) return ASR_ERR_ARG;
    if (!d->mask) return ASR_ERR_ARG;
    for (int k = 0; k < d->ndir; ++k) {
        const LstmDir& D = d->dir[k];
        if (!D.wpack || !D.cst) return ASR_ERR_ARG;
        if (!backward && (!D.xin || !D.r_out)) return ASR_ERR_ARG;
        if (backward && (!D.gates || !D.dr_ext || !D.dG)) return ASR_ERR_ARG;
        if (d->ln_kind && (!D.alpha_hid || !D.ahat || !D.rs_a)) return ASR_ERR_ARG;
        if (d->ln_kind && backward && !D.dA) return ASR_ERR_ARG;
        if (d->cell_ln && (!D.alpha_c || !D.beta_c || !D.chat || !D.rs_c)) return ASR_ERR_ARG;
    }
    if (ws_bytes < asr_lstm_scan_ws_bytes(d->T, d->B, d->N, d->ndir, backward)) return ASR_ERR_WS;
    if (d->T == 0) return ASR_OK;

    LstmScanArgs A;
    A.T = d->T; A.B = d->B; A.N = d->N;
    lstm_dims(d->N, &A.KS, &A.KSP, &A.KT);
    A.ndir = d->ndir;
    const int nchunk_total = (d->B + 15) / 16;
    A.per_gate = d->ln_kind >= 3 ? 1 : 0;
    A.ln_kind = d->ln_kind >= 3 ? d->ln_kind - 2 : d->ln_kind;
    A.cell_ln = d->cell_ln; A.clip = d->grad_clip;
    A.mask = d->mask; A.msb = d->mask_stride_b; A.mst = d->mask_stride_t;
    A.d[0] = d->dir[0];
    A.d[1] = d->dir[d->ndir > 1 ? 1 : 0];
    // chunks per launch limited by co-residency (<= 256 workgroups in flight)
    int cpl = 256 / (d->ndir * (d->N / 4));
    int UT = 1;
    if (cpl >= 1) {
        if (cpl > nchunk_total) cpl = nchunk_total;
    } else {
        cpl = 1;
        UT = pick_ut(d->N, d->ndir);
        if (!UT) return ASR_ERR_LIMIT;   // N too large for a co-resident grid
    }
    A.UT = UT; A.G = d->N / (4 * UT);

    char* p = (char*)ws;
    const size_t ng = (size_t)d->ndir * nchunk_total;
    A.sync = (unsigned*)p;                         p += ng * SYNC_STRIDE * 4;
    A.rex = (float*)p;                             p += ng * 4 * A.KSP * 64 * 4;
    A.sta = (float*)p;                             p += ng * 2 * A.KS * 128 * 4;
    A.stc = (float*)p;                             p += ng * 2 * A.KS * 32 * 4;
    A.dex = backward ? (float*)p : nullptr;
    A.dbg = (unsigned long long*)((char*)ws + asr_lstm_scan_ws_bytes(d->T, d->B, d->N, d->ndir, backward) - STAMP_BYTES);
    if (hipMemsetAsync(A.sync, 0, ng * SYNC_STRIDE * 4, stream) != hipSuccess) return ASR_ERR_LAUNCH;

    // cell-LN nets at UT = 1, N <= 512: the forward kernel with one exchange fewer per step
    // ... for batches whose chunks all fit one launch as workgroups of their own.  With two teams that kernel needs
    // ~330 registers per lane against 256 and spills (cfg2 at 32 rows: 114.6 k frames/s, against 146.2 k with the
    // generic kernel's teams and 96-114 k with one launch per chunk), so larger batches take the generic kernel for
    // EVERY chunk: the two kernels sum the cell-LN statistics in different orders, and a batch's rows must not
    // depend on which launch carried them.
    // options of the recurrent_layers.py classes: the generic kernels' OPT instantiation, one chunk per workgroup; the
    // backward weights must then be packed with mode 1 whatever asr_lstm_bwd_pack_mode says
    const bool opt = lstm_dir_has_options(d->dir[0]) || (d->ndir > 1 && lstm_dir_has_options(d->dir[1]));
    for (int k = 0; k < d->ndir; ++k) {
        const LstmDir& D = d->dir[k];
        if (D.a0_until_valid && !D.a0) return ASR_ERR_ARG;
        if (D.peep && (d->ln_kind || d->cell_ln)) return ASR_ERR_ARG;      // no class has both (:53 vs :555)
        if (!backward && (D.dc_ext || D.dc0 || D.dr0 || D.da0 || D.dpeep)) return ASR_ERR_ARG;
        if (D.dpeep && !D.peep) return ASR_ERR_ARG;
    }
    const bool fused_cln = !backward && A.cell_ln && UT == 1 && A.KSP / 16 <= 8 && nchunk_total <= cpl &&
                           !d->dir[0].a0 && !(d->ndir > 1 && d->dir[1].a0) && !opt;
    const bool ln2d = backward && UT == 1 && !opt && asr_lstm_bwd_pack_mode(d->N, d->ndir, d->ln_kind) == 2;
    // LDS of a workgroup whose nc teams carry nc chunks: the resident weights once, the staging buffers per team
    bool wstream = false;
    auto lds_of = [&](int nc) -> size_t {
        if (ln2d) return ((size_t)(A.KS / 32) * 2048 + (size_t)nc * 2304) * 4 + 64;
        if (backward) return ((wstream ? 0 : (size_t)UT * A.KT * 256) + (size_t)nc * UT * 1280) * 4 + 64;
        return ((wstream ? 0 : (size_t)UT * A.KSP * 64) + (size_t)nc * UT * 1024) * 4 + 64 + (fused_cln ? 1024 * nc : 0);
    };
    if (lds_of(1) > 160 * 1024) {
        // The resident weights do not fit a CU's LDS.  A bidirectional launch is refused (the caller runs the directions
        // one by one: half the workgroups, possibly fewer tiles per workgroup, possibly resident again); a single
        // direction streams its weights from L2 / Infinity Cache every step (UT >= 2 here; options: not instantiated)
        if (d->ndir > 1 || opt || UT < 2) return ASR_ERR_LIMIT;
        wstream = true;
    }
    const int nc_cap = scan_nc_cap() >= 2 ? 2 : 1;

    for (int c0 = 0; c0 < nchunk_total;) {
        const int rem = nchunk_total - c0;
        // chunks that do not fit one launch as workgroups of their own (cpl z slices) ride as teams of two inside a
        // workgroup instead of waiting for a launch of their own
        int nc = (rem > cpl && nc_cap >= 2 && !opt && !wstream && lds_of(2) <= 160 * 1024) ? 2 : 1;
        int nz = nc == 2 ? rem / 2 : rem;
        if (nz > cpl) nz = cpl;
        size_t lds = lds_of(nc);
        // one workgroup per CU (what the sc1 hand-off is measured for, MI355X_MICROARCH.md 'Valid forms'): a launch
        // never exceeds 256 workgroups, so asking for more than half of a CU's 160 KB spreads them one per CU
        if (lds < 82 * 1024) lds = 82 * 1024;
        LstmScanArgs Al = A;
        Al.nchunk = nz * nc;
        Al.chunk0 = c0;
        dim3 grid(A.G, d->ndir, nz), block(256 * nc);
        if (!scan_grid_fits((size_t)A.G * d->ndir * nz, lds)) return ASR_ERR_LIMIT;
        const void* fn = nullptr;
#define PICK_NC_(...) (nc == 2 ? (const void*)__VA_ARGS__, 2> : (const void*)__VA_ARGS__, 1>)
        if (wstream) {
            if (!backward) fn = UT == 2 ? (const void*)lstm_scan_fwd_kernel<2, 1, false, true> : (const void*)lstm_scan_fwd_kernel<4, 1, false, true>;
            else fn = UT == 2 ? (const void*)lstm_scan_bwd_kernel<2, 1, false, true> : (const void*)lstm_scan_bwd_kernel<4, 1, false, true>;
        } else if (opt) {
            if (!backward) fn = UT == 1 ? (const void*)lstm_scan_fwd_kernel<1, 1, true> : UT == 2 ? (const void*)lstm_scan_fwd_kernel<2, 1, true> : (const void*)lstm_scan_fwd_kernel<4, 1, true>;
            else fn = UT == 1 ? (const void*)lstm_scan_bwd_kernel<1, 1, true> : UT == 2 ? (const void*)lstm_scan_bwd_kernel<2, 1, true> : (const void*)lstm_scan_bwd_kernel<4, 1, true>;
        } else if (!backward) {
            if (fused_cln) fn = (const void*)lstm_scan_fwd_cln_kernel<1>;
            else if (UT == 1) fn = PICK_NC_(lstm_scan_fwd_kernel<1);
            else if (UT == 2) fn = PICK_NC_(lstm_scan_fwd_kernel<2);
            else fn = PICK_NC_(lstm_scan_fwd_kernel<4);
        } else {
            if (ln2d) fn = A.per_gate ? PICK_NC_(lstm_scan_bwd_ln2d_kernel<1) : PICK_NC_(lstm_scan_bwd_ln2d_kernel<0);
            else if (UT == 1) fn = PICK_NC_(lstm_scan_bwd_kernel<1);
            else if (UT == 2) fn = PICK_NC_(lstm_scan_bwd_kernel<2);
            else fn = PICK_NC_(lstm_scan_bwd_kernel<4);
        }
#undef PICK_NC_
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return ASR_ERR_LAUNCH;
        void* kargs[] = {(void*)&Al};
        if (hipLaunchKernel(fn, grid, block, kargs, lds, stream) != hipSuccess) return ASR_ERR_LAUNCH;
        ASR_CHECK_LAUNCH();
        c0 += nz * nc;
    }
    return ASR_OK;
}

extern "C" int asr_lstm_scan_fwd(const asr_lstm_scan_desc* d, void* ws, size_t ws_bytes, hipStream_t stream) {
    return lstm_scan_launch(d, ws, ws_bytes, 0, stream);
}
extern "C" int asr_lstm_scan_bwd(const asr_lstm_scan_desc* d, void* ws, size_t ws_bytes, hipStream_t stream) {
    return lstm_scan_launch(d, ws, ws_bytes, 1, stream);
}

// Host-visible check of the group abort words after the stream has been synchronised by the caller:
// returns ASR_ERR_TIMEOUT when any group gave up.  `ngroups` = ndir * ceil(B/16).
extern "C" int asr_scan_status(const void* ws, int ngroups) {
    if (!ws || ngroups <= 0 || ngroups > 4096) return ASR_ERR_ARG;
    unsigned* host = (unsigned*)malloc((size_t)ngroups * SYNC_STRIDE * 4);
    if (!host) return ASR_ERR_ARG;
    int rc = ASR_OK;
    if (hipMemcpy(host, ws, (size_t)ngroups * SYNC_STRIDE * 4, hipMemcpyDeviceToHost) != hipSuccess)
        rc = ASR_ERR_LAUNCH;
    else
        for (int g = 0; g < ngroups; ++g)
            if (host[(size_t)g * SYNC_STRIDE + SYNC_ABORT]) rc = ASR_ERR_TIMEOUT;
    free(host);
    return rc;
}
