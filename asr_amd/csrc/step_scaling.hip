// ScalingHyperLSTMLayer with its options (libs/lasagne_libs/hyper_layers.py:525-1205: use_peepholes :977-979,
// :1006,:1042-1044,:1072; use_layer_norm :981-990,:1009-1015,:1046-1055,:1075-1081; learn_init :1127-1131).
//
// Without options the layer runs on the persistent HyperLSTM scan (engine_scaling.ScalingPair).  Per-gate layer
// norms put five all-unit reductions into each of the two cells of a step, i.e. four more chip-wide exchanges per
// step than that kernel has, so the option path is a step loop inside the library like ProjectionHyperLSTMLayer's
// (step_prjhyper.hip): per time step two skinny products on the previous hidden state, the inner cell, the scale
// maps, the outer cell; one workgroup per (batch row, direction) runs a cell, so every layer norm is a
// workgroup-wide reduction.  The reference uses this layer at TIMIT scale (64 inner / 128 outer units,
// scripts/timit/exp_scalehyper.py:331-333); the path is launch-bound and sized for that.
//
// Layouts are the reference's: gate vectors gate-major [i | f | c | o], sequences time-major (T,B,.).  The hidden
// state is carried unprojected, hid = [inner_hid | outer_hid] (H + N wide).
#include "step_skinny.h"

struct ScalDir {                // mirrored in include/asr_amd.h (asr_scaling_dir)
    int reverse;
    const float* inner_in;      // (T,B,4H) x.W_inner_in + b_inner                          (:951)
    const float* outer_in;      // (T,B,4N) x.W_outer_in                                    (:952)
    const float* Wio;           // (H+N,4H) [W_inner_hid_to_inner ; W_outer_hid_to_inner]   (:974-975)
    const float* S;             // (H,NS*4N) [to_in | to_hid | to_bias | to_cell]           (:1030-1033)
    const float* s;             // (NS*4N)
    const float* Woo;           // (N,4N)                                                   (:1039)
    const float* wpeep;         // (4H) inner peephole vectors, cell block zero; null without peepholes
    const float *lna_i, *lnb_i; // (5,H) alpha / beta of the layer norms of ingate, forgetgate, cell, outgate, outcell; null without LN
    const float *lna_o, *lnb_o; // (5,N)
    const float* init;          // (2H+2N) inner cell, inner hid, outer cell, outer hid initial rows; null = zeros
    float* hid;                 // (T,B,H+N) [inner_hid | outer_hid]
    // saved for BPTT
    float *gi, *go;             // (T,B,4H) / (T,B,4N) post-activation gates
    float *ci, *co;             // (T,B,H) / (T,B,N) cell state (masked)
    float *xh_i, *xh_o;         // (T,B,5H) / (T,B,5N) normalised values of the five layer norms (LN only)
    float *rs_i, *rs_o;         // (T,B,5) rstd (LN only)
    float* sc;                  // (T,B,NS*4N) scale maps
    float* a;                   // (T,B,4N) outer_hid_prev . W_outer_hid_to_outer
    // backward
    const float* dhid;          // (T,B,H+N) dLoss / d hid
    float *dgi, *dgo;           // (T,B,4H) d inner gate pre-activations ; (T,B,4N) d outer_input
    float* dsc;                 // (T,B,NS*4N)
    float* da;                  // (T,B,4N)
    float* dh;                  // (T,B,H+N) total d hid_t (work)
    float *dy_i, *dy_o;         // (T,B,5H) / (T,B,5N) clipped gradients at the layer-norm outputs (LN only)
    float* dpeep;               // (T,B,3H) d gate * cell for the inner peepholes (i, f, o); null without peepholes
    float* dinit;               // (B,2H+2N) gradient wrt the initial state rows, per batch row; may be null
    // EqualScaleHyperLSTMLayer with learn_init (hyper_layers.py:1782-1788): its learned inner_hid_init lives in the
    // PROJECTED space, while the inner state carried here is the unprojected one (the projection is folded into Wio / S)
    const float* tg0;           // (4H) inner_hid_init . W_inner_hid_to_inner: added to the inner gates of a row as long as
                                // its previous state is still the initial one (first step, or masked so far); null = none
    float* dgi0;                // (Btot,4H) backward out: d inner gate pre-activations of each row's first unmasked step
                                // (rows without one are left untouched: clear beforehand); null = not wanted
};

struct ScalArgs {
    int T, B, N, H, ndir, NS, use_ln;
    int b0, Btot;
    float clip;
    const float* mask; int msb, mst;
    ScalDir d[2];
    int t[2], tp[2];
    float* tg;                  // (ndir,B,4H) recurrent part of the inner gates
    float* g_c[2];              // (ndir,B,H) / (ndir,B,N) carried d cell
    float* h0;                  // (ndir,B,H+N) initial hid rows
    float* c0[2];               // (ndir,B,H) / (ndir,B,N) initial cells
    float* dh0;                 // (ndir,B,H+N) d hid of the state before the first step
};

template <int K>
__device__ __forceinline__ void block_sumK(float (&v)[K], float* sm) {
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) sm[(threadIdx.x >> 6) * K + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = sm[k] + sm[K + k] + sm[2 * K + k] + sm[3 * K + k];
}

// state rows before the first step: the learned initial rows broadcast over the batch (:1127-1131), or zeros
__global__ void scal_init_kernel(ScalArgs A, int backward) {
    const int dir = blockIdx.y, b = blockIdx.x;
    const ScalDir& D = A.d[dir];
    const int H = A.H, N = A.N;
    const size_t r = (size_t)dir * A.B + b;
    for (int i = threadIdx.x; i < H + N; i += blockDim.x) {
        const bool inner = i < H;
        const int u = inner ? i : i - H;
        const float cv = D.init ? D.init[inner ? u : 2 * H + u] : 0.f;
        const float hv = D.init ? D.init[inner ? H + u : 2 * H + N + u] : 0.f;
        (inner ? A.c0[0] + r * H : A.c0[1] + r * N)[u] = cv;
        A.h0[r * (H + N) + i] = hv;
        if (backward) {
            A.dh0[r * (H + N) + i] = 0.f;
            (inner ? A.g_c[0] + r * H : A.g_c[1] + r * N)[u] = 0.f;
        }
    }
}

// one workgroup per (batch row, direction); NU = units per thread (U <= 256 NU)
template <int OUTER, int NU>
__global__ __launch_bounds__(256) void scal_cell_fwd_kernel(ScalArgs A) {
    __shared__ float sm[12];
    const int b = blockIdx.x, dir = blockIdx.y;
    const ScalDir& D = A.d[dir];
    const int H = A.H, N = A.N, U = OUTER ? N : H, W = H + N, off = OUTER ? H : 0;
    const int t = A.t[dir], tp = A.tp[dir];
    const size_t row = (size_t)t * A.Btot + A.b0 + b, rowp = (size_t)(tp < 0 ? 0 : tp) * A.Btot + A.b0 + b;
    const size_t wr = (size_t)dir * A.B + b;
    const bool on = A.mask[(size_t)(A.b0 + b) * A.msb + (size_t)t * A.mst] != 0.f;
    const float* cprev_p = tp < 0 ? A.c0[OUTER] + wr * U : (OUTER ? D.co : D.ci) + rowp * U;
    const float* hprev_p = tp < 0 ? A.h0 + wr * W + off : D.hid + rowp * W + off;
    // left-aligned masks: the previous scan step masked (or none) means no step of this row has run yet
    const bool virgin = !OUTER && D.tg0 && (tp < 0 || A.mask[(size_t)(A.b0 + b) * A.msb + (size_t)tp * A.mst] == 0.f);
    const float* lna = OUTER ? D.lna_o : D.lna_i;
    const float* lnb = OUTER ? D.lnb_o : D.lnb_i;
    float* gsv = (OUTER ? D.go : D.gi) + row * 4 * U;
    float* xh_sv = (OUTER ? D.xh_o : D.xh_i);
    float pre[NU][4], cp[NU], pk[NU][4];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int n = min((int)threadIdx.x + u * 256, U - 1);
        cp[u] = cprev_p[n];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (OUTER) {
                const float* sc = D.sc + row * A.NS * 4 * N;
                pre[u][k] = D.outer_in[row * 4 * N + k * N + n] * sc[k * N + n] + D.a[row * 4 * N + k * N + n] * sc[4 * N + k * N + n] +
                            sc[8 * N + k * N + n];                                                  // :1038-1040
                pk[u][k] = A.NS == 4 ? sc[12 * N + k * N + n] : 0.f;
            } else {
                pre[u][k] = D.inner_in[row * 4 * H + k * H + n] + A.tg[wr * 4 * H + k * H + n];        // :973-975
                if (virgin) pre[u][k] += D.tg0[k * H + n];
                pk[u][k] = D.wpeep ? D.wpeep[k * H + n] : 0.f;
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) pre[u][k] += cp[u] * pk[u][k];                                  // :977-979 | :1042-1044
    }
    float rstd[5] = {1.f, 1.f, 1.f, 1.f, 1.f};
    float xh[NU][5];
    if (A.use_ln) {                                                                                 // :981-990 | :1046-1055
        float s1[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < NU; ++u)
            if ((int)threadIdx.x + u * 256 < U) { s1[0] += pre[u][0]; s1[1] += pre[u][1]; s1[2] += pre[u][2]; }
        block_sumK<3>(s1, sm);
        float s2[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < NU; ++u)
            if ((int)threadIdx.x + u * 256 < U) {
#pragma unroll
                for (int k = 0; k < 3; ++k) { const float dl = pre[u][k] - s1[k] / U; s2[k] += dl * dl; }
            }
        block_sumK<3>(s2, sm);
#pragma unroll
        for (int k = 0; k < 3; ++k) rstd[k] = 1.0f / sqrtf(s2[k] / U + ASR_EPS);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int n = min((int)threadIdx.x + u * 256, U - 1);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                xh[u][k] = (pre[u][k] - s1[k] / U) * rstd[k];
                pre[u][k] = lna[k * U + n] * xh[u][k] + lnb[k * U + n];
            }
        }
    }
    float ig[NU], fg[NU], gc[NU], cn[NU], oc[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        ig[u] = sigmoidf_(pre[u][0]); fg[u] = sigmoidf_(pre[u][1]); gc[u] = tanhf_(pre[u][2]);
        cn[u] = fg[u] * cp[u] + ig[u] * gc[u];                                                      // :1005 | :1071
        pre[u][3] += cn[u] * pk[u][3];                                                              // :1006 | :1072
        oc[u] = cn[u];
    }
    if (A.use_ln) {                                                                                 // :1009-1015 | :1075-1081
        float s1[2] = {0.f, 0.f};
#pragma unroll
        for (int u = 0; u < NU; ++u)
            if ((int)threadIdx.x + u * 256 < U) { s1[0] += pre[u][3]; s1[1] += cn[u]; }
        block_sumK<2>(s1, sm);
        float s2[2] = {0.f, 0.f};
#pragma unroll
        for (int u = 0; u < NU; ++u)
            if ((int)threadIdx.x + u * 256 < U) {
                const float d0 = pre[u][3] - s1[0] / U, d1 = cn[u] - s1[1] / U;
                s2[0] += d0 * d0; s2[1] += d1 * d1;
            }
        block_sumK<2>(s2, sm);
        rstd[3] = 1.0f / sqrtf(s2[0] / U + ASR_EPS); rstd[4] = 1.0f / sqrtf(s2[1] / U + ASR_EPS);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int n = min((int)threadIdx.x + u * 256, U - 1);
            xh[u][3] = (pre[u][3] - s1[0] / U) * rstd[3];
            xh[u][4] = (cn[u] - s1[1] / U) * rstd[4];
            pre[u][3] = lna[3 * U + n] * xh[u][3] + lnb[3 * U + n];
            oc[u] = lna[4 * U + n] * xh[u][4] + lnb[4 * U + n];
        }
        if (threadIdx.x < 5) (OUTER ? D.rs_o : D.rs_i)[row * 5 + threadIdx.x] = rstd[threadIdx.x];
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int n = threadIdx.x + u * 256;
        if (n >= U) continue;
        const float o = sigmoidf_(pre[u][3]);
        const float h = o * tanhf_(oc[u]);                                                          // :1025 | :1091
        gsv[n] = ig[u]; gsv[U + n] = fg[u]; gsv[2 * U + n] = gc[u]; gsv[3 * U + n] = o;
        if (A.use_ln) {
#pragma unroll
            for (int k = 0; k < 5; ++k) xh_sv[row * 5 * U + k * U + n] = xh[u][k];
        }
        (OUTER ? D.co : D.ci)[row * U + n] = on ? cn[u] : cp[u];                                    // :1113-1116
        D.hid[row * W + off + n] = on ? h : hprev_p[n];
    }
}

template <int OUTER, int NU>
__global__ __launch_bounds__(256) void scal_cell_bwd_kernel(ScalArgs A) {
    __shared__ float sm[24];
    const int b = blockIdx.x, dir = blockIdx.y;
    const ScalDir& D = A.d[dir];
    const int H = A.H, N = A.N, U = OUTER ? N : H, W = H + N, off = OUTER ? H : 0;
    const int t = A.t[dir], tp = A.tp[dir];
    const size_t row = (size_t)t * A.Btot + A.b0 + b, rowp = (size_t)(tp < 0 ? 0 : tp) * A.Btot + A.b0 + b;
    const size_t wr = (size_t)dir * A.B + b;
    const bool on = A.mask[(size_t)(A.b0 + b) * A.msb + (size_t)t * A.mst] != 0.f;
    const float* cprev_p = tp < 0 ? A.c0[OUTER] + wr * U : (OUTER ? D.co : D.ci) + rowp * U;
    const float* cnew_p = (OUTER ? D.co : D.ci) + row * U;
    float* dhprev_p = tp < 0 ? A.dh0 + wr * W + off : D.dh + rowp * W + off;
    const bool first_on = !OUTER && D.dgi0 && on && (tp < 0 || A.mask[(size_t)(A.b0 + b) * A.msb + (size_t)tp * A.mst] == 0.f);
    const float* lna = OUTER ? D.lna_o : D.lna_i;
    const float* lnb = OUTER ? D.lnb_o : D.lnb_i;
    const float* gsv = (OUTER ? D.go : D.gi) + row * 4 * U;
    const float* xh_sv = (OUTER ? D.xh_o : D.xh_i) + (A.use_ln ? row * 5 * U : 0);
    float* g_c = A.g_c[OUTER] + wr * U;
    float* dy_sv = OUTER ? D.dy_o : D.dy_i;
    float rstd[5] = {1.f, 1.f, 1.f, 1.f, 1.f};
    if (A.use_ln) {
#pragma unroll
        for (int k = 0; k < 5; ++k) rstd[k] = (OUTER ? D.rs_o : D.rs_i)[row * 5 + k];
    }
    float dh[NU], gcar[NU], g4[NU][4], cp[NU], cn[NU], pk[NU][4], xh[NU][5], al[NU][5];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int n = min((int)threadIdx.x + u * 256, U - 1);
        const bool act = on && (int)threadIdx.x + u * 256 < U;
        dh[u] = act ? D.dh[row * W + off + n] : 0.f;
        gcar[u] = act ? g_c[n] : 0.f;
        cp[u] = cprev_p[n]; cn[u] = cnew_p[n];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            g4[u][k] = gsv[k * U + n];
            if (OUTER) pk[u][k] = A.NS == 4 ? D.sc[row * A.NS * 4 * N + 12 * N + k * N + n] : 0.f;
            else pk[u][k] = D.wpeep ? D.wpeep[k * H + n] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            xh[u][k] = A.use_ln ? xh_sv[k * U + n] : 0.f;
            al[u][k] = A.use_ln ? lna[k * U + n] : 1.f;
        }
    }
    // gradient at the (clipped) outputs of the out-gate and out-cell layer norms, then through them
    float d[NU][5];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int n = min((int)threadIdx.x + u * 256, U - 1);
        const float oc = A.use_ln ? al[u][4] * xh[u][4] + lnb[4 * U + n] : cn[u];
        const float tc = tanhf_(oc), o = g4[u][3];
        d[u][3] = clipf_(dh[u] * tc * o * (1.f - o), A.clip);                                       // :1016-1019 | :1083-1086
        d[u][4] = clipf_(dh[u] * o * (1.f - tc * tc), A.clip);                                      // :1020-1022 | :1087-1089
    }
    float dpre[NU][5];
    if (A.use_ln) {
        float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            s[0] += d[u][3] * al[u][3]; s[1] += d[u][3] * al[u][3] * xh[u][3];
            s[2] += d[u][4] * al[u][4]; s[3] += d[u][4] * al[u][4] * xh[u][4];
        }
        block_sumK<4>(s, sm);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            dpre[u][3] = rstd[3] * (d[u][3] * al[u][3] - s[0] / U - xh[u][3] * s[1] / U);
            dpre[u][4] = rstd[4] * (d[u][4] * al[u][4] - s[2] / U - xh[u][4] * s[3] / U);
        }
    } else {
#pragma unroll
        for (int u = 0; u < NU; ++u) { dpre[u][3] = d[u][3]; dpre[u][4] = d[u][4]; }
    }
    float dc[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const bool act = on && (int)threadIdx.x + u * 256 < U;
        dc[u] = act ? gcar[u] + dpre[u][4] + dpre[u][3] * pk[u][3] : 0.f;
        const float ig = g4[u][0], fg = g4[u][1], gc = g4[u][2];
        d[u][0] = clipf_(dc[u] * gc * ig * (1.f - ig), A.clip);                                     // :992-1001 | :1057-1066
        d[u][1] = clipf_(dc[u] * cp[u] * fg * (1.f - fg), A.clip);
        d[u][2] = clipf_(dc[u] * ig * (1.f - gc * gc), A.clip);
    }
    if (A.use_ln) {
        float s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int k = 0; k < 3; ++k) { s[2 * k] += d[u][k] * al[u][k]; s[2 * k + 1] += d[u][k] * al[u][k] * xh[u][k]; }
        block_sumK<6>(s, sm);
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int k = 0; k < 3; ++k)
                dpre[u][k] = rstd[k] * (d[u][k] * al[u][k] - s[2 * k] / U - xh[u][k] * s[2 * k + 1] / U);
    } else {
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int k = 0; k < 3; ++k) dpre[u][k] = d[u][k];
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int n = threadIdx.x + u * 256;
        if (n >= U) continue;
        if (on) g_c[n] = dc[u] * g4[u][1] + dpre[u][0] * pk[u][0] + dpre[u][1] * pk[u][1] + dpre[u][2] * pk[u][2];
        else dhprev_p[n] += D.dh[row * W + off + n];     // switched off: the state was copied, its gradient passes on
        if (A.use_ln) {
#pragma unroll
            for (int k = 0; k < 5; ++k) dy_sv[row * 5 * U + k * U + n] = d[u][k];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float dg = dpre[u][k];
            const size_t i4 = row * 4 * U + k * U + n;
            if (OUTER) {
                const float* sc = D.sc + row * A.NS * 4 * N;
                D.dgo[i4] = dg * sc[k * N + n];                                                    // d outer_input
                D.da[i4] = dg * sc[4 * N + k * N + n];                                             // d (outer_hid_prev.W)
                float* ds = D.dsc + row * A.NS * 4 * N;
                ds[k * N + n] = dg * D.outer_in[i4];
                ds[4 * N + k * N + n] = dg * D.a[i4];
                ds[8 * N + k * N + n] = dg;
                if (A.NS == 4) ds[12 * N + k * N + n] = k == 2 ? 0.f : dg * (k == 3 ? cn[u] : cp[u]);   // no map for the cell gate
            } else {
                D.dgi[i4] = dg;
                if (first_on) D.dgi0[(size_t)(A.b0 + b) * 4 * H + k * H + n] = dg;
            }
        }
        if (!OUTER && D.dpeep) {
            float* dp = D.dpeep + row * 3 * H;
            dp[n] = dpre[u][0] * cp[u]; dp[H + n] = dpre[u][1] * cp[u]; dp[2 * H + n] = dpre[u][3] * cn[u];
        }
    }
}

// gradient wrt the state rows before the first step, per batch row: [d inner cell | d inner hid | d outer cell | d outer hid]
__global__ void scal_dinit_kernel(ScalArgs A) {
    const int dir = blockIdx.y, b = blockIdx.x;
    const ScalDir& D = A.d[dir];
    if (!D.dinit) return;
    const int H = A.H, N = A.N;
    const size_t r = (size_t)dir * A.B + b;
    float* o = D.dinit + (size_t)(A.b0 + b) * (2 * H + 2 * N);
    for (int i = threadIdx.x; i < H; i += blockDim.x) { o[i] = A.g_c[0][r * H + i]; o[H + i] = A.dh0[r * (H + N) + i]; }
    for (int i = threadIdx.x; i < N; i += blockDim.x) { o[2 * H + i] = A.g_c[1][r * N + i]; o[2 * H + N + i] = A.dh0[r * (H + N) + H + i]; }
}

#define SC_MAXU 4
template <int OUTER>
static void scal_cell_launch(const ScalArgs& A, bool backward, hipStream_t st) {
    const dim3 grid(A.B, A.ndir), block(256);
    const int U = OUTER ? A.N : A.H;
#define SC_GO(NU_)                                                                                       \
    do {                                                                                                 \
        if (backward) hipLaunchKernelGGL((scal_cell_bwd_kernel<OUTER, NU_>), grid, block, 0, st, A);     \
        else hipLaunchKernelGGL((scal_cell_fwd_kernel<OUTER, NU_>), grid, block, 0, st, A);              \
    } while (0)
    if (U <= 256) SC_GO(1);
    else if (U <= 512) SC_GO(2);
    else SC_GO(SC_MAXU);
#undef SC_GO
}

struct asr_scaling_desc {       // mirrored in include/asr_amd.h
    int T, B, N, H, ndir;
    int b0, Btot;               // rows [b0, b0+B) of Btot; B <= 64 per call
    int use_peepholes, use_layer_norm;
    float grad_clip;
    const float* mask; int mask_stride_b, mask_stride_t;
    ScalDir dir[2];
};

extern "C" size_t asr_scaling_ws_bytes(int B, int N, int H, int ndir) {
    return ((size_t)ndir * B * (4 * H + 2 * (H + N) + 2 * (H + N)) + SK_PART_FLOATS) * sizeof(float) +
           SK_TICKETS * sizeof(unsigned) + 256;
}

struct ScalHost {
    ScalArgs A;
    float* part; unsigned* tickets;
    int nd;
    hipStream_t st;
};

static int scal_setup(const asr_scaling_desc* d, void* ws, size_t ws_bytes, hipStream_t st, ScalHost& Hh) {
    if (!d || !ws || d->T < 0 || d->B <= 0 || d->B > 64 || d->N <= 0 || d->H <= 0 || (d->N & 3) || (d->H & 3) ||
        d->N > 256 * SC_MAXU || d->H > 256 * SC_MAXU || d->ndir < 1 || d->ndir > 2 || !d->mask || d->b0 < 0 ||
        d->b0 + d->B > d->Btot || ((uintptr_t)ws & 15))
        return ASR_ERR_ARG;
    if (ws_bytes < asr_scaling_ws_bytes(d->B, d->N, d->H, d->ndir)) return ASR_ERR_WS;
    ScalArgs& A = Hh.A;
    A.T = d->T; A.B = d->B; A.N = d->N; A.H = d->H; A.ndir = d->ndir; A.clip = d->grad_clip;
    A.NS = d->use_peepholes ? 4 : 3; A.use_ln = d->use_layer_norm ? 1 : 0;
    A.b0 = d->b0; A.Btot = d->Btot;
    A.mask = d->mask; A.msb = d->mask_stride_b; A.mst = d->mask_stride_t;
    A.d[0] = d->dir[0]; A.d[1] = d->dir[d->ndir > 1 ? 1 : 0];
    float* p = (float*)ws;
    const size_t nb = (size_t)d->ndir * d->B;
    A.tg = p; p += nb * 4 * d->H;
    A.g_c[0] = p; p += nb * d->H; A.g_c[1] = p; p += nb * d->N;
    A.c0[0] = p; p += nb * d->H; A.c0[1] = p; p += nb * d->N;
    A.h0 = p; p += nb * (d->H + d->N);
    A.dh0 = p; p += nb * (d->H + d->N);
    Hh.part = p; p += SK_PART_FLOATS;
    Hh.tickets = (unsigned*)p;
    Hh.nd = d->ndir; Hh.st = st;
    if (hipMemsetAsync(Hh.tickets, 0, SK_TICKETS * sizeof(unsigned), st) != hipSuccess) return ASR_ERR_LAUNCH;
    return ASR_OK;
}

static int scal_mm(ScalHost& Hh, bool nt, const float* const A[2], int lda, const float* const W[2], int ldw,
                   const float* const bias[2], float* const Cm[2], int ldc, int K, int Nc, int beta) {
    SkinnyArgs S;
    for (int k = 0; k < SK_MAXP; ++k) {
        const int kk = k < Hh.nd ? k : 0;
        S.p[k].A = A[kk]; S.p[k].W = W[kk]; S.p[k].bias = bias ? bias[kk] : nullptr; S.p[k].C = Cm[kk];
    }
    S.lda = lda; S.ldw = ldw; S.ldc = ldc; S.Bm = Hh.A.B; S.K = K; S.Nc = Nc; S.beta = beta; S.nsplit = 0;
    S.raw = 0; skinny_no_zero_block(S, nt);
    S.part = Hh.part; S.tickets = Hh.tickets;
    return skinny(nt, S, Hh.nd, Hh.st);
}

static void scal_times(const asr_scaling_desc* d, ScalArgs& A, int step) {
    const int T = d->T;
    for (int k = 0; k < d->ndir; ++k) {
        A.t[k] = A.d[k].reverse ? T - 1 - step : step;
        A.tp[k] = step == 0 ? -1 : (A.d[k].reverse ? A.t[k] + 1 : A.t[k] - 1);
    }
    if (d->ndir == 1) { A.t[1] = A.t[0]; A.tp[1] = A.tp[0]; }
}

// Batches of more than 64 rows are run by the caller in chunks (b0 advanced); a launch carries every 16-row chunk.
extern "C" int asr_scaling_fwd(const asr_scaling_desc* d, void* ws, size_t ws_bytes, hipStream_t st) {
    ScalHost Hh;
    int rc = scal_setup(d, ws, ws_bytes, st, Hh);
    if (rc) return rc;
    for (int k = 0; k < d->ndir; ++k) {
        const ScalDir& D = d->dir[k];
        if (!D.inner_in || !D.outer_in || !D.Wio || !D.S || !D.s || !D.Woo || !D.hid || !D.gi || !D.go || !D.ci || !D.co ||
            !D.sc || !D.a || (d->use_layer_norm && (!D.lna_i || !D.lnb_i || !D.lna_o || !D.lnb_o || !D.xh_i || !D.xh_o || !D.rs_i || !D.rs_o)))
            return ASR_ERR_ARG;
    }
    if (d->T == 0) return ASR_OK;
    ScalArgs& A = Hh.A;
    const int T = d->T, B = d->B, N = d->N, H = d->H, W = H + N, NS4N = A.NS * 4 * N;
    hipLaunchKernelGGL(scal_init_kernel, dim3(B, d->ndir), dim3(256), 0, st, A, 0);
    ASR_CHECK_LAUNCH();
    for (int step = 0; step < T; ++step) {
        scal_times(d, A, step);
        const float *hprev[2], *ohprev[2], *hrow_c[2], *Wio[2], *Sm[2], *sb[2], *Woo[2];
        float *tg[2], *sc[2], *a[2];
        for (int k = 0; k < 2; ++k) {
            const int kk = k < d->ndir ? k : 0;
            const ScalDir& D = A.d[kk];
            const size_t row0 = (size_t)A.t[kk] * d->Btot + d->b0;
            hprev[k] = A.tp[kk] < 0 ? A.h0 + (size_t)kk * B * W : D.hid + ((size_t)A.tp[kk] * d->Btot + d->b0) * W;
            ohprev[k] = hprev[k] + H;
            hrow_c[k] = D.hid + row0 * W;
            sc[k] = D.sc + row0 * NS4N; a[k] = D.a + row0 * 4 * N;
            tg[k] = A.tg + (size_t)kk * B * 4 * H;
            Wio[k] = D.Wio; Sm[k] = D.S; sb[k] = D.s; Woo[k] = D.Woo;
        }
        if ((rc = scal_mm(Hh, false, hprev, W, Wio, 4 * H, nullptr, tg, 4 * H, W, 4 * H, 0))) return rc;       // :974-975
        if ((rc = scal_mm(Hh, false, ohprev, W, Woo, 4 * N, nullptr, a, 4 * N, N, 4 * N, 0))) return rc;        // :1039
        scal_cell_launch<0>(A, false, st);
        ASR_CHECK_LAUNCH();
        if ((rc = scal_mm(Hh, false, hrow_c, W, Sm, NS4N, sb, sc, NS4N, H, NS4N, 0))) return rc;                // :1030-1033
        scal_cell_launch<1>(A, false, st);
        ASR_CHECK_LAUNCH();
    }
    return ASR_OK;
}

extern "C" int asr_scaling_bwd(const asr_scaling_desc* d, void* ws, size_t ws_bytes, hipStream_t st) {
    ScalHost Hh;
    int rc = scal_setup(d, ws, ws_bytes, st, Hh);
    if (rc) return rc;
    for (int k = 0; k < d->ndir; ++k) {
        const ScalDir& D = d->dir[k];
        if (!D.outer_in || !D.Wio || !D.S || !D.Woo || !D.gi || !D.go || !D.ci || !D.co || !D.sc || !D.a || !D.dhid ||
            !D.dgi || !D.dgo || !D.dsc || !D.da || !D.dh ||
            (d->use_layer_norm && (!D.lna_i || !D.lnb_i || !D.lna_o || !D.lnb_o || !D.xh_i || !D.xh_o || !D.rs_i || !D.rs_o || !D.dy_i || !D.dy_o)))
            return ASR_ERR_ARG;
    }
    if (d->T == 0) return ASR_OK;
    ScalArgs& A = Hh.A;
    const int T = d->T, B = d->B, N = d->N, H = d->H, W = H + N, NS4N = A.NS * 4 * N;
    hipLaunchKernelGGL(scal_init_kernel, dim3(B, d->ndir), dim3(256), 0, st, A, 1);
    ASR_CHECK_LAUNCH();
    // dh starts as dhid (rows of this chunk, every t); each step adds its recurrent part to dh of the step before
    for (int k = 0; k < d->ndir; ++k) {
        const ScalDir& D = d->dir[k];
        if (hipMemcpy2DAsync(D.dh + (size_t)d->b0 * W, (size_t)d->Btot * W * sizeof(float), D.dhid + (size_t)d->b0 * W,
                             (size_t)d->Btot * W * sizeof(float), (size_t)B * W * sizeof(float), T, hipMemcpyDeviceToDevice,
                             st) != hipSuccess)
            return ASR_ERR_LAUNCH;
    }
    for (int step = T - 1; step >= 0; --step) {
        scal_times(d, A, step);
        const float *dsc[2], *dgi[2], *da[2], *Wio[2], *Sm[2], *Woo[2];
        float *dh_i[2], *gh[2], *gh_o[2];
        for (int k = 0; k < 2; ++k) {
            const int kk = k < d->ndir ? k : 0;
            const ScalDir& D = A.d[kk];
            const size_t row0 = (size_t)A.t[kk] * d->Btot + d->b0;
            dh_i[k] = D.dh + row0 * W;
            dsc[k] = D.dsc + row0 * NS4N; dgi[k] = D.dgi + row0 * 4 * H; da[k] = D.da + row0 * 4 * N;
            gh[k] = A.tp[kk] < 0 ? A.dh0 + (size_t)kk * B * W : D.dh + ((size_t)A.tp[kk] * d->Btot + d->b0) * W;
            gh_o[k] = gh[k] + H;
            Wio[k] = D.Wio; Sm[k] = D.S; Woo[k] = D.Woo;
        }
        scal_cell_launch<1>(A, true, st);
        ASR_CHECK_LAUNCH();
        // d inner_hid_t += d scale maps . S^T
        if ((rc = scal_mm(Hh, true, dsc, NS4N, Sm, NS4N, nullptr, dh_i, W, NS4N, H, 1))) return rc;
        scal_cell_launch<0>(A, true, st);
        ASR_CHECK_LAUNCH();
        // d hid_{t-1} += d inner gates . Wio^T ; its outer half += da . Woo^T
        if ((rc = scal_mm(Hh, true, dgi, 4 * H, Wio, 4 * H, nullptr, gh, W, 4 * H, W, 1))) return rc;
        if ((rc = scal_mm(Hh, true, da, 4 * N, Woo, 4 * N, nullptr, gh_o, W, 4 * N, N, 1))) return rc;
    }
    hipLaunchKernelGGL(scal_dinit_kernel, dim3(B, d->ndir), dim3(256), 0, st, A);
    ASR_CHECK_LAUNCH();
    return ASR_OK;
}
