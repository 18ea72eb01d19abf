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

// ------------------------------------------------------------------------------------------------
// The same steps as ONE persistent launch per pass (round 3; the form step_prjhyper.hip's kernels have): the launches of
// a step become phases of a loop inside one kernel, separated by scan_common.h's exchange; one workgroup of 512 threads
// per CU.  The products leave per-split partial blocks (step_skinny.h's persistent bodies, raw mode) that the cell behind
// them sums in split order, bias included where the step loop adds it in the product; what the step loop accumulates
// into `dh` with beta = 1 is summed by the cell that consumes it.  Every phase is
//     loads that do not depend on the exchange before it -> wait -> dependent loads, arithmetic, handed-over stores ->
//     publish -> stores only later kernels read.
// The weights are re-read every step (they are small at the sizes this layer is used at and stay in L2); their loads are
// issued before the wait.  Taken when every wave needs one round of weight loads per product and the blocks of a phase
// fit the grid (scal_persist_ok); the step loop otherwise.  Same expressions in the same order as the cell kernels above.
#define SCP_T 512
struct ScalPersist {
    float *ptg[2], *pa[2], *psc[2];         // forward partial blocks [split][Bpad][4H] / [4N] / [NS 4N]
    float *pb1[2], *pgh[2], *pgo[2];        // backward [split][Bpad][H] / [H+N] / [N]
    int ns_tg, ns_a, ns_sc, ns_b1, ns_gh, ns_go;
    int Bpad;
};

template <int K>
__device__ __forceinline__ void block_sumK8(float (&v)[K], float* sm) {      // 8 waves; sm: 8 K floats
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
    for (int k = 0; k < K; ++k)
        v[k] = (sm[k] + sm[K + k] + sm[2 * K + k] + sm[3 * K + k]) + (sm[4 * K + k] + sm[5 * K + k] + sm[6 * K + k] + sm[7 * K + k]);
}

template <int SL, class F>
__device__ __forceinline__ void scal_part_sum_sl(int ns, F f) {
    for (int sp0 = 0; sp0 < ns; sp0 += SL) {
#pragma unroll
        for (int i4 = 0; i4 < SL; ++i4) f(min(sp0 + i4, ns - 1), sp0 + i4 < ns ? 1.f : 0.f);
    }
}
// sum of ns partial blocks in split order, every load unconditional, as many in flight as there are splits (up to 4)
template <class F>
__device__ __forceinline__ void scal_part_sum(int ns, F f) {
    if (ns == 1) scal_part_sum_sl<1>(ns, f);
    else if (ns == 2) scal_part_sum_sl<2>(ns, f);
    else scal_part_sum_sl<4>(ns, f);
}

template <int OUTER, int NU, class WaitF, class PubF>
__device__ __forceinline__ bool scal_cell_fwd_pbody(const ScalArgs& A, const ScalPersist& Q, int b, int dir, int t, int tp,
                                                    float* sm, WaitF waitf, PubF pubf) {
    const ScalDir& D = A.d[dir];
    const int H = A.H, N = A.N, U = OUTER ? N : H, W = H + N, off = OUTER ? H : 0, NS4 = A.NS * 4;
    const size_t row = (size_t)t * A.Btot + A.b0 + b, rowp = (size_t)(tp < 0 ? 0 : tp) * A.Btot + A.b0 + b;
    const size_t wr = (size_t)dir * A.B + b;
    const bool on = A.mask[(size_t)(A.b0 + b) * A.msb + (size_t)t * A.mst] != 0.f;
    const float* cprev_p = tp < 0 ? A.c0[OUTER] + wr * U : (OUTER ? D.co : D.ci) + rowp * U;
    const float* hprev_p = tp < 0 ? A.h0 + wr * W + off : D.hid + rowp * W + off;
    const bool virgin = !OUTER && D.tg0 && (tp < 0 || A.mask[(size_t)(A.b0 + b) * A.msb + (size_t)tp * A.mst] == 0.f);
    const float* lna = OUTER ? D.lna_o : D.lna_i;
    const float* lnb = OUTER ? D.lnb_o : D.lnb_i;
    float* gsv = (OUTER ? D.go : D.gi) + row * 4 * U;
    float* xh_sv = (OUTER ? D.xh_o : D.xh_i);
    int nn[NU];
    float pre[NU][4], cp[NU], hpv[NU], pk[NU][4], xin[NU][4], rec[NU][4], scv[NU][16], t0v[NU][4];
#pragma unroll
    for (int u = 0; u < NU; ++u) {                   // ---- independent of the exchange
        const int n = nn[u] = min((int)threadIdx.x + u * SCP_T, U - 1);
        cp[u] = ldx<true>(cprev_p + n);
        hpv[u] = ldx<true>(hprev_p + n);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            rec[u][k] = 0.f;
            xin[u][k] = OUTER ? D.outer_in[row * 4 * N + k * N + n] : D.inner_in[row * 4 * H + k * H + n];
            pk[u][k] = OUTER ? 0.f : (D.wpeep ? D.wpeep[k * H + n] : 0.f);
            t0v[u][k] = (!OUTER && virgin) ? D.tg0[k * H + n] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) scv[u][k] = (OUTER && k < NS4) ? D.s[k * N + n] : 0.f;
    }
    if (!waitf()) return false;
    if (OUTER) {                                     // ---- sums of the partial blocks
        const float* pa = Q.pa[dir] + (size_t)b * 4 * N;
        scal_part_sum(Q.ns_a, [&](int sp, float pw) {
            const float* pp = pa + (size_t)sp * Q.Bpad * 4 * N;
#pragma unroll
            for (int u = 0; u < NU; ++u)
#pragma unroll
                for (int k = 0; k < 4; ++k) rec[u][k] += pw * ldx<true>(pp + k * N + nn[u]);
        });
        const float* ps = Q.psc[dir] + (size_t)b * NS4 * N;
        scal_part_sum(Q.ns_sc, [&](int sp, float pw) {
            const float* pp = ps + (size_t)sp * Q.Bpad * NS4 * N;
#pragma unroll
            for (int u = 0; u < NU; ++u)
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    if (k < 12 || A.NS == 4) scv[u][k] += pw * ldx<true>(pp + k * N + nn[u]);
        });
    } else {
        const float* pt = Q.ptg[dir] + (size_t)b * 4 * H;
        scal_part_sum(Q.ns_tg, [&](int sp, float pw) {
            const float* pp = pt + (size_t)sp * Q.Bpad * 4 * H;
#pragma unroll
            for (int u = 0; u < NU; ++u)
#pragma unroll
                for (int k = 0; k < 4; ++k) rec[u][k] += pw * ldx<true>(pp + k * H + nn[u]);
        });
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (OUTER) {
                pre[u][k] = xin[u][k] * scv[u][k] + rec[u][k] * scv[u][4 + k] + scv[u][8 + k];     // :1038-1040
                pk[u][k] = A.NS == 4 ? scv[u][12 + k] : 0.f;
            }
        }
    }
    if (!OUTER) {
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                pre[u][k] = xin[u][k] + rec[u][k];                                                  // :973-975
                if (virgin) pre[u][k] += t0v[u][k];
            }
    }
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int k = 0; k < 3; ++k) pre[u][k] += cp[u] * pk[u][k];                                  // :977-979 | :1042-1044
    float rstd[5] = {1.f, 1.f, 1.f, 1.f, 1.f};
    float xh[NU][5];
    if (A.use_ln) {                                                                                 // :981-990 | :1046-1055
        float s1[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < NU; ++u)
            if ((int)threadIdx.x + u * SCP_T < U) { s1[0] += pre[u][0]; s1[1] += pre[u][1]; s1[2] += pre[u][2]; }
        block_sumK8<3>(s1, sm);
        float s2[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < NU; ++u)
            if ((int)threadIdx.x + u * SCP_T < U) {
#pragma unroll
                for (int k = 0; k < 3; ++k) { const float dl = pre[u][k] - s1[k] / U; s2[k] += dl * dl; }
            }
        block_sumK8<3>(s2, sm);
#pragma unroll
        for (int k = 0; k < 3; ++k) rstd[k] = 1.0f / sqrtf(s2[k] / U + ASR_EPS);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                xh[u][k] = (pre[u][k] - s1[k] / U) * rstd[k];
                pre[u][k] = lna[k * U + nn[u]] * xh[u][k] + lnb[k * U + nn[u]];
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
            if ((int)threadIdx.x + u * SCP_T < U) { s1[0] += pre[u][3]; s1[1] += cn[u]; }
        block_sumK8<2>(s1, sm);
        float s2[2] = {0.f, 0.f};
#pragma unroll
        for (int u = 0; u < NU; ++u)
            if ((int)threadIdx.x + u * SCP_T < U) {
                const float d0 = pre[u][3] - s1[0] / U, d1 = cn[u] - s1[1] / U;
                s2[0] += d0 * d0; s2[1] += d1 * d1;
            }
        block_sumK8<2>(s2, sm);
        rstd[3] = 1.0f / sqrtf(s2[0] / U + ASR_EPS); rstd[4] = 1.0f / sqrtf(s2[1] / U + ASR_EPS);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            xh[u][3] = (pre[u][3] - s1[0] / U) * rstd[3];
            xh[u][4] = (cn[u] - s1[1] / U) * rstd[4];
            pre[u][3] = lna[3 * U + nn[u]] * xh[u][3] + lnb[3 * U + nn[u]];
            oc[u] = lna[4 * U + nn[u]] * xh[u][4] + lnb[4 * U + nn[u]];
        }
    }
    float og[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int n = threadIdx.x + u * SCP_T;
        og[u] = sigmoidf_(pre[u][3]);
        const float h = og[u] * tanhf_(oc[u]);                                                      // :1025 | :1091
        if (n < U) {
            stx<true>((OUTER ? D.co : D.ci) + row * U + n, on ? cn[u] : cp[u]);                       // :1113-1116
            stx<true>(D.hid + row * W + off + n, on ? h : hpv[u]);
        }
    }
    pubf();
    if (A.use_ln && threadIdx.x < 5) (OUTER ? D.rs_o : D.rs_i)[row * 5 + threadIdx.x] = rstd[threadIdx.x];
#pragma unroll
    for (int u = 0; u < NU; ++u) {                   // ---- read by later kernels only
        const int n = threadIdx.x + u * SCP_T;
        if (n >= U) continue;
        gsv[n] = ig[u]; gsv[U + n] = fg[u]; gsv[2 * U + n] = gc[u]; gsv[3 * U + n] = og[u];
        if (A.use_ln) {
#pragma unroll
            for (int k = 0; k < 5; ++k) xh_sv[row * 5 * U + k * U + n] = xh[u][k];
        }
        if (OUTER) {
#pragma unroll
            for (int k = 0; k < 4; ++k) D.a[row * 4 * N + k * N + n] = rec[u][k];
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (k < NS4) D.sc[row * NS4 * N + k * N + n] = scv[u][k];
        }
    }
    return true;
}

template <int OUTER, int NU, class WaitF, class PubF>
__device__ __forceinline__ bool scal_cell_bwd_pbody(const ScalArgs& A, const ScalPersist& Q, int b, int dir, int t, int tp,
                                                    bool first, float* sm, WaitF waitf, PubF pubf) {
    const ScalDir& D = A.d[dir];
    const int H = A.H, N = A.N, U = OUTER ? N : H, W = H + N, off = OUTER ? H : 0, NS4 = A.NS * 4;
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
    int nn[NU];
    float dht[NU], gc0[NU], dhp[NU], g4[NU][4], cp[NU], cn[NU], pk[NU][4], xh[NU][5], al[NU][5], lb4[NU];
    float s_in[NU][4], s_hid[NU][4], x_in[NU][4], x_a[NU][4];
#pragma unroll
    for (int u = 0; u < NU; ++u) {                   // ---- independent of the exchange
        const int n = nn[u] = min((int)threadIdx.x + u * SCP_T, U - 1);
        dht[u] = ldx<true>(D.dh + row * W + off + n);      // dhid, plus what switched-off rows of later steps passed on
        gc0[u] = ldx<true>(g_c + n);
        dhp[u] = ldx<true>(dhprev_p + n);
        cp[u] = cprev_p[n]; cn[u] = cnew_p[n];
        lb4[u] = A.use_ln ? lnb[4 * U + n] : 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            g4[u][k] = gsv[k * U + n];
            if (OUTER) {
                const float* sc = D.sc + row * NS4 * N;
                pk[u][k] = A.NS == 4 ? sc[12 * N + k * N + n] : 0.f;
                s_in[u][k] = sc[k * N + n]; s_hid[u][k] = sc[4 * N + k * N + n];
                x_in[u][k] = D.outer_in[row * 4 * N + k * N + n]; x_a[u][k] = D.a[row * 4 * N + k * N + n];
            } else {
                pk[u][k] = D.wpeep ? D.wpeep[k * H + n] : 0.f;
            }
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            xh[u][k] = A.use_ln ? xh_sv[k * U + n] : 0.f;
            al[u][k] = A.use_ln ? lna[k * U + n] : 1.f;
        }
    }
    if (!waitf()) return false;
    if (!first) {                                    // ---- what the products of the step before handed to this row
        const float* pg = Q.pgh[dir] + (size_t)b * W + off;
        scal_part_sum(Q.ns_gh, [&](int sp, float pw) {
#pragma unroll
            for (int u = 0; u < NU; ++u) dht[u] += pw * ldx<true>(pg + (size_t)sp * Q.Bpad * W + nn[u]);
        });
        if (OUTER) {
            const float* po = Q.pgo[dir] + (size_t)b * N;
            scal_part_sum(Q.ns_go, [&](int sp, float pw) {
#pragma unroll
                for (int u = 0; u < NU; ++u) dht[u] += pw * ldx<true>(po + (size_t)sp * Q.Bpad * N + nn[u]);
            });
        }
    }
    if (!OUTER) {                                    // d inner_hid_t += d scale maps . S^T (this step)
        const float* p1 = Q.pb1[dir] + (size_t)b * H;
        scal_part_sum(Q.ns_b1, [&](int sp, float pw) {
#pragma unroll
            for (int u = 0; u < NU; ++u) dht[u] += pw * ldx<true>(p1 + (size_t)sp * Q.Bpad * H + nn[u]);
        });
    }
    float dh[NU], gcar[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const bool act = on && (int)threadIdx.x + u * SCP_T < U;
        dh[u] = act ? dht[u] : 0.f;
        gcar[u] = act ? gc0[u] : 0.f;
    }
    // gradient at the (clipped) outputs of the out-gate and out-cell layer norms, then through them
    float d[NU][5];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const float oc = A.use_ln ? al[u][4] * xh[u][4] + lb4[u] : cn[u];
        const float tc = tanhf_(oc), o = g4[u][3];
        d[u][3] = clipf_(dh[u] * tc * o * (1.f - o), A.clip);                                       // :1016-1019 | :1083-1086
        d[u][4] = clipf_(dh[u] * o * (1.f - tc * tc), A.clip);                                      // :1020-1022 | :1087-1089
    }
    float dpre[NU][5];
    if (A.use_ln) {
        float s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            s4[0] += d[u][3] * al[u][3]; s4[1] += d[u][3] * al[u][3] * xh[u][3];
            s4[2] += d[u][4] * al[u][4]; s4[3] += d[u][4] * al[u][4] * xh[u][4];
        }
        block_sumK8<4>(s4, sm);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            dpre[u][3] = rstd[3] * (d[u][3] * al[u][3] - s4[0] / U - xh[u][3] * s4[1] / U);
            dpre[u][4] = rstd[4] * (d[u][4] * al[u][4] - s4[2] / U - xh[u][4] * s4[3] / U);
        }
    } else {
#pragma unroll
        for (int u = 0; u < NU; ++u) { dpre[u][3] = d[u][3]; dpre[u][4] = d[u][4]; }
    }
    float dc[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const bool act = on && (int)threadIdx.x + u * SCP_T < U;
        dc[u] = act ? gcar[u] + dpre[u][4] + dpre[u][3] * pk[u][3] : 0.f;
        const float ig = g4[u][0], fg = g4[u][1], gc = g4[u][2];
        d[u][0] = clipf_(dc[u] * gc * ig * (1.f - ig), A.clip);                                     // :992-1001 | :1057-1066
        d[u][1] = clipf_(dc[u] * cp[u] * fg * (1.f - fg), A.clip);
        d[u][2] = clipf_(dc[u] * ig * (1.f - gc * gc), A.clip);
    }
    if (A.use_ln) {
        float s6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int k = 0; k < 3; ++k) { s6[2 * k] += d[u][k] * al[u][k]; s6[2 * k + 1] += d[u][k] * al[u][k] * xh[u][k]; }
        block_sumK8<6>(s6, sm);
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int k = 0; k < 3; ++k)
                dpre[u][k] = rstd[k] * (d[u][k] * al[u][k] - s6[2 * k] / U - xh[u][k] * s6[2 * k + 1] / U);
    } else {
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int k = 0; k < 3; ++k) dpre[u][k] = d[u][k];
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {                   // ---- what the products of this step read
        const int n = threadIdx.x + u * SCP_T;
        if (n >= U) continue;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float dg = dpre[u][k];
            const size_t i4 = row * 4 * U + k * U + n;
            if (OUTER) {
                stx<true>(D.da + i4, dg * s_hid[u][k]);                                            // d (outer_hid_prev.W)
                float* ds = D.dsc + row * NS4 * N;
                stx<true>(ds + k * N + n, dg * x_in[u][k]);
                stx<true>(ds + 4 * N + k * N + n, dg * x_a[u][k]);
                stx<true>(ds + 8 * N + k * N + n, dg);
                if (A.NS == 4) stx<true>(ds + 12 * N + k * N + n, k == 2 ? 0.f : dg * (k == 3 ? cn[u] : cp[u]));   // no map for the cell gate
            } else {
                stx<true>(D.dgi + i4, dg);
            }
        }
    }
    pubf();
#pragma unroll
    for (int u = 0; u < NU; ++u) {                   // ---- this workgroup's own carry, and what later kernels read
        const int n = threadIdx.x + u * SCP_T;
        if (n >= U) continue;
        if (on) stx<true>(g_c + n, dc[u] * g4[u][1] + dpre[u][0] * pk[u][0] + dpre[u][1] * pk[u][1] + dpre[u][2] * pk[u][2]);
        else stx<true>(dhprev_p + n, dhp[u] + dht[u]);   // switched off: the state was copied, its gradient passes on
        if (A.use_ln) {
#pragma unroll
            for (int k = 0; k < 5; ++k) dy_sv[row * 5 * U + k * U + n] = d[u][k];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (OUTER) D.dgo[row * 4 * U + k * U + n] = dpre[u][k] * s_in[u][k];                     // d outer_input
            else if (first_on) D.dgi0[(size_t)(A.b0 + b) * 4 * H + k * H + n] = dpre[u][k];
        }
        if (!OUTER && D.dpeep) {
            float* dp = D.dpeep + row * 3 * H;
            dp[n] = dpre[u][0] * cp[u]; dp[H + n] = dpre[u][1] * cp[u]; dp[2 * H + n] = dpre[u][3] * cn[u];
        }
    }
    return true;
}

#define SCP_LDS_BYTES (82 * 1024)            /* more than half a CU's LDS: one workgroup per CU */
#define SCP_SETUP \
    extern __shared__ __attribute__((aligned(16))) float smem[]; \
    const int half = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8); \
    float* sm = smem + 2 * 3 * 4 * 64 * 4; \
    int* flag = (int*)(sm + 64); \
    GroupSync gs; \
    const int wg = blockIdx.x, G = gridDim.x; \
    group_sync_init(gs, sync, G, wg, flag); \
    bool pending = false;                    /* a publish whose wait is still to come (uniform over the grid) */ \
    auto waitf = [&]() -> bool { if (!pending) return true; pending = false; return group_wait(gs); }; \
    auto pubf = [&]() { group_publish(gs); pending = true; }; \
    const int H = A.H, N = A.N, W = H + N, NS4N = A.NS * 4 * N, npz = A.ndir * ((A.B + 15) / 16); \
    const int cb = wg % A.B, cd = wg / A.B;  /* this workgroup's cell (workgroups >= B ndir have none) */ \
    const bool cell = wg < A.B * A.ndir; \
    const int v = 2 * wg + half;             /* this half-workgroup's block in a product phase */

__device__ __forceinline__ void scal_sk_args(SkinnyArgs& S, const ScalArgs& A, const float* const Wm[2], int lda, int ldw,
                                             float* const Cp[2], int K, int Nc, int ns, bool nt) {
#pragma unroll
    for (int k = 0; k < SK_MAXP; ++k) { S.p[k].A = nullptr; S.p[k].W = Wm[k]; S.p[k].bias = nullptr; S.p[k].C = Cp[k]; }
    S.lda = lda; S.ldw = ldw; S.ldc = Nc; S.Bm = A.B; S.K = K; S.Nc = Nc; S.beta = 0; S.nsplit = ns; S.raw = 1;
    S.nchunk = (A.B + 15) / 16;
    S.zfrom = nt ? 0 : 0x7fffffff; S.zk0 = 0; S.zk1 = 0;
    S.part = nullptr; S.tickets = nullptr;
}
// the product of a two-product phase this half-workgroup works for: b's fields where `second`, a's otherwise (wave-uniform)
__device__ __forceinline__ SkinnyArgs scal_sk_pick(const SkinnyArgs& a, const SkinnyArgs& b, bool second) {
    SkinnyArgs S = a;
#pragma unroll
    for (int k = 0; k < SK_MAXP; ++k) { S.p[k].W = second ? b.p[k].W : a.p[k].W; S.p[k].C = second ? b.p[k].C : a.p[k].C; }
    S.lda = second ? b.lda : a.lda; S.ldw = second ? b.ldw : a.ldw; S.ldc = second ? b.ldc : a.ldc;
    S.K = second ? b.K : a.K; S.Nc = second ? b.Nc : a.Nc; S.nsplit = second ? b.nsplit : a.nsplit;
    return S;
}

template <int NU>
__global__ __launch_bounds__(SCP_T) void scal_persist_fwd_kernel(ScalArgs A, ScalPersist Q, unsigned* sync) {
    SCP_SETUP
    f32x4 (*red)[4][64] = (f32x4 (*)[4][64])smem + half * 3;
    const float *Wio[2] = {A.d[0].Wio, A.d[1].Wio}, *Woo[2] = {A.d[0].Woo, A.d[1].Woo}, *Sm[2] = {A.d[0].S, A.d[1].S};
    SkinnyArgs Sa, Sb, Sc;
    scal_sk_args(Sa, A, Wio, W, 4 * H, Q.ptg, W, 4 * H, Q.ns_tg, false);      // tg = hid_prev . Wio          :974-975
    scal_sk_args(Sb, A, Woo, W, 4 * N, Q.pa, N, 4 * N, Q.ns_a, false);        // a = outer_hid_prev . Woo     :1039
    scal_sk_args(Sc, A, Sm, W, NS4N, Q.psc, H, NS4N, Q.ns_sc, false);         // scale maps = inner_hid . S   :1030-1033
    const int nb_a = (4 * H + 63) / 64, nb_b = (4 * N + 63) / 64, nb_c = (NS4N + 63) / 64;
    const int tot_a = nb_a * Q.ns_tg * npz, tot_b = nb_b * Q.ns_a * npz, tot_c = nb_c * Q.ns_sc * npz;
    // phase 0: blocks [0, tot_a) of the first product, [tot_a, tot_a + tot_b) of the second
    const bool sec = v >= tot_a, l0 = v < tot_a + tot_b;
    const int v0 = sec ? v - tot_a : v, nb0 = sec ? nb_b : nb_a, ns0 = sec ? Q.ns_a : Q.ns_tg;
    SkinnyArgs S0 = scal_sk_pick(Sa, Sb, sec);
    const int bx0 = v0 % nb0, by0 = (v0 / nb0) % ns0, bz0 = v0 / (nb0 * ns0);
    const SkinnyNNCtx c0 = skinny_nn_ctx(S0, bx0, by0, bz0, l0);
    const bool l2 = v < tot_c;
    const int bx2 = v % nb_c, by2 = (v / nb_c) % Q.ns_sc, bz2 = v / (nb_c * Q.ns_sc);
    const SkinnyNNCtx c2 = skinny_nn_ctx(Sc, bx2, by2, bz2, l2);
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
            const float* hprev = tp[k] < 0 ? A.h0 + (size_t)k * A.B * W : A.d[k].hid + ((size_t)tp[k] * A.Btot + A.b0) * W;
            S0.p[k].A = sec ? hprev + H : hprev;
            Sc.p[k].A = A.d[k].hid + ((size_t)t[k] * A.Btot + A.b0) * W;
        }
        {
            f32x4 w[16];
            skinny_nn_wload<16>(S0, c0, bz0, c0.g0, w);
            ok = skinny_nn_pbody<16>(S0, c0, by0, bz0, [&](int i) { return w[i]; }, red, l0, waitf);
            __syncthreads();
            if (!ok) break;
            pubf();
        }
        if (cell) ok = scal_cell_fwd_pbody<0, NU>(A, Q, cb, cd, cd ? t[1] : t[0], cd ? tp[1] : tp[0], sm, waitf, pubf);
        else { ok = waitf(); pubf(); }
        if (!ok) break;
        {
            f32x4 w[16];
            skinny_nn_wload<16>(Sc, c2, bz2, c2.g0, w);
            ok = skinny_nn_pbody<16>(Sc, c2, by2, bz2, [&](int i) { return w[i]; }, red, l2, waitf);
            __syncthreads();
            if (!ok) break;
            pubf();
        }
        if (cell) ok = scal_cell_fwd_pbody<1, NU>(A, Q, cb, cd, cd ? t[1] : t[0], cd ? tp[1] : tp[0], sm, waitf, pubf);
        else { ok = waitf(); pubf(); }
    }
}

template <int NU>
__global__ __launch_bounds__(SCP_T) void scal_persist_bwd_kernel(ScalArgs A, ScalPersist Q, unsigned* sync) {
    SCP_SETUP
    f32x4 (*red)[64] = (f32x4 (*)[64])smem + half * 3;
    const float *Wio[2] = {A.d[0].Wio, A.d[1].Wio}, *Woo[2] = {A.d[0].Woo, A.d[1].Woo}, *Sm[2] = {A.d[0].S, A.d[1].S};
    SkinnyArgs S1, Sg, So;
    scal_sk_args(S1, A, Sm, NS4N, NS4N, Q.pb1, NS4N, H, Q.ns_b1, true);       // d inner_hid_t += d scale maps . S^T
    scal_sk_args(Sg, A, Wio, 4 * H, 4 * H, Q.pgh, 4 * H, W, Q.ns_gh, true);   // d hid_{t-1} += d inner gates . Wio^T
    scal_sk_args(So, A, Woo, 4 * N, 4 * N, Q.pgo, 4 * N, N, Q.ns_go, true);   // its outer half += da . Woo^T
    const int nb_1 = (H + 15) / 16, nb_g = (W + 15) / 16, nb_o = (N + 15) / 16;
    const int tot_1 = nb_1 * Q.ns_b1 * npz, tot_g = nb_g * Q.ns_gh * npz, tot_o = nb_o * Q.ns_go * npz;
    const bool l1 = v < tot_1;
    const int bx1 = v % nb_1, by1 = (v / nb_1) % Q.ns_b1, bz1 = v / (nb_1 * Q.ns_b1);
    const SkinnyNTCtx c1 = skinny_nt_ctx(S1, bx1, by1, bz1, l1);
    const bool sec = v >= tot_g, l3 = v < tot_g + tot_o;
    const int v3 = sec ? v - tot_g : v, nb3 = sec ? nb_o : nb_g, ns3 = sec ? Q.ns_go : Q.ns_gh;
    SkinnyArgs S3 = scal_sk_pick(Sg, So, sec);
    const int bx3 = v3 % nb3, by3 = (v3 / nb3) % ns3, bz3 = v3 / (nb3 * ns3);
    const SkinnyNTCtx c3 = skinny_nt_ctx(S3, bx3, by3, bz3, l3);
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
            S1.p[k].A = A.d[k].dsc + row0 * NS4N;
            S3.p[k].A = sec ? A.d[k].da + row0 * 4 * N : A.d[k].dgi + row0 * 4 * H;
        }
        if (cell) ok = scal_cell_bwd_pbody<1, NU>(A, Q, cb, cd, cd ? t[1] : t[0], cd ? tp[1] : tp[0], first, sm, waitf, pubf);
        else { ok = waitf(); pubf(); }
        if (!ok) break;
        {
            f32x4 w[16];
            skinny_nt_wload<4>(S1, c1, bz1, c1.c0, w);
            ok = skinny_nt_pbody<4>(S1, c1, by1, bz1, [&](int i) { return w[i]; }, red, l1, waitf);
            __syncthreads();
            if (!ok) break;
            pubf();
        }
        if (cell) ok = scal_cell_bwd_pbody<0, NU>(A, Q, cb, cd, cd ? t[1] : t[0], cd ? tp[1] : tp[0], first, sm, waitf, pubf);
        else { ok = waitf(); pubf(); }
        if (!ok) break;
        {
            f32x4 w[16];
            skinny_nt_wload<4>(S3, c3, bz3, c3.c0, w);
            ok = skinny_nt_pbody<4>(S3, c3, by3, bz3, [&](int i) { return w[i]; }, red, l3, waitf);
            __syncthreads();
            if (!ok) break;
            pubf();
        }
    }
    // what the last step's products hand to the state before the first step (the step loop accumulates it into dh0)
    if (ok && A.T > 0 && waitf() && cell) {
        const size_t wr = (size_t)cd * A.B + cb;
        for (int i = threadIdx.x; i < W; i += SCP_T) {
            float acc = ldx<true>(A.dh0 + wr * W + i);
            for (int sp = 0; sp < Q.ns_gh; ++sp) acc += ldx<true>(Q.pgh[cd] + ((size_t)sp * Q.Bpad + cb) * W + i);
            if (i >= H)
                for (int sp = 0; sp < Q.ns_go; ++sp) acc += ldx<true>(Q.pgo[cd] + ((size_t)sp * Q.Bpad + cb) * N + i - H);
            A.dh0[wr * W + i] = acc;
        }
    }
}

struct asr_scaling_desc {       // mirrored in include/asr_amd.h
    int T, B, N, H, ndir;
    int b0, Btot;               // rows [b0, b0+B) of Btot; B <= 64 per call
    int use_peepholes, use_layer_norm;
    float grad_clip;
    const float* mask; int mask_stride_b, mask_stride_t;
    ScalDir dir[2];
};

// ---- persistent launch: splits, eligibility, workspace
static int scal_mode_ = 0;
extern "C" int asr_scaling_set_mode(int mode) { const int old = scal_mode_; scal_mode_ = mode ? 1 : 0; return old; }

struct ScalSplits { int tg, a, sc, b1, gh, go; };
// the fewest splits with ONE round of weight loads per wave (16 k-groups of 4 / 4 chunks of 64): the cells sum the splits
static ScalSplits scal_splits(int H, int N, int NS) {
    auto nn = [](int K) { const int g = (K + 3) / 4; return (g + 63) / 64; };
    auto nt = [](int K) { const int c = (K + 63) / 64; return (c + 15) / 16; };
    ScalSplits s;
    s.tg = nn(H + N); s.a = nn(N); s.sc = nn(H);
    s.b1 = nt(NS * 4 * N); s.gh = nt(4 * H); s.go = nt(4 * N);
    return s;
}
// workgroups of the persistent launch for this shape (one per CU), 0: step loop
static int scal_use_persist(int B, int N, int H, int ndir) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("ASR_SCAL_PERSIST"); on = e ? atoi(e) != 0 : 1; }
    if (!on || scal_mode_ || N > SCP_T || H > SCP_T) return 0;    // one unit per thread (two spill)
    int dev = 0, g = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&g, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || g < 16) return 0;
    if (g < B * ndir) return 0;                               // every cell needs a workgroup of its own
    const ScalSplits s = scal_splits(H, N, 4);
    const int npz = ndir * ((B + 15) / 16), W = H + N;
    auto cd = [](int a, int b) { return (a + b - 1) / b; };
    const int f0 = (cd(4 * H, 64) * s.tg + cd(4 * N, 64) * s.a) * npz, f2 = cd(16 * N, 64) * s.sc * npz;
    const int b1 = cd(H, 16) * s.b1 * npz, b3 = (cd(W, 16) * s.gh + cd(N, 16) * s.go) * npz;
    if (!(f0 <= 2 * g && f2 <= 2 * g && b1 <= 2 * g && b3 <= 2 * g)) return 0;
    // as many workgroups as the widest phase has work for (a cell each, two product blocks each), not the whole chip: a
    // small layer leaves the other CUs to whatever else runs on the GPU (step_prjhyper.hip, prj_persist_wgs)
    int m = f0 > f2 ? f0 : f2;
    m = m > b1 ? m : b1; m = m > b3 ? m : b3;
    int w = cd(m, 2);
    if (w < B * ndir) w = B * ndir;
    return w < g ? w : g;
}
#define SCP_SYNC_BYTES ((size_t)((SYNC_STRIDE * 4 + 255) / 256) * 256)
static size_t scal_part_floats(int B, int N, int H, int ndir) {
    const ScalSplits s = scal_splits(H, N, 4);
    const size_t bp = (size_t)((B + 15) / 16) * 16;
    const size_t fw = bp * ((size_t)s.tg * 4 * H + (size_t)s.a * 4 * N + (size_t)s.sc * 16 * N);
    const size_t bw = bp * ((size_t)s.b1 * H + (size_t)s.gh * (H + N) + (size_t)s.go * N);
    return (fw > bw ? fw : bw) * ndir;
}

// workspace: [sync block of the persistent launch (asr_scan_status(ws, 1) reads it) | state rows | step-loop partials and
// tickets | persistent launch's partial blocks]
extern "C" size_t asr_scaling_ws_bytes(int B, int N, int H, int ndir) {
    return SCP_SYNC_BYTES + ((size_t)ndir * B * (4 * H + 2 * (H + N) + 2 * (H + N)) + SK_PART_FLOATS) * sizeof(float) +
           SK_TICKETS * sizeof(unsigned) + 256 + scal_part_floats(B, N, H, ndir) * sizeof(float) + 256;
}

struct ScalHost {
    ScalArgs A;
    float* part; unsigned* tickets;
    float* ppart;               // partial blocks of the persistent launch
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
    float* p = (float*)((char*)ws + SCP_SYNC_BYTES);
    const size_t nb = (size_t)d->ndir * d->B;
    A.tg = p; p += nb * 4 * d->H;
    A.g_c[0] = p; p += nb * d->H; A.g_c[1] = p; p += nb * d->N;
    A.c0[0] = p; p += nb * d->H; A.c0[1] = p; p += nb * d->N;
    A.h0 = p; p += nb * (d->H + d->N);
    A.dh0 = p; p += nb * (d->H + d->N);
    Hh.part = p; p += SK_PART_FLOATS;
    Hh.tickets = (unsigned*)p;
    Hh.ppart = (float*)(((uintptr_t)(Hh.tickets + SK_TICKETS) + 255) & ~(uintptr_t)255);
    Hh.nd = d->ndir; Hh.st = st;
    if (hipMemsetAsync(Hh.tickets, 0, SK_TICKETS * sizeof(unsigned), st) != hipSuccess) return ASR_ERR_LAUNCH;
    if (hipMemsetAsync(ws, 0, SCP_SYNC_BYTES, st) != hipSuccess) return ASR_ERR_LAUNCH;
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

static int scal_persist_launch(const ScalHost& Hh, void* ws, bool backward, int G) {
    const ScalArgs& A = Hh.A;
    const int H = A.H, N = A.N, W = H + N, NS4N = A.NS * 4 * N;
    const ScalSplits sp = scal_splits(H, N, A.NS);
    ScalPersist Q;
    Q.ns_tg = sp.tg; Q.ns_a = sp.a; Q.ns_sc = sp.sc; Q.ns_b1 = sp.b1; Q.ns_gh = sp.gh; Q.ns_go = sp.go;
    const size_t bp = (size_t)((A.B + 15) / 16) * 16;
    Q.Bpad = (int)bp;
    const size_t fw = bp * ((size_t)sp.tg * 4 * H + (size_t)sp.a * 4 * N + (size_t)sp.sc * NS4N);
    const size_t bw = bp * ((size_t)sp.b1 * H + (size_t)sp.gh * W + (size_t)sp.go * N);
    for (int k = 0; k < 2; ++k) {
        const int kk = k < A.ndir ? k : 0;
        float* q = Hh.ppart + (size_t)kk * (backward ? bw : fw);
        Q.ptg[k] = q; Q.pa[k] = q + bp * sp.tg * 4 * H; Q.psc[k] = Q.pa[k] + bp * sp.a * 4 * N;
        Q.pb1[k] = q; Q.pgh[k] = q + bp * sp.b1 * H; Q.pgo[k] = Q.pgh[k] + bp * sp.gh * W;
    }
    if (N > SCP_T || H > SCP_T) return ASR_ERR_LIMIT;
    const void* fn = backward ? (const void*)scal_persist_bwd_kernel<1> : (const void*)scal_persist_fwd_kernel<1>;
    if (!scan_grid_fits((size_t)G, SCP_LDS_BYTES)) return ASR_ERR_LIMIT;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, SCP_LDS_BYTES) != hipSuccess) return ASR_ERR_LAUNCH;
    unsigned* sync = (unsigned*)ws;
    void* kargs[] = {(void*)&A, (void*)&Q, (void*)&sync};
    if (hipLaunchKernel(fn, dim3(G), dim3(SCP_T), kargs, SCP_LDS_BYTES, Hh.st) != hipSuccess) return ASR_ERR_LAUNCH;
    ASR_CHECK_LAUNCH();
    return ASR_OK;
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
    if (const int G = scal_use_persist(B, N, H, d->ndir)) return scal_persist_launch(Hh, ws, false, G);
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
    if (const int G = scal_use_persist(B, N, H, d->ndir)) {
        if ((rc = scal_persist_launch(Hh, ws, true, G))) return rc;
        hipLaunchKernelGGL(scal_dinit_kernel, dim3(B, d->ndir), dim3(256), 0, st, A);
        ASR_CHECK_LAUNCH();
        return ASR_OK;
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
