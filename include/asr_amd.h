/* asr_amd C-ABI  --  MI355X (gfx950) kernels for the acoustic-model hot path of gunkisu/asr.
 *
 * The reference has NO native/FFI interface for this path: all arithmetic goes through Theano ops
 * (T.dot / theano.scan / theano.grad) called from Python layer classes, so every entry point below
 * cites the reference Python it replaces (paths relative to the reference root).  The boundary a
 * reference maintainer binds is Python -> ctypes.
 *
 * Conventions: all pointers are DEVICE pointers owned by the caller; all tensors float32, row-major;
 * sequences are time-major (T,B,...) with rows r = t*B + b;  "4N" gate vectors are stored in the
 * engine's gate-interleaved column order  col = unit*4 + gate  (gate order ingate, forgetgate,
 * cell, outgate: libs/lhuclstm_layers.py:58-60).  Every function takes the HIP stream to enqueue on,
 * returns 0 or a negative ASR_ERR_* code, never throws, never allocates (workspaces are sized by the
 * *_ws_bytes helpers and passed in).  Thread-compatible: concurrent calls must use different
 * workspaces; two persistent scans must never run concurrently on one device.
 */
#ifndef ASR_AMD_H
#define ASR_AMD_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* asr_stream_t; /* hipStream_t */

#define ASR_OK 0
#define ASR_ERR_ARG (-1)
#define ASR_ERR_LAUNCH (-2)
#define ASR_ERR_WS (-3)
#define ASR_ERR_TIMEOUT (-4)
#define ASR_ERR_LIMIT (-5)   /* larger than a fixed capacity: asr_colreduce_static's built-in buffer; a layer whose
                              * resident weights / persistent grid do not fit the chip (asr_*_scan_*) */

/* K1.  C = alpha*op(A)*op(B) + beta*C (+bias[N] per row); exact-fp32 MFMA.  Replaces every T.dot on
 * the path: input projection libs/lhuclstm_layers.py:189, libs/lasagne_libs/recurrent_layers.py:701,
 * :1233, libs/lasagne_libs/hyper_lhuc_layers.py:469; output layer libs/builder_utils.py:44-45;
 * and the dX / dW contractions theano.grad derives from them (libs/comp_graph_utils.py:74). */
int asr_gemm_f32(int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda,
                 const float* B, int ldb, float beta, float* C, int ldc, const float* bias,
                 asr_stream_t stream);

/* Same with a scratch buffer for deterministic split-K partials (used when the output has few tiles
 * and K is long: the weight-gradient contractions with K = T*B). */
int asr_gemm_f32_ws(int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda,
                    const float* B, int ldb, float beta, float* C, int ldc, const float* bias, float* ws,
                    size_t ws_bytes, asr_stream_t stream);

/* The optional bf16 mode of K1 (BASELINE.json north_star): tensors stay fp32 in memory, both
 * operands are rounded to bf16 (nearest even) on the way into LDS and multiplied with bf16 MFMA, fp32 accumulate.
 * Bound measured against the float64 oracle at the full-width configs: tests/test_gpu_bf16.py. */
int asr_gemm_bf16_ws(int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda,
                     const float* B, int ldb, float beta, float* C, int ldc, const float* bias, float* ws,
                     size_t ws_bytes, asr_stream_t stream);
/* fp32-grade products on the bf16 matrix cores: every operand element is split into three bf16 parts while it
 * is staged into LDS and `terms` partial products are accumulated in fp32 (8: error below fp32's own product
 * rounding; 6: <= 2^-23 |a b|; 3: a reduced-precision mode from TWO parts per operand, error <= 2^-15 |a|.|b| element
 * by element, with a stated end-to-end bound on every config, tests/test_gpu_bf16.py; 1: plain bf16 in the same
 * kernel).  Same T.dot call sites as asr_gemm_f32_ws. */
int asr_gemm_bf16x3_ws(int terms, int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda,
                       const float* B, int ldb, float beta, float* C, int ldc, const float* bias, float* ws,
                       size_t ws_bytes, asr_stream_t stream);

/* The same products with operands split ONCE per tensor instead of once per product: asr_split_bf16x3 writes the
 * three bf16 parts of an fp32 matrix (rows x cols, leading dimension ldx) as planes of the matrix's own row-major
 * shape, leading dimension ldp = asr_split_ld(cols) (cols rounded up to 8, padding zeroed), `plane_stride` elements
 * apart (>= rows * ldp, a multiple of 8; `planes` 16-byte aligned).  asr_gemm_presplit_ws is asr_gemm_bf16x3_ws with
 * terms = 6 whose A and / or B (preA / preB != 0) is such a set of planes: the pointer is plane 0 of the stored matrix
 * (sub-matrices: offset by whole rows, or by a multiple of 8 columns), lda / ldb = ldp.  Bit-identical to the
 * fp32-operand call; a layer input feeds four T.dot's (x.W_x, x.W_xhat and their weight gradients,
 * libs/lasagne_libs/hyper_lhuc_layers.py:469,216-224 and theano.grad of them). */
int asr_split_ld(int cols);
int asr_split_bf16x3(const float* X, int ldx, int rows, int cols, void* planes, int ldp, size_t plane_stride,
                     asr_stream_t stream);
int asr_gemm_presplit_ws(int preA, int preB, int transA, int transB, int M, int N, int K, float alpha, const void* A,
                         int lda, size_t planeA, const void* B, int ldb, size_t planeB, float beta, float* C, int ldc,
                         const float* bias, float* ws, size_t ws_bytes, asr_stream_t stream);

/* dst[b][a][:] = src[a][b][:]  -- input.dimshuffle(1,0,2), libs/lhuclstm_layers.py:181,213 */
int asr_swap01(const float* src, float* dst, int A, int B, int D, int ld_src, int ld_dst,
               asr_stream_t stream);

/* Row layer norm hoisted out of the recurrence: ln()/sample_ln() of the input-side pre-activation,
 * libs/lhuclstm_layers.py:16-21,77; libs/lasagne_libs/recurrent_layers.py:10-15,27-32,711,1241.
 * groups = 4: gate_ln()/sample_gate_ln() (:17-25,:34-42), each gate's columns (col & 3) normalised
 * separately, rstd holds 4 values per row; groups = 1 otherwise. */
int asr_ln_rows_fwd(const float* x, int rows, int n, const float* pre, int pmod, const float* alpha,
                    const float* beta, int amod, const float* bias, float* y, float* xhat, float* rstd,
                    int groups, asr_stream_t stream);
int asr_ln_rows_bwd(const float* dy, int rows, int n, const float* xhat, const float* rstd,
                    const float* alpha, int amod, float* dx, int groups, asr_stream_t stream);

/* Deterministic column reductions = the gradients theano.grad produces for b / alpha / beta vectors. */
size_t asr_colreduce_ws_bytes(int n, int gmod);
int asr_colreduce(const float* x, int ldx, const float* y, int ldy, int rows, int n, int gmod,
                  float* out, int accumulate, float* ws, size_t ws_bytes, asr_stream_t stream);

/* asr_colreduce for callers that cannot allocate or carry a workspace (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  The stage-1
 * partials live in a 4 MiB device array built into the library (one copy per GPU, held until the process
 * ends), so gmod * n <= 32768 or ASR_ERR_LIMIT.  The call allocates nothing; `stream` must belong to the
 * current device (ASR_ERR_ARG otherwise: every GPU has its own copy of the array).  All calls on one GPU share that
 * array: calls that overlap on one GPU race.  The first call on a device resolves the array's address
 * (hipGetSymbolAddress, under a lock): make it outside graph capture, e.g. the warm-up form x = NULL, rows = 0
 * (n = gmod = 1), which resolves the address and launches nothing. */
int asr_colreduce_static(const float* x, int ldx, const float* y, int ldy, int rows, int n, int gmod,
                         float* out, int accumulate, asr_stream_t stream);

/* Test probe: y[i] = the library's own sigmoid (kind 0) / tanh (kind 1) / exp (kind 2) of x[i], i.e. the inline
 * forms every recurrence and the softmax use for T.nnet.sigmoid / T.tanh / T.exp (libs/lhuclstm_layers.py:93-108,
 * libs/comp_graph_utils.py:35), so that their accuracy can be measured against float64 on the host. */
int asr_act_eval(int kind, const float* x, float* y, size_t n, asr_stream_t stream);

/* small MLP pieces: LHUC predictor libs/lhuclstm_layers.py:402-409, conditioning prologue
 * libs/lasagne_libs/recurrent_layers.py:1209-1211.  act: 0 identity, 1 relu, 2 tanh */
int asr_bias_act(float* x, const float* bias, int rows, int n, int act, asr_stream_t stream);
int asr_act_bwd(float* dy, const float* y, int rows, int n, int act, asr_stream_t stream);
int asr_rowscale(const float* x, int ldx, const float* s, int B, int relu, float* y, int ldy, int rows,
                 int n, asr_stream_t stream);                 /* LHUC, libs/lhuclstm_layers.py:461-462 */
/* Inverted dropout, lasagne DropoutLayer(p) as models/gating_hyper_nets.py:276-307 places it (in front of every
 * layer and of the output layer; identity when deterministic): y = x * keep / (1-p).  keep comes from
 * Philox4x32-10(key = seed, counter = (row*ceil(n/4) + column/4, stream_id)), u = (word >> 8) * 2^-24, keep <=>
 * u >= p; the mask is never stored: the backward pass applies the same call to the gradient.  x == y allowed. */
int asr_dropout(const float* x, int ldx, float* y, int ldy, int rows, int n, float p, unsigned long long seed,
                unsigned long long stream_id, asr_stream_t stream);
/* y[r,:] = a[b,:]*x[r,:] + c0[b,:] + mask[b,t]*c1[b,:] over time-major rows r = t*B + b: the gradient of the
 * inter / intra variance costs on the inner hidden sequences, scripts/hypernets/exp_prj_hyper.py:108-124 */
int asr_affine_rows(const float* x, int ldx, const float* a, const float* c0, const float* c1, const float* mask,
                    int mask_stride_b, int mask_stride_t, int B, float* y, int ldy, int rows, int n,
                    asr_stream_t stream);
/* The (B,F) algebra of the same costs: st = [mu ; S0 ; S2] (masked mean, sum and sum of squares over time, each
 * (Bg,F)) and the sequence lengths -> costs[0] = inter, costs[1] = intra (added to what is there if accumulate), and,
 * if coef != NULL, the (a, c0, c1) of asr_affine_rows for the batch rows [b0, b0+Bloc) of this process, each (Bloc,F).
 * tpad (or NULL = T for every row): frames the sums of each row ran over, for rows gathered from shards padded apart. */
int asr_var_costs(const float* st, const float* len, const float* tpad, int Bg, int F, int T, float w_inter, float w_intra, int b0,
                  int Bloc, float* coef, float* costs, int accumulate, asr_stream_t stream);
int asr_masked_mean(const float* x, const float* mask, int B, int T, int F, float* out,
                    asr_stream_t stream);
int asr_masked_mean_tanh_bwd(const float* ds, const float* mask, const float* pre, int B, int T, int F,
                             float* dpre, asr_stream_t stream);

/* Strided pooling over time of the conditioning prologue (x holds the activated values, element
 * (b,t,f) at x[b*sb + t*st + f]; mask (B,T) contiguous).  mode 0: masked mean of tanh, libs/lasagne_libs/
 * recurrent_layers.py:945-946,1210-1211; mode 1: unmasked max of relu, :1469-1470,1734-1735; mode 2:
 * plain masked mean (SummarizingLayer, libs/builder_utils.py:5-22, with an all-ones mask).  The
 * backward includes the activation derivative and Theano's max gradient (every tied frame). */
int asr_seq_pool_fwd(const float* x, size_t sb, size_t st, const float* mask, int B, int T, int F, int mode,
                     float* out, asr_stream_t stream);
int asr_seq_pool_bwd(const float* ds, const float* s, const float* x, size_t sb, size_t st, const float* mask,
                     int B, int T, int F, int mode, float* dx, asr_stream_t stream);

/* K3.  Fused log-softmax / CE / argmax / dlogits: libs/comp_graph_utils.py:29-45,
 * scripts/hypernets/exp_prj_ln_lstm_fix.py:84-95, bin/forward_ln_lstmp.py:55-58.
 * out_mode 0 none, 1 log-posteriors, 2 probabilities, written batch-major (B,T,C). */
int asr_softmax_ce(const float* logits, int C, int B, int T, const int* target, int tgt_sb, int tgt_st,
                   const float* tmask, int msk_sb, int msk_st, float scale, float* dlogits,
                   float* row_ce, float* row_ok, int* pred, int out_mode, float* out,
                   asr_stream_t stream);
int asr_reduce_sum(const float* x, size_t n, float* out, int do_sqrt, asr_stream_t stream);

/* K4.  Adam + gradient norm: libs/lasagne_libs/updates.py:587-617, libs/comp_graph_utils.py:77.
 * ws: 1024 floats. */
int asr_adam_step(float* p, const float* g, float* m, float* v, size_t n, float a_t, float b1, float b2,
                  float eps, float* grads_norm_out, int want_sqrt, float* ws, asr_stream_t stream);
int asr_sqsum(const float* g, size_t n, float* out, int want_sqrt, float* ws, asr_stream_t stream);

/* The other optimisers of libs/lasagne_libs/updates.py as one elementwise pass: kind 0 sgd (:64-92),
 * 1 momentum (:95-149,152-188), 2 nesterov_momentum (:191-295), 3 adagrad (:298-363), 4 rmsprop
 * (:366-433), 5 adadelta (:436-531), 6 adamax (:621-698; lr = a_t = lr/(1-b1^t)), 7 adam (:534-618;
 * lr = a_t).  s1/s2: the optimiser's state tensors (velocity | accu | m, delta_accu | u | v), h1/h2:
 * momentum | rho | beta1, beta2.  gscale: NULL or a device scalar every gradient is multiplied by first
 * (the multiplier of total_norm_constraint, :781-855, written by asr_norm_clip_mult). */
int asr_opt_step(int kind, float* p, const float* g, float* s1, float* s2, size_t n, float lr, float h1,
                 float h2, float eps, const float* gscale, asr_stream_t stream);
/* mult_out[0] = clip(norm[0], 0, max_norm) / (eps + norm[0])   (libs/lasagne_libs/updates.py:850-853) */
int asr_norm_clip_mult(const float* norm, float max_norm, float eps, float* mult_out, asr_stream_t stream);

/* K2 (LSTM family).  Persistent recurrent scan, forward and BPTT: theano.scan(step_masked) and its
 * gradient for libs/lhuclstm_layers.py:73-130,205-211; libs/lasagne_libs/recurrent_layers.py:437-521,
 * :707-789, :1237-1310. */
typedef struct asr_lstm_dir {
    int reverse;              /* go_backwards */
    const float* xin;         /* (T,B,4N) input-side pre-activation incl. bias / input LN */
    const float* wpack;       /* asr_lstm_pack_w output (mode 0 for fwd, 1 for bwd) */
    const float* alpha_hid;   /* (4N) or (B,4N) for ln_kind 2 */
    const float* beta_hid;    /* (4N) or NULL */
    const float* alpha_c;     /* (N) */
    const float* beta_c;      /* (N) */
    const float* c0;          /* (B,N) initial cell or NULL (zeros) */
    const float* r0;          /* (B,N) initial r    or NULL (zeros) */
    const float* a0;          /* (B,4N) or NULL: h_init . W_h, the first step's recurrent pre-activation, taken in place
                               * of r0 . W' there (forward): TBPTTLSTMLayer's hid_init is the PROJECTED state of the
                               * previous window, libs/lhuclstm_layers.py:313-319,230-235 */
    float* r_out;             /* (T,B,ld_r) */
    int ld_r;
    float* gates;             /* (T,B,4N) saved for BPTT (NULL = inference) */
    float* cst;               /* (T,B,N) */
    float* ahat;              /* (T,B,4N)  LN */
    float* rs_a;              /* (T,B)     LN */
    float* chat;              /* (T,B,N)   cell LN */
    float* rs_c;              /* (T,B)     cell LN */
    const float* dr_ext;      /* (T,B,ld_dr) backward: dLoss/dr_t from above */
    int ld_dr;
    float* dG;                /* (T,B,4N) */
    float* dA;                /* (T,B,4N) LN only */
    float* dOut;              /* (T,B,N)  cell LN only */
    /* Options of the recurrent_layers.py classes (libs/lasagne_libs/recurrent_layers.py:52-57); all NULL / 0 by default.
     * Any of them set runs the generic kernels' option form, one 16-row chunk per workgroup, and the backward weights
     * must be packed with mode 1 whatever asr_lstm_bwd_pack_mode returns. */
    const float* peep;        /* (N,4) per unit [w_cell_to_ingate, w_cell_to_forgetgate, 0, w_cell_to_outgate] (:112-123,
                               * :190-192,:212-213); classes without layer norm only */
    int a0_until_valid;       /* a0 replaces r.W' until the row's first unmasked step (learned hid_init of a projected
                               * layer held through leading masked steps, :246-248 with :239) */
    const float* dc_ext;      /* backward in: (T,B,ld_dc) dLoss/d cell_t (only_return_hidden=False, :275-288) or NULL */
    int ld_dc;
    float* dc0;               /* backward out (B,N): gradient wrt the initial cell rows (learn_init, :125-129) */
    float* dr0;               /* backward out (B,N): gradient that reached r_{-1} through masked steps; add the first scan
                               * step's dA . W'^T for the whole gradient wrt r0 */
    float* da0;               /* backward out (B,4N): dA of each row's first unmasked step */
    float* dpeep;             /* backward out (B,N,4): per-row sums over time of [dG_i c_{t-1}, dG_f c_{t-1}, 0, dG_o c_t] */
} asr_lstm_dir;

typedef struct asr_lstm_scan_desc {
    int T, B, N, ndir, ln_kind, cell_ln;   /* ln_kind: 0 none, 1 static, 2 per-sample alpha,
                                            * 3 / 4 = 1 / 2 with per-gate statistics (rs_a is (T,B,4)) */
    float grad_clip;                       /* theano.gradient.grad_clip bound, 0 = off */
    const float* mask; int mask_stride_b, mask_stride_t;
    asr_lstm_dir dir[2];
} asr_lstm_scan_desc;

size_t asr_lstm_pack_floats(int N, int mode);
int asr_lstm_pack_w(const float* W, int N, int mode, float* out, asr_stream_t stream);
/* Packing the backward scan of a layer expects for W (mode of asr_lstm_pack_w): 1, or 2 for the LN variants (ln_kind 1..4)
 * whose backward runs as the 2-D decomposition (N % 128 == 0, N <= 512, N/4 * ndir <= 256).  Pack with the mode
 * this returns; asr_lstm_scan_bwd selects its kernel by the same rule. */
int asr_lstm_bwd_pack_mode(int N, int ndir, int ln_kind);
size_t asr_lstm_scan_ws_bytes(int T, int B, int N, int ndir, int backward);
int asr_lstm_scan_fwd(const asr_lstm_scan_desc* d, void* ws, size_t ws_bytes, asr_stream_t stream);
int asr_lstm_scan_bwd(const asr_lstm_scan_desc* d, void* ws, size_t ws_bytes, asr_stream_t stream);
/* after the stream is synchronised: ASR_ERR_TIMEOUT if any group's bounded wait expired */
int asr_scan_status(const void* ws, int ngroups);
/* How many 16-row batch chunks one workgroup of a persistent scan carries per step against the same resident
 * weights (1, 2 or 4; 0 = default: up to 4, or ASR_SCAN_NC).  theano.scan steps any batch size at once
 * (libs/lasagne_libs/hyper_lhuc_layers.py:493-500); here batches above 16 rows ride the same per-step exchanges.
 * Per-row results do not depend on the setting (tests/test_gpu_multichunk.py); process-wide, not thread-safe. */
int asr_scan_set_chunks_per_wg(int nc);

/* (rows,4N) gate-major [i|f|c|o] <-> gate-interleaved col = unit*4+gate (dir 0: to interleaved,
 * 1: back).  The concatenation of the reference's per-gate matrices, libs/lhuclstm_layers.py:58-60,
 * libs/lasagne_libs/hyper_lhuc_layers.py:447-466, in the engine's column order. */
int asr_perm4(const float* src, float* dst, int rows, int N, int dir, asr_stream_t stream);

/* K2 (HyperLSTM).  Persistent scan forward and BPTT: theano.scan(step_masked) over
 * [orig_input, input, mask] with states (hyper_c, hyper_h, c, h) and its gradient,
 * libs/lasagne_libs/hyper_lhuc_layers.py:216-237 (eq.10), :363-412 (step), :414-425, :493-500.
 * The caller hoists the x_t half of eq.10 (xhproj = x.W_xhat[N:] + bhat) and folds eq.11 into eq.12
 * (M = [W_hhat_* ; b_*].W_*z, shape (H+1, 12N), gate-interleaved; row H = cvec). */
typedef struct asr_hyper_dir {
    int reverse;
    const float* xproj;       /* (T,B,4N) x.W_x */
    const float* xhproj;      /* (T,B,4H) */
    const float* wpack;       /* asr_hyper_pack_w output (mode 0 for fwd, 1 for bwd) */
    const float* cvec;        /* (12N) constant row of the folded matrix: [c_h | c_x | c_b] */
    float* r_out; int ld_r;   /* (T,B,ld_r) h_t */
    float* hh_out;            /* (T,B,H)  hhat_t */
    float* gates;             /* (T,B,4N) saved for BPTT (NULL = inference) */
    float* cst;               /* (T,B,N) */
    float* hgates;            /* (T,B,4H) */
    float* hcst;              /* (T,B,H) */
    float* aD;                /* (T,B,3,4N): a = h_{t-1}.W_h, d_h, d_x */
    const float* dr_ext; int ld_dr;   /* backward: dLoss/dh_t from above */
    float* dXp;               /* (T,B,4N) dgates*d_x  -> dW_x, dx */
    float* dXh;               /* (T,B,4H) hyper gate grads -> dW_xhat, dbhat, dW_hhat, dx */
    float* dA;                /* (T,B,4N) dgates*d_h  -> dW_h */
    float* dD;                /* (T,B,3,4N) [dgates*a, dgates*xproj, dgates] -> dM, dcvec */
    float* dcv;               /* (ceil(B/8),12N) or NULL: column sums of dD (d cvec) per 16-row batch chunk */
    float* dbh;               /* (ceil(B/8),4H) or NULL: column sums of dXh (d bhat) per batch chunk */
    float* carry;             /* (B, 2N+2H) BPTT state handed from one time chunk's launch to the next (backward with
                               * step0 > 0 or step1 < T), else NULL */
} asr_hyper_dir;

typedef struct asr_hyper_scan_desc {
    int T, B, N, H, ndir;
    int step0, step1;         /* run steps [step0, step1) of the T-step recurrence (processing order: the reverse
                               * direction's step s is time T-1-s; the BPTT's step s is the forward's step T-1-s);
                               * step1 == 0 means T.  A layer may be run as consecutive time chunks, one launch each:
                               * the forward continues from the rows it saved for step0-1 (training mode: gates/cst/
                               * hgates/hcst given), the backward from `carry`; dcv / dbh accumulate across chunks.  The
                               * caller can then run the next layer's input projection on finished chunks meanwhile. */
    float grad_clip;
    int clip_outcell;         /* 1: also grad_clip the outcell of both cells (ScalingHyperLSTMLayer,
                               * libs/lasagne_libs/hyper_layers.py:1016-1022,1083-1089) */
    int lhuc_reparam;         /* 0, or HyperLHUCLSTMLayer (hyper_lhuc_layers.py:509-678): 1 relu, 2 exp, 3 identity,
                               * 4 2sigmoid (:17-27).  M then carries d_h = d_x = 1 and the LHUC pre-scale in gate
                               * column 0 of its bias tile; gates = h.W_h + xproj, h = o*tanh(c)*reparam(d) */
    const float* mask; int mask_stride_b, mask_stride_t;
    asr_hyper_dir dir[2];
} asr_hyper_scan_desc;

size_t asr_hyper_pack_floats(int N, int H, int mode);
int asr_hyper_pack_w(const float* Wh, const float* Whx, const float* Whh, const float* M, int N, int H,
                     int mode, float* out, asr_stream_t stream);
size_t asr_hyper_scan_ws_bytes(int T, int B, int N, int H, int ndir, int backward);
int asr_hyper_scan_fwd(const asr_hyper_scan_desc* d, void* ws, size_t ws_bytes, asr_stream_t stream);
int asr_hyper_scan_bwd(const asr_hyper_scan_desc* d, void* ws, size_t ws_bytes, asr_stream_t stream);

/* ProjectionHyperLSTMLayer (libs/lasagne_libs/hyper_layers.py:14-523; step :319-417, step_masked
 * :419-456): two N-unit cell-LN LSTMP cells coupled through three scale maps of the new inner hid.  A step and pass
 * is 2 skinny products + 2 cells (the products leave per-split partial blocks in the workspace, the cell behind them
 * sums them while loading).  Two forms, chosen inside the library:
 *   - ONE persistent launch per pass (round 3): one workgroup of 512 threads per CU, the four phases of a step separated
 *     by in-kernel exchanges, every wave's slice of the folded weights kept on the chip for the whole sequence (Wf in
 *     registers, W3 in LDS).  Taken when the slices fit: N <= 512, one round of weight loads per wave at a split into
 *     2 x CUs blocks, B ndir <= CUs (the reference script's 512 units / 16 rows / both directions do).  Its bounded
 *     waits report through asr_scan_status(ws, 1) like the scans'.
 *   - the step loop inside the library (4 launches per step and pass) otherwise, or when ASR_PRJ_PERSIST=0 /
 *     asr_prjhyper_set_mode(1).
 * The two differ in how the reduction axes are split (results agree to rounding, not bit for bit).
 * Gate vectors are gate-major [i|f|c|o] here.  One call handles batch rows [b0, b0+B), B <= 64 (every launch carries
 * all 16-row chunks of both directions). */
typedef struct asr_prjhyper_dir {
    int reverse;
    const float* inner_in;    /* (T,Btot,4N) x.W_in_to_inner + b_inner (:316) */
    const float* outer_in;    /* (T,Btot,4N) x.W_in_to_outer (:317) */
    const float* Wf;          /* (2N,8N) both projections folded into the recurrent products:
                                 [[W_inner_prj.W_inner_to_inner, 0], [W_outer_prj.W_outer_to_inner, W_outer_prj.W_outer_to_outer]]
                                 so that [recurrent inner gates | a]_t = [r_inner | r_outer]_{t-1} . Wf (:329-331,:376) */
    const float* W3;          /* (N,12N) W_inner_prj . [W_inner_to_in | W_inner_to_outer | W_inner_to_bias] (:364-370) */
    const float* s;           /* (12N) */
    const float *ai, *bi_c, *ao, *bo_c;   /* cell layer norm alpha / beta (N) */
    float* r;                 /* (T,Btot,2N) [r_inner | r_outer], r = o*tanh(outcell), masked; hid = r . W_prj after the loop */
    float *gi, *go;           /* (T,Btot,4N) post-activation gates, saved for BPTT */
    float *ci, *co;           /* (T,Btot,N) masked cell state */
    float *xhi, *xho;         /* (T,Btot,N) normalised cell */
    float *rsi, *rso;         /* (T,Btot) rstd */
    float* sc;                /* (T,Btot,12N) scale_outer */
    float* a;                 /* (T,Btot,4N) a = outer_hid_prev . W_outer_to_outer */
    float* dr;                /* backward: (T,Btot,2N) dhid . W_prj^T (the part of d r_t from outside the loop) on entry;
                                 work: switched-off rows add what they pass on to the row of the step before */
    float* dta;               /* (T,Btot,8N) [d inner gate pre-activations | d a] */
    float* doi;               /* (T,Btot,4N) d outer_input */
    float* dsc;               /* (T,Btot,12N) d scale_outer */
    float *douti, *douto;     /* (T,Btot,N) clipped d outcell */
} asr_prjhyper_dir;
typedef struct asr_prjhyper_desc {
    int T, B, N, F, ndir;
    int b0, Btot;
    float grad_clip;
    const float* mask; int mask_stride_b, mask_stride_t;
    asr_prjhyper_dir dir[2];
} asr_prjhyper_desc;
size_t asr_prjhyper_ws_bytes(int B, int N, int F, int ndir);
int asr_prjhyper_fwd(const asr_prjhyper_desc* d, void* ws, size_t ws_bytes, asr_stream_t stream);
int asr_prjhyper_bwd(const asr_prjhyper_desc* d, void* ws, size_t ws_bytes, asr_stream_t stream);
/* 0: the library chooses (default), 1: always the step loop.  Returns the previous mode.  For tests and measurements
 * (both forms must agree); asr_prjhyper_ws_bytes follows the mode, so query it after the call.  No counterpart in the
 * reference. */
int asr_prjhyper_set_mode(int mode);

/* ScalingHyperLSTMLayer with use_peepholes / use_layer_norm / learn_init (libs/lasagne_libs/hyper_layers.py:525-1205;
 * step :963-1095, step_masked :1097-1118, initial rows :1127-1131, scan :1182-1194).  Without options the layer runs on
 * asr_hyper_scan_* (clip_outcell = 1); per-gate layer norms need five all-unit reductions per cell and step, so this
 * path is a step loop inside the library like asr_prjhyper_*: one workgroup per (batch row, direction) runs a cell.
 * Gate vectors gate-major [i|f|c|o]; hid = [inner_hid | outer_hid].  One call handles batch rows [b0, b0+B), B <= 64. */
typedef struct asr_scaling_dir {
    int reverse;
    const float* inner_in;    /* (T,Btot,4H) x.W_inner_in_to_inner + b_inner (:951) */
    const float* outer_in;    /* (T,Btot,4N) x.W_outer_in_to_outer (:952) */
    const float* Wio;         /* (H+N,4H) [W_inner_hid_to_inner ; W_outer_hid_to_inner] (:974-975) */
    const float* S;           /* (H,NS*4N) [to_outer_in | to_outer_hid | to_outer_bias | to_outer_cell], NS = 3 or 4 (:1030-1033) */
    const float* s;           /* (NS*4N) */
    const float* Woo;         /* (N,4N) W_outer_hid_to_outer (:1039) */
    const float* wpeep;       /* (4H) W_inner_cell_to_inner, cell block zero; NULL without peepholes (:977-979,:1006) */
    const float *lna_i, *lnb_i; /* (5,H) alpha / beta of the layer norms of ingate, forgetgate, cell, outgate, outcell; NULL without LN */
    const float *lna_o, *lnb_o; /* (5,N) */
    const float* init;        /* (2H+2N) inner cell | inner hid | outer cell | outer hid initial rows; NULL = zeros */
    float* hid;               /* (T,Btot,H+N) */
    float *gi, *go;           /* (T,Btot,4H) / (T,Btot,4N) post-activation gates, saved for BPTT */
    float *ci, *co;           /* (T,Btot,H) / (T,Btot,N) masked cell state */
    float *xh_i, *xh_o;       /* (T,Btot,5H) / (T,Btot,5N) normalised values of the five layer norms (LN only) */
    float *rs_i, *rs_o;       /* (T,Btot,5) rstd (LN only) */
    float* sc;                /* (T,Btot,NS*4N) scale maps */
    float* a;                 /* (T,Btot,4N) outer_hid_prev . W_outer_hid_to_outer */
    const float* dhid;        /* backward: (T,Btot,H+N) dLoss/d hid */
    float *dgi, *dgo;         /* (T,Btot,4H) d inner gate pre-activations ; (T,Btot,4N) d outer_input */
    float* dsc;               /* (T,Btot,NS*4N) */
    float* da;                /* (T,Btot,4N) */
    float* dh;                /* (T,Btot,H+N) total d hid_t (work) */
    float *dy_i, *dy_o;       /* (T,Btot,5H) / (T,Btot,5N) clipped gradients at the layer-norm outputs (LN only) */
    float* dpeep;             /* (T,Btot,3H) d gate * cell of the inner peepholes (i, f, o); NULL without peepholes */
    float* dinit;             /* (Btot,2H+2N) gradient wrt the initial rows, per batch row; may be NULL */
    const float* tg0;         /* (4H) or NULL: added to a row's inner gates while its previous state is still the initial one.
                               * EqualScaleHyperLSTMLayer(learn_init=True): inner_hid_init . W_inner_hid_to_inner, the learned row
                               * being in the projected space (libs/lasagne_libs/hyper_layers.py:1782-1788,:1657) */
    float* dgi0;              /* (Btot,4H) or NULL, backward out: dgi of each row's first unmasked step (clear beforehand) */
} asr_scaling_dir;
typedef struct asr_scaling_desc {
    int T, B, N, H, ndir;
    int b0, Btot;
    int use_peepholes, use_layer_norm;
    float grad_clip;
    const float* mask; int mask_stride_b, mask_stride_t;
    asr_scaling_dir dir[2];
} asr_scaling_desc;
size_t asr_scaling_ws_bytes(int B, int N, int H, int ndir);
int asr_scaling_fwd(const asr_scaling_desc* d, void* ws, size_t ws_bytes, asr_stream_t stream);
int asr_scaling_bwd(const asr_scaling_desc* d, void* ws, size_t ws_bytes, asr_stream_t stream);
/* Like asr_prjhyper_*, the library runs this recurrence as ONE persistent launch per pass where the shape allows (both
 * unit counts <= 512, B ndir <= CUs, the blocks of a product phase within 2 x CUs: the TIMIT script's 64 / 128 units do)
 * and as the step loop otherwise; ASR_SCAL_PERSIST=0 or asr_scaling_set_mode(1) force the step loop (0: the library
 * chooses; returns the previous mode; asr_scaling_ws_bytes does not depend on it).  A bounded wait of the persistent launch
 * that gave up is reported by asr_scan_status(ws, 1).  No counterpart in the reference. */
int asr_scaling_set_mode(int mode);

#ifdef __cplusplus
}
#endif
#endif
