/*
 * atr_policy.h — C ABI of the fused policy-side kernels in libtrack2d_hip.so (the batched
 * policy step is the next row after the env path). Plain pointers and sizes; PyTorch only owns the buffers.
 *
 * atr_stem_*: the conv stem of CNN_maze (perception.py:68-92 of the reference): conv(1->16,k3,s2,p1) + ReLU +
 * conv(16->32,k3,s2,p1) + ReLU on 13x13 frames, fp32.
 *   x   [M,169]  frames (row-major 13x13, one channel); row m starts at x + m * x_stride floats (x_stride >= 169), so a
 *                strided view of the env's obs tensor [N,2,13,13] (one agent's frames: stride 338) needs no copy
 *   w1  [16,9] (conv1.weight [16,1,3,3]), b1 [16], w2 [32,144] (conv2.weight [32,16,3,3]), b2 [32]
 *   y   [M,512]  conv2 output after ReLU in (c,h,w) order — the layout x.view(N,-1) feeds the fc layer
 * All pointers are device pointers; `stream` is a hipStream_t. Return 0, -1 (bad argument) or -2 (launch failure).
 */
#ifndef ATR_POLICY_H
#define ATR_POLICY_H

#ifdef __cplusplus
extern "C" {
#endif

int atr_stem_forward(const float *x, long long x_stride, const float *w1, const float *b1, const float *w2,
                     const float *b2, float *y, long long M, void *stream);
/* Two independent stems (different weights and inputs — the rollout's tracker and target encoders) in ONE launch:
 * the workgroups are split between the problems in proportion to M0 : M1. Same result as two atr_stem_forward calls. */
int atr_stem_forward2(const float *x0, long long x0_stride, const float *w1_0, const float *b1_0, const float *w2_0,
                      const float *b2_0, float *y0, long long M0, const float *x1, long long x1_stride,
                      const float *w1_1, const float *b1_1, const float *w2_1, const float *b2_1, float *y1,
                      long long M1, void *stream);
/* Floats of scratch atr_stem_backward needs for M frames (per-wave partial gradient records). */
long long atr_stem_workspace_floats(long long M);
/* Gradients of the four parameter tensors given dy = dL/dy (the observation needs no gradient). Outputs are
 * overwritten (not accumulated); the reduction order is fixed, so results are run-to-run deterministic. */
int atr_stem_backward(const float *x, long long x_stride, const float *y, const float *dy, const float *w1,
                      const float *b1, const float *w2, float *dw1, float *db1, float *dw2, float *db2,
                      float *workspace, long long M, void *stream);

/* The same three entry points on u8 frames (the env's t2d_step_u8 observations, one byte per cell, values 0/1/2/4;
 * x_stride in bytes): the float32 cast frame_stack applies (environment.py:138,146 of the reference) happens inside
 * conv1's load, so the observation crosses HBM once, as bytes. Bit-identical results to the float entry points on
 * float(x). */
int atr_stem_forward_u8(const unsigned char *x, long long x_stride, const float *w1, const float *b1, const float *w2,
                        const float *b2, float *y, long long M, void *stream);
int atr_stem_forward2_u8(const unsigned char *x0, long long x0_stride, const float *w1_0, const float *b1_0,
                         const float *w2_0, const float *b2_0, float *y0, long long M0, const unsigned char *x1,
                         long long x1_stride, const float *w1_1, const float *b1_1, const float *w2_1,
                         const float *b2_1, float *y1, long long M1, void *stream);
int atr_stem_backward_u8(const unsigned char *x, long long x_stride, const float *y, const float *dy, const float *w1,
                         const float *b1, const float *w2, float *dw1, float *db1, float *dw2, float *db2,
                         float *workspace, long long M, void *stream);

/* Actor head of the rollout in one launch (replaces actor_linear -> softmax -> multinomial, model.py:41-49 of the
 * reference): logits = w h + b with h [n,R] (R <= 256, multiple of 4), w [A,R], b [A], A <= 8; one categorical draw
 * per row by inverse CDF on a Philox4x32-10 uniform keyed (seed; row, *counter, ordinal). `counter` is a device-side
 * uint64 that the call advances by one in stream order when bump != 0, so hipGraph replays draw fresh numbers;
 * `ordinal` distinguishes the calls made between two bumps (e.g. bump once per rollout, ordinal = call index inside
 * it: one tiny launch per rollout instead of one per call). n == 0 with bump != 0 only advances the counter.
 * actions: int64 [n]. */
int atr_sample_actions(const float *h, const float *w, const float *b, long long *actions, unsigned long long *counter,
                       unsigned long long seed, unsigned ordinal, int bump, int n, int R, int A, void *stream);

/* One LSTMCell step (torch.nn.LSTMCell semantics, gate order i,f,g,o; model.py:110,172 of the reference) for P <= 2
 * players x N envs x R hidden units (R multiple of 4), given the two GEMM results:
 *   ig0/ig1 [N,4R] per player   x W_ih^T + b_ih + b_hh
 *   hg      [P,N,4R]            h_prev W_hh^T, h_prev NOT yet masked
 * with the episode-boundary mask of the previous step folded in (k[n] = keep[n], or done[n] == 0, or 1 if both are
 * NULL): gates = ig + k hg, c' = f (k c_prev) + i g, h' = o tanh(c') — identical to masking h_prev and c_prev first.
 * Per-player tensors are addressed as base + p * pstride + n * R (floats). acts (nullable) receives the activated
 * gates [N,4R] per player for atr_lstm_cell_backward. */
int atr_lstm_cell_forward(const float *ig0, const float *ig1, const float *hg, const float *c_prev,
                          long long c_prev_pstride, const float *keep, const unsigned char *done, float *h_out,
                          long long h_pstride, float *c_out, long long c_pstride, float *acts, long long acts_pstride,
                          int P, int N, int R, void *stream);
/* The actor's step for ONE player in one launch: the cell above (P = 1, done-flag mask) and, on the fresh hidden row,
 * the actor head + categorical draw of atr_sample_actions (same key: seed; row, *counter, ordinal — the counter is NOT
 * advanced here). Optionally adds emb[act_in[n]] ([*,4R] rows: the tracker-action embedding already projected through
 * W_ih, model.py:178 of the reference) to the pre-activations. R/4 must be 16, 32 or 64 (a row's lanes share a wave). */
int atr_lstm_cell_forward_act(const float *ig, const float *hg, const float *c_prev, const unsigned char *done,
                              float *h_out, float *c_out, float *acts, const float *emb, const long long *act_in,
                              const float *actor_w, const float *actor_b, int A, long long *actions_out,
                              const unsigned long long *counter, unsigned long long seed, unsigned ordinal, int N,
                              int R, void *stream);
/* atr_lstm_cell_forward_act with the bias (b_ih + b_hh, [4R], nullable) added inside: ig then comes from a bias-free
 * (batched) GEMM. */
int atr_lstm_cell_forward_act1(const float *ig, const float *hg, const float *bias, const float *c_prev,
                               const unsigned char *done, float *h_out, float *c_out, float *acts, const float *emb,
                               const long long *act_in, const float *actor_w, const float *actor_b, int A,
                               long long *actions_out, const unsigned long long *counter, unsigned long long seed,
                               unsigned ordinal, int N, int R, void *stream);
/* The same for BOTH players of a model whose players do not depend on each other's action (maze-lstm pairs), one launch:
 * ig [2,N,4R] = the two input projections WITHOUT bias (one batched GEMM), bias0 / bias1 [4R] (b_ih + b_hh, nullable)
 * added here, hg [2,N,4R]; per-player strides for the state / gate stores as in atr_lstm_cell_forward; actions_out [2,N];
 * player p draws under ordinal + p (the numbers two one-player calls with consecutive ordinals would draw). */
int atr_lstm_cell_forward_act2(const float *ig, const float *hg, const float *bias0, const float *bias1, const float *c_prev,
                               long long c_prev_pstride, const unsigned char *done, float *h_out, long long h_pstride,
                               float *c_out, long long c_pstride, float *acts, long long acts_pstride,
                               const float *actor_w0, const float *actor_b0, const float *actor_w1, const float *actor_b1,
                               int A, long long *actions_out, const unsigned long long *counter, unsigned long long seed,
                               unsigned ordinal, int N, int R, void *stream);
/* The END of a rollout step as ONE launch (csrc/track2d_hip.hip: k_act_step): both players' LSTM cells + actor heads +
 * categorical draws — tracker first, then the target, whose pre-activations receive emb[a_tracker] when emb != NULL (the
 * tracker-aware target, model.py:190-209,249-257 of the reference) — and, with the two fresh actions still in registers, the
 * env step + observation of t2d_step / t2d_step_u8 (Track1v1Env.step, track_1v1.py:71-127; in-launch auto-reset). It is
 * what train.py:81-88 -> player_util.py:44-67 does after the two GEMMs of each LSTMCell: replaces two
 * atr_lstm_cell_forward_act1 launches, the action round trip and the step launch, with bit-identical results.
 * Per player p: ig[p] [N,4R] input projection; hg[p] [N,4R] = h_prev W_hh^T (un-masked; NULL = ig[p] already holds the
 * whole pre-activation); bias[p] [4R] (nullable, added here); c_prev[p] / h_out[p] / c_out[p] [N,R]; acts[p] [N,4R]
 * (nullable: activated gates for atr_lstm_cell_backward); actor_w[p] [A,R], actor_b[p] [A]. done_prev [N] (nullable): the
 * previous step's done flags (k = done == 0 masks hg and c_prev). actions_out int64 [2,N]. Draw key: (seed; row, *counter,
 * ordinal + p) — the key of atr_lstm_cell_forward_act*. R must be 128. */
typedef struct atr_act_step {
    const float *ig[2];
    const float *hg[2];
    const float *bias[2];
    const float *c_prev[2];
    float *h_out[2];
    float *c_out[2];
    float *acts[2];
    const float *actor_w[2];
    const float *actor_b[2];
    const float *emb;
    const unsigned char *done_prev;
    long long *actions_out;
    const unsigned long long *counter;
    unsigned long long seed;
    unsigned ordinal;
    int A, N, R;
    /* with an env handle only (nullable): hm_out[p] + e * hm_ld receives (done[e] == 0 ? h_out row : 0) — the hidden row as
     * the NEXT step's LSTMCell GEMM wants it (episode-boundary mask applied, player_util.py:98-102), written next to that
     * step's features so that the cell's two GEMMs are one product over rows [features | k h] (atr_linear, K = 256 + 128) */
    float *hm_out[2];
    long long hm_ld;
} atr_act_step;
/* OPT-IN (ATR_GATE_CELL=1; round 6) — parity-green, measured 3.5 us per launch slower than the library product at 4096 rows while
 * saving 2.2 us in atr_act_env_step: DESIGN.md section 5 has the per-workgroup timeline.
 * The step's LSTMCell product with the cell as its epilogue (csrc/gate_cell_hip.hip; f32 MFMA, gfx950): for both players
 *     pre[p] = a[p] w[p]^T            a[p] [N, K] rows [fc features | k h_prev] (row stride lda), w[p] [4R, K] = [W_ih | W_hh]
 * (model.py:116,137,165,196 of the reference: both GEMMs of nn.LSTMCell as one product, K = 256 + 128), written to pre[p] [N, 4R]
 * WITHOUT the bias (nullable where cell[p] != 0), and for every player with cell[p] != 0 the cell itself in the kernel's epilogue
 * — ((pre + bias[p])) -> i, f, g, o (csrc/atr_cell.h: the expressions atr_act_env_step evaluates), c' = f (k c_prev) + i g,
 * h' = o tanh(c'), k = (done_prev == 0) — into h_out[p] / c_out[p] [N, R]: that player's gate tensor is then never read back by
 * the rollout (atr_act_env_step is told so by ig[p] == NULL: it takes h_out[p] as the fresh hidden row). A player whose gates
 * still lack a term when the product is done — the tracker-aware target: + fc_action_tracker(one_hot(a_tracker)), model.py:193-194,
 * known only after the tracker's draw — keeps cell[p] = 0. R must be 128, K a multiple of 32, pointers 16-byte aligned.
 * probe: NULL, or u64 [atr_gate_cell_workgroups(N)][4] clock stamps (start, main loop end, end, XCD). Returns 0, -1, -2. */
typedef struct atr_gate_cell_args {
    const float *a[2];
    const float *w[2];
    const float *bias[2];
    float *pre[2];
    const float *c_prev[2];
    float *h_out[2];
    float *c_out[2];
    const unsigned char *done_prev;
    long long lda;
    int cell[2];
    int N, K, R;
    void *probe;
} atr_gate_cell_args;
int atr_gate_cell(const atr_gate_cell_args *args, void *stream);
int atr_gate_cell_workgroups(int N);

struct t2d_handle;
/* env == NULL: the policy half alone (N rows; the learner's bootstrap step). Otherwise N must equal the handle's env count,
 * obs is u8 [N,2,13,13] (obs_is_u8 != 0; 4-byte aligned) or float32 [N,2,13,13], rew float32 [N,2], done u8 [N]; only for
 * handles t2d_step_u8 accepts ('Partial' observations, no Nav/RPF target). Returns 0 or a T2D_ERR_* code (t2d_last_error). */
int atr_act_env_step(struct t2d_handle *env, const atr_act_step *args, void *obs, int obs_is_u8, float *rew,
                     unsigned char *done, void *stream);

/* EXPERIMENTAL — off by default (ATR_COOP_STEP=1), measured SLOWER than the four-launch step it replaces (1.51 against 1.42 ms per
 * synchronous iteration at 512 envs: profiles/EXPERIMENTS.md "The two-launch step"); kept built and parity-tested.
 * The small-shard rollout step after the stem as ONE launch (csrc/track2d_hip.hip: k_coop_step; csrc/coop_gemm.h) — what
 * train.py:81-88 -> player_util.py:44-67 -> model.py:238-265 of the reference do between the conv stem and the next
 * observation: CNN_maze's fc + ReLU for both players (perception.py:81,90), both GEMMs of nn.LSTMCell for both players
 * (model.py:110,137,172,203) as one product over [features | k h_prev] rows, then atr_act_env_step's cells + heads + draws +
 * env step (Track1v1Env.step, track_1v1.py:71-127). Every XCD of the chip owns one eighth of the envs end to end; its
 * workgroups exchange the layers' activations through the L2 they share and meet at two barriers that live in that L2
 * (no chip-wide fence): 2 launches per env step (stem, this) instead of 4.
 *   y[p] [N, kfc[p]] (row stride ldy[p]): player p's stem output; fc_w[p] [F, kfc[p]], fc_b[p] [F];
 *   fh: this step's rows — player p at fh + p * fh_pstride, row stride fh_ld = F + R: the fc features are written into
 *       columns [0, F), columns [F, F + R) already hold k h_prev (written by the previous step through act->hm_out);
 *   w_cat[p] [4R, F + R] = [W_ih | W_hh]; gates [2, N, 4R] scratch; workgroups: the grid size = the number of CUs the
 *   stream may use (a multiple of 8; ALL of them must be able to be resident at once — the barriers spin).
 * `act`: as for atr_act_env_step with bias[p] = b_ih + b_hh and hm_out set; act->ig / hg are ignored. N must be a multiple
 * of 128 and small enough for the grid (N <= 2048 with 256 workgroups). Returns 0 or a T2D_ERR_* code (t2d_last_error);
 * a placement or barrier failure on the device raises bits 1-3 of the handle's fault word (t2d_get_faults). */
typedef struct atr_coop_step {
    const float *y[2];
    const float *fc_w[2];
    const float *fc_b[2];
    const float *w_cat[2];
    long long ldy[2];
    int kfc[2];
    float *fh;
    float *gates;
    long long fh_pstride, fh_ld;
    int F;
    int workgroups;
    void *probe;              /* NULL, or u64 [workgroups][8]: clock stamps at the kernel's phase boundaries (tools/coop_step_timeline.py) */
} atr_coop_step;
int atr_coop_env_step(struct t2d_handle *env, const atr_act_step *act, const atr_coop_step *coop, void *obs, int obs_is_u8,
                      float *rew, unsigned char *done, void *stream);

/* The rollout step's two small GEMM pairs as ONE launch each, for small row counts (csrc/pair_gemm_hip.hip; f32 MFMA):
 *     C[p] = act( A1[p] W1[p]^T [+ (k A2[p]) W2[p]^T] + bias[p] ),  p = 0, 1 (the two players), k[m] = (done[m] == 0)
 * A1[p] [M, k1[p]] (row stride lda1[p] floats), W1[p] [N, k1[p]] (nn.Linear layout), optional second term A2[p] [M, k2[p]],
 * W2[p] [N, k2[p]] (both NULL to leave it out), bias[p] [N] (nullable), C[p] [M, N] (row stride ldc[p]); relu != 0 applies
 * max(., 0). Uses: CNN_maze's fc + ReLU for both players (perception.py:81,90 of the reference; k1 = 512 / 1024), and both
 * GEMMs of nn.LSTMCell for both players in one pass (model.py:110,137,172,203: A1 = features, W1 = weight_ih, A2 = previous
 * hidden state masked by the previous step's done flags, W2 = weight_hh, bias = b_ih + b_hh). N and every k multiples of
 * 32, (k1 + k2) >= 128, 16-byte aligned pointers, strides multiples of 4. Returns 0, -1 (bad argument), -2 (launch failure). */
/* A plain Linear layer (batch of them) through hipBLASLt called directly (csrc/lt_gemm.cpp):
 *     C[b] = act(A[b] W[b]^T + bias),  b < batch;  A [M, K] (row stride lda), W [N, K] (nn.Linear layout, row stride ldw),
 *     C [M, N] (row stride ldc >= N: the output may be a column block of wider rows), bias [N] (nullable, batch == 1 only),
 *     relu != 0 applies max(., 0); stride_* = floats between consecutive batch members.
 * Uses: CNN_maze's fc + ReLU (perception.py:81,90 of the reference) written into the first 256 columns of the rollout's
 * [features | k h_prev] rows, and both GEMMs of nn.LSTMCell for both players (model.py:110,137,172,203) as ONE batched
 * product over those rows (K = 384) -> one gate tensor. Each distinct problem is timed once over the library's candidate
 * kernels (first call outside a stream capture) and keeps the fastest for the life of the process.
 * atr_lt_init: once per process, with the path of the libhipblaslt.so PyTorch itself loads (this library does not link one).
 * Return 0 or -1 (atr_lt_last_error()). */
typedef struct atr_linear_args {
    const float *a, *w, *bias;
    float *c;
    long long lda, ldw, ldc;
    long long stride_a, stride_w, stride_c;
    int M, N, K, batch, relu;
    void *workspace;              /* device scratch for split-K style kernels (16-byte aligned; NULL = the library's own, which */
    long long workspace_bytes;    /* callers on DIFFERENT streams must not share: give each concurrent chain its own) */
} atr_linear_args;
int atr_lt_init(const char *libhipblaslt_path);
int atr_linear(const atr_linear_args *args, void *stream);
/* where a problem's kernel choice came from (atr_linear_plan_info's `source`) */
#define ATR_LT_SOURCE_NONE 0
#define ATR_LT_SOURCE_TIMED 1          /* the candidate list was timed at first use */
#define ATR_LT_SOURCE_RECORDED 2       /* atr_linear_set_choice's record, verified against the list */
#define ATR_LT_SOURCE_FIRST_USABLE 3   /* first call came inside a stream capture: first usable candidate, timed later */
#define ATR_LT_SOURCE_REFUSED 4        /* (transient) a record that did not match the loaded library's list */
#define ATR_LT_SOURCE_REFUSED_TIMED 5  /* ... and the list was timed instead */
/* solution: the library's solution index of the kernel in use (hipblaslt_ext::getIndexFromAlgo; -1 if it does not say). */
int atr_linear_plan_info(const atr_linear_args *args, int *candidates, int *chosen, int *tuned, float *best_us, int *solution,
                         int *source);
/* Select candidate `index` of the library's list for this problem (a choice an earlier run timed and recorded) instead of
 * timing the list at first use; called again later it replaces the earlier choice (the next atr_linear call resolves it).
 * `solution` >= 0 is the recorded kernel's solution index: a position means nothing on another build of the library, so the
 * candidate at `index` is only taken when it IS that solution, else the list is searched for it; an out-of-range index, an
 * unusable candidate or an absent solution REFUSES the record and the list is timed instead (source = REFUSED_TIMED). */
int atr_linear_set_choice(const atr_linear_args *args, int index, int solution);
/* The loaded library's hipblasLtGetVersion and revision, and the version of the header csrc/lt_gemm.cpp was compiled against
 * (they may differ: PyTorch ships its own copy; atr_lt_init's functional self-test — one small RELU_BIAS product against host
 * loops — is what decides whether the direct path is used). 0, or -1 before atr_lt_init. */
int atr_lt_library_info(int *version, int *header_version, char *gitrev, int len);
int atr_linear_kernel_name(const atr_linear_args *args, char *out, int len);
const char *atr_lt_last_error(void);

typedef struct atr_pair_linear_args {
    const float *a1[2], *w1[2], *a2[2], *w2[2], *bias[2];
    float *c[2];
    long long lda1[2], lda2[2], ldc[2];
    int k1[2], k2[2];
    const unsigned char *done;
    int M, N, relu;
} atr_pair_linear_args;
int atr_pair_linear(const atr_pair_linear_args *args, void *stream);

/* EXPERIMENTAL — off by default (ATR_MFMA_MIN_ROWS / bench.py --actor-step mfma select it): round 2's per-player fused GEMM + cell,
 * 22 us per 4096-row call against 14.9 + 6 for the library product + k_act_step's share since round 3; superseded as an
 * experiment by atr_gate_cell (both players, one launch, DMA-to-LDS operands). Kept built and tested.
 * The actor's whole LSTMCell step for ONE player as one f32-MFMA kernel (csrc/actor_step_hip.hip): both GEMMs of
 * nn.LSTMCell (model.py:110,172 of the reference) and the cell, without materialising the gate pre-activations:
 *   gates = f W_ih^T + (k h_prev) W_hh^T + bias [+ emb[act_in[n]]],  k[n] = (done[n] == 0) (1 if done is NULL)
 *   c' = sigm(f) (k c_prev) + sigm(i) tanh(g),  h' = sigm(o) tanh(c')        (gate order i, f, g, o)
 * f [N,F], h_prev / c_prev / h_out / c_out [N,R] (out must not alias prev), w_ih [4R,F], w_hh [4R,R], bias [4R]
 * (= b_ih + b_hh), emb (nullable) [*,4R] rows selected by act_in [N] (the tracker-action embedding projected through
 * W_ih, model.py:178), acts (nullable) [N,4R] receives the activated gates for atr_lstm_cell_backward.
 * Only F = 256, R = 128 (the maze policies); -1 otherwise. Follow with atr_sample_actions on h_out for the draw. */
int atr_actor_step(const float *f, const float *h_prev, const float *c_prev, const unsigned char *done,
                   const float *w_ih, const float *w_hh, const float *bias, const float *emb, const long long *act_in,
                   float *h_out, float *c_out, float *acts, int N, int F, int R, void *stream);
/* One step of back-propagation through time for the cell above. dh_out: dL/dh' from the heads; dh_next [P,N,R]:
 * dg_{t+1} W_hh (gradient arriving through the next step's hidden GEMM, unmasked); dc_carry [P,N,R]: in = dc f of step
 * t+1, out = dc f of this step; both are scaled by keep_out (this step's mask k_t) when has_next != 0 and ignored
 * otherwise. keep_in = k_{t-1}. dg: pre-activation gate gradients [N,4R] per player = dL/d ig. */
int atr_lstm_cell_backward(const float *dh_out, long long dh_pstride, const float *dh_next, float *dc_carry,
                           const float *keep_out, const float *keep_in, const float *acts, long long acts_pstride,
                           const float *c, long long c_pstride, const float *c_prev, long long c_prev_pstride, float *dg,
                           long long dg_pstride, int has_next, int P, int N, int R, void *stream);
/* Back-propagation through time of P <= 2 LSTMCells over a whole T-step rollout as ONE launch (csrc/bptt_hip.hip) — T calls
 * of atr_lstm_cell_backward interleaved with T batched GEMMs dG_t W_hh (the gradient arriving through the hidden state).
 * dh{0,1}_heads [T,N,R] per player: dL/dh_t from the heads (NULL = zero); keep [T,N] float episode masks (keep[t] cuts what
 * arrives from step t + 1, keep[t-1] masks step t's previous state); acts + p*acts_pstride [T,N,4R] activated gates; c_all +
 * p*c_pstride [T+1,N,R] cell states (slot t before step t); whh{0,1} = weight_hh [4R,R]. Out: dg + p*dg_pstride [T,N,4R]
 * = dL/d(pre-activations), dh_init / dc_init [P,N,R] = gradient into the rollout's initial state. R must be 128. */
int atr_lstm_bptt(const float *dh0_heads, const float *dh1_heads, const float *keep, const float *acts, long long acts_pstride,
                  const float *c_all, long long c_pstride, const float *whh0, const float *whh1, float *dg,
                  long long dg_pstride, float *dh_init, float *dc_init, int P, int T, int N, int R, void *stream);
/* atr_lstm_bptt for a rollout that kept the OUTPUT of the step's gate GEMM — the pre-activations without bias, [P][T][N][4R] at
 * `pre` + p * pre_pstride — instead of the activated gates (atr_act_env_step / atr_coop_env_step with acts == NULL: the step's
 * last kernel then writes 4R floats per row less). The activations are recomputed inside with the expressions the step used:
 * ((pre + bias_p) + emb[a_tracker]) then the sigmoid / tanh of csrc/atr_cell.h — bias0 / bias1 = b_ih + b_hh per player [4R];
 * emb (nullable) [n_act, 4R] = fc_action_tracker(one_hot(.)) projected through W_ih (model.py:193-194), added for player
 * emb_player with row act_tracker[t * act_tstride + n] (int64: the rollout's action store, read in place). */
int atr_lstm_bptt_pre(const float *dh0_heads, const float *dh1_heads, const float *keep, const float *pre, long long pre_pstride,
                      const float *bias0, const float *bias1, const float *emb, int emb_player, int n_act,
                      const long long *act_tracker, long long act_tstride, const float *c_all, long long c_pstride,
                      const float *whh0, const float *whh1, float *dg, long long dg_pstride, float *dh_init, float *dc_init,
                      int P, int T, int N, int R, void *stream);
/* atr_lstm_bptt_pre with one more output (round 6): act_sums (nullable; n_act must be 4) — atr_lstm_bptt_act_sums_floats(N)
 * floats, [row tiles of 16][4][4R]: per row tile of player emb_player, the column sums of dG over the tile's rows grouped by the
 * tracker's action of the row (all T steps). Their sum over the tiles, S [4][4R], is everything the tracker-action embedding
 * (TAT.forward, model.py:193-194) needs of the backward pass — see atr_embed_fold. */
long long atr_lstm_bptt_act_sums_floats(int N);
int atr_lstm_bptt_pre2(const float *dh0_heads, const float *dh1_heads, const float *keep, const float *pre, long long pre_pstride,
                       const float *bias0, const float *bias1, const float *emb, int emb_player, int n_act,
                       const long long *act_tracker, long long act_tstride, const float *c_all, long long c_pstride,
                       const float *whh0, const float *whh1, float *dg, long long dg_pstride, float *dh_init, float *dc_init,
                       float *act_sums, int P, int T, int N, int R, void *stream);
/* The embedding's share of the target's backward pass from those sums, instead of two passes over [T N, C] tensors (the learner
 * no longer materialises f + E[a] for the dW_ih product nor gathers dL/df by action): with S = sum over tiles of act_sums
 * (fixed order; written to S [4][J], J = 4R) and E[a][c] = fa_w[c][a] + fa_b[c] (fc_action_tracker.weight [C, 4], .bias [C]):
 *     dwih [J, C] += S^T E     (dwih already holds dG^T f, the product over the RAW fc features)
 *     dfa_w [C, 4] = (S wih)^T,  dfa_b [C] = its row sums     (wih = the target's weight_ih [J, C])
 * Returns 0 or a non-zero code. */
int atr_embed_fold(const float *act_sums, int tiles, const float *fa_w, const float *fa_b, const float *wih, float *dwih,
                   float *dfa_w, float *dfa_b, float *S, int J, int C, void *stream);
/* fc_action_tracker(one_hot(a_tracker)) added to the target's features over all stored steps (TAT.forward, model.py:193-194 of
 * the reference): out[r][c] = f[r][c] + w[c][action of row r] + b[c], w = fc_action_tracker.weight [C, A] (A <= 8, C
 * multiple of 4), actions int64; the action of row r is actions[(r / act_n) * act_tstride + (r % act_n) * act_stride] — a flat
 * vector is act_n >= rows; one player's column of a [T, players, N] action store is act_n = N, act_tstride = players * N. atr_embed_grad: the parameter gradients given dout = dL/dout (the gradient w.r.t. f is dout
 * itself): dw[c][a] = sum of dout[r][c] over the rows with action a, db[c] = sum over all rows; workspace:
 * atr_embed_grad_workspace_floats(rows, C, A) floats; fixed summation order (reproducible). */
int atr_embed_add(const float *f, const float *w, const float *b, const long long *actions, long long act_stride,
                  long long act_n, long long act_tstride, float *out, long long rows, int C, int A, void *stream);
/* ... with f's rows ldf floats apart (the features as a column block of the rollout's [features | k h] rows); out stays dense */
int atr_embed_add_ld(const float *f, long long ldf, const float *w, const float *b, const long long *actions,
                     long long act_stride, long long act_n, long long act_tstride, float *out, long long rows, int C, int A,
                     void *stream);
/* ReLU backward against a stored activation: out[r][c] = f[r * ldf + c] > 0 ? df[r][c] : 0 (df, out dense [rows, C]; C a
 * multiple of 4, 16-byte aligned rows) — aten::threshold_backward for a strided activation, one vectorised pass. */
int atr_relu_backward_ld(const float *df, const float *f, long long ldf, float *out, long long rows, int C, void *stream);
long long atr_embed_grad_workspace_floats(long long rows, int C, int A);
int atr_embed_grad(const float *dout, const long long *actions, long long act_stride, long long act_n, long long act_tstride,
                   float *dw, float *db, float *workspace, long long rows, int C, int A, void *stream);
/* n-step returns and GAE terms of the A3C loss (player_util.py:118-141 of the reference) for all (env, agent) pairs:
 * rewards [T,N,A], values [T+1,N,A] (row T = bootstrap value), notdone [T,N] -> returns, gae [T,N,A]. */
int atr_gae_returns(const float *rewards, const float *values, const float *notdone, float gamma, float tau,
                    float *returns, float *gae, int T, int N, int A, void *stream);

/* Heads + loss of ONE player over all stored steps (rows = T*N; R hidden units, R/4 in {16,32,64}; A <= 8 actions):
 * critic/actor heads (model.py:24-52,120-126 of the reference), softmax statistics of the taken action and the loss
 * terms of Agent.optimize (player_util.py:118-154) with analytic gradients.
 *   atr_heads_values: values[row*vstride + voff] = h[row] . wc + bc   (needed first: returns/GAE use detached values)
 *   atr_heads_loss:   ret/gae/val are read at [row*stride + off]; r_aux (nullable) at [row*aux_stride + aux_off] is
 *     the reward the aux head (waux, baux; nullable) regresses with an L1 loss. Objective contribution of this player:
 *       scale * sum_rows( -logp(a) gae - w_ent H + 0.5 * 0.5 (ret - v)^2 ) + scale_aux * sum_rows |pred - r_aux|
 *     Outputs: dh [rows,R] = dL/dh; grads_and_sums = dWa [A*R] | dWc [R] | dWaux [R] | dba [A] | dbc | dbaux |
 *     sum(-logp gae - w_ent H) | sum(0.5 (ret-v)^2) | sum(H) | sum|pred - r_aux| (the four sums are UNscaled) |
 *     the objective contribution above (one float): (A+2)*R + (A+2) + 5 floats in all.
 *     workspace: atr_heads_workspace_floats(rows, R, A) floats. Fixed reduction order (reproducible). */
int atr_heads_values(const float *h, const float *wc, const float *bc, float *values, long long rows, int R, int vstride,
                     int voff, void *stream);
/* ... for two players (different hidden rows and critic weights, columns voff / voff1 of the same values array) in one launch;
 * h1 == NULL: one player. */
int atr_heads_values2(const float *h, const float *wc, const float *bc, int voff, const float *h1, const float *wc1,
                      const float *bc1, int voff1, float *values, long long rows, int R, int vstride, void *stream);
long long atr_heads_workspace_floats(long long rows, int R, int A);
int atr_heads_loss(const float *h, const long long *actions, const float *ret, const float *gae, const float *val,
                   int stride, int off, const float *r_aux, int aux_stride, int aux_off, const float *wa,
                   const float *ba, const float *wc, const float *waux, const float *baux, float scale, float scale_aux,
                   float w_ent, float *dh, float *grads_and_sums, float *workspace, long long rows, int R, int A,
                   void *stream);
/* Both players' heads + loss terms as ONE launch + ONE reduction launch (count <= 2; same R and A): per player the arguments
 * of atr_heads_loss, with actions read in place from the rollout's [T, players, N] store (row r: actions[(r / act_n) *
 * act_tstride + r % act_n]; a flat vector is act_n >= rows), and stats_out (nullable, 4 floats) receiving the policy / value /
 * entropy / |aux error| sums x stats_scale. */
typedef struct atr_heads_loss_args {
    const float *h;
    const long long *actions;
    long long act_n, act_tstride;
    const float *ret, *gae, *val;
    int stride, off;
    const float *r_aux;
    int aux_stride, aux_off;
    const float *wa, *ba, *wc, *waux, *baux;
    float scale, scale_aux, w_ent;
    float *dh, *grads_and_sums, *workspace, *stats_out;
    long long rows;
    int R, A;
} atr_heads_loss_args;
int atr_heads_loss_multi(const atr_heads_loss_args *players, int count, float stats_scale, void *stream);

/* C [M,N] = X1^T X2 for tall row-major X1 [K,M], X2 [K,N] (the learner's weight-gradient GEMMs: K = T*N_envs rows;
 * M, N multiples of 128), fp32 on the f32 matrix cores with split-K and a fixed-order reduction (reproducible).
 * Optional: row_scale [K] (nullable) multiplies row k of X1 first (the episode mask of dW_hh = (k h)^T dG); colsum [M]
 * (nullable) receives the column sums of the (scaled) X1 — the bias gradient that goes with the weight gradient.
 * workspace: atr_gemm_tn_workspace_floats(K, M, N) floats (-1 for unsupported shapes). */
long long atr_gemm_tn_workspace_floats(long long K, int M, int N);
int atr_gemm_tn(const float *x1, const float *x2, float *c, float *workspace, long long K, int M, int N,
                const float *row_scale, float *colsum, void *stream);

/* The grouped form: up to 8 such products over the SAME K rows (all weight gradients of one backward pass) as ONE launch
 * plus ONE reduction launch. Each problem names its own destination c [M,N] (e.g. a slice of the flat gradient bucket) and up
 * to two destinations of the column sums of its (scaled) X1 (bias_ih and bias_hh of an LSTMCell receive the same gradient).
 * row_scale (nullable) with row_scale_shift s: row k of X1 is multiplied by row_scale[k - s] for k >= s, by 1 below — the
 * episode mask on h_{t-1} is keep[t-1], i.e. the keep array itself shifted by one step of N rows.
 * workspace: atr_gemm_tn_grouped_workspace_floats(problems, count, K) floats (-1: unsupported shapes or count > 8). */
/* count <= atr_scatter_max_segments segments (src[q] (null: zeros), n[q] floats) copied to dst + dst_off[q] in one launch:
 * autograd's per-parameter gradient tensors -> the flat gradient bucket (shared_optim.FlatParams.set_grads). */
#define atr_scatter_max_segments 64
int atr_scatter_segments(const float *const *src, const long long *dst_off, const int *n, int count, float *dst, void *stream);

typedef struct atr_gemm_tn_problem {
    const float *x1, *x2;
    float *c;
    const float *row_scale;
    long long row_scale_shift;
    float *colsum0, *colsum1;
    int M, N;
    long long ld1, ld2;      /* row strides of x1 / x2 in floats (0 = dense: M / N); multiples of 4 */
} atr_gemm_tn_problem;
long long atr_gemm_tn_grouped_workspace_floats(const atr_gemm_tn_problem *problems, int count, long long K);
/* Co-run mode of the weight-gradient kernel, a setting of the CALLING THREAD, returns the previous setting: while on, launches (and the workspace
 * sizes that go with them) are planned for ONE workgroup per CU, so that a chain of short kernels on another stream keeps
 * running beside them (the pipelined schedule captures its learner graphs in this mode; results stay a fixed-order sum, the
 * K split differs from the default mode's). Not a per-stream setting: set it around the launches (or the graph capture) that
 * want it, from the thread that issues them; launches and workspace-size queries of other threads are not affected. */
int atr_gemm_tn_set_corun(int on);
int atr_gemm_tn_grouped(const atr_gemm_tn_problem *problems, int count, long long K, float *workspace, void *stream);

/* The rollout driver's bookkeeping, one launch each (csrc/driver_hip.hip).
 *   atr_rollout_begin: what Agent keeps between rollouts (player_util.py:98-106: hxs/cxs [N,A,R]) -> slot 0 of the
 *     per-player rollout stores h0/c0 (player p at + p*pstride floats, rows [N,R]); optionally the current observation
 *     (obs_bytes bytes, multiple of 4) -> row 0 of the rollout's observation store.
 *   atr_rollout_end: the final state (slot T, same addressing) -> hxs/cxs [N,A,R], zeroed for envs whose last step
 *     finished an episode (the reset() of train.py:73-74); eps_len [N] int32 advanced as train.py:75-76 would have step by
 *     step (restart at a done, +1 per step); keep [T,N] float = (dones == 0), the episode mask the learner's kernels read.
 *   atr_adam_step: SharedAdam.step (shared_optim.py:122-175 of the reference: Adam with AMSGrad when max_exp_avg_sq != NULL,
 *     eps added to the raw square root, both bias corrections folded into the step size) over a flat fp32 bucket of n
 *     elements; torch_eps != 0 gives torch.optim.Adam's form instead (the reference's per-worker optimizer when
 *     --shared-optimizer is absent, train.py:45-49: sqrt(v / (1 - beta2^t)) + eps). state = {step, beta1^step, beta2^step}
 *     float64 on the device (advanced here); scalars: two floats of scratch.
 *   atr_rmsprop_step: SharedRMSprop.step (shared_optim.py:43-87; momentum 0, not centered, as main.py:89-90 builds it) =
 *     torch.optim.RMSprop with the same settings. */
int atr_rollout_begin(const float *hxs, const float *cxs, float *h0, float *c0, long long pstride, const void *obs_src,
                      void *obs_dst, long long obs_bytes, int N, int A, int R, void *stream);
/* ... and, in the same launch, the per-rollout constants of the actor (weights are fixed inside a rollout; every output
 * optional): bsum [2,4R] = b_ih + b_hh; w_cat [2,4R,F+R] = [W_ih | W_hh] (the one-GEMM LSTMCell); emb_ih [A,4R] =
 * (fc_action_tracker.weight^T + bias) W_ih[1]^T (model.py:193-194 projected through the target's input weights); the action
 * sampler's counter += 1; the hidden columns of slot 0 of the [features | k h] rows <- hxs. A (players) must be 2. */
typedef struct atr_rollout_consts {
    const float *w_ih[2], *w_hh[2], *b_ih[2], *b_hh[2];
    float *bsum, *w_cat;
    const float *fa_w, *fa_b;
    float *emb_ih;
    unsigned long long *counter;
    float *fh0;
    long long fh_pstride, fh_ld;
    int F, A_act;
} atr_rollout_consts;
int atr_rollout_begin2(const float *hxs, const float *cxs, float *h0, float *c0, long long pstride, const void *obs_src,
                       void *obs_dst, long long obs_bytes, int N, int A, int R, const atr_rollout_consts *consts, void *stream);
int atr_rollout_end(const float *hT, const float *cT, long long pstride, const uint8_t *dones, float *hxs, float *cxs,
                    int *eps_len, float *keep, int T, int N, int A, int R, void *stream);
/* ... and, in the same launch, what else the next rollout starts from: the observation after the last step (obs_bytes
 * bytes, multiple of 4; nullable pair) -> obs_dst, and the last step's done flags -> done_dst [N] (nullable). */
int atr_rollout_end2(const float *hT, const float *cT, long long pstride, const uint8_t *dones, float *hxs, float *cxs,
                     int *eps_len, float *keep, int T, int N, int A, int R, const void *obs_src, void *obs_dst,
                     long long obs_bytes, uint8_t *done_dst, void *stream);
int atr_adam_step(float *params, const float *grad, float *exp_avg, float *exp_avg_sq, float *max_exp_avg_sq, double *state,
                  float *scalars, double lr, double beta1, double beta2, double eps, double weight_decay, int torch_eps,
                  long long n, void *stream);
int atr_rmsprop_step(float *params, const float *grad, float *square_avg, double lr, double alpha, double eps,
                     double weight_decay, long long n, void *stream);

#ifdef __cplusplus
}
#endif
#endif
