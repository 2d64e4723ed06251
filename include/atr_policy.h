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
/* Floats of scratch atr_stem_backward needs for M frames (per-wave partial gradient records). */
long long atr_stem_workspace_floats(long long M);
/* Gradients of the four parameter tensors given dy = dL/dy (the observation needs no gradient). Outputs are
 * overwritten (not accumulated); the reduction order is fixed, so results are run-to-run deterministic. */
int atr_stem_backward(const float *x, long long x_stride, const float *y, const float *dy, const float *w1,
                      const float *b1, const float *w2, float *dw1, float *db1, float *dw2, float *db2,
                      float *workspace, long long M, void *stream);

/* Actor head of the rollout in one launch (replaces actor_linear -> softmax -> multinomial, model.py:41-49 of the
 * reference): logits = w h + b with h [n,R] (R <= 256, multiple of 4), w [A,R], b [A], A <= 8; one categorical draw
 * per row by inverse CDF on a Philox4x32-10 uniform keyed (seed; row, *counter). `counter` is a device-side
 * uint64 the call advances by one (in stream order), so hipGraph replays draw fresh numbers. actions: int64 [n]. */
int atr_sample_actions(const float *h, const float *w, const float *b, long long *actions, unsigned long long *counter,
                       unsigned long long seed, int n, int R, int A, void *stream);

#ifdef __cplusplus
}
#endif
#endif
