// policy_hip.hip — small fused policy-side kernels (C ABI in include/atr_policy.h).
//
// atr_sample_actions: the actor head of the rollout in ONE launch — logits = W h + b (n_actions x R, R <= 256),
// softmax, and one categorical draw per env by inverse CDF on a Philox4x32-10 uniform. It replaces
// actor_linear -> softmax -> torch.multinomial (the reference's sample_action, model.py:41-49), which on ROCm is a
// GEMM plus ~12 tiny launches (multinomial alone: min/max/nan asserts, exponential noise, divide, argmax ...).
// The stream position is (device-side counter, host-side ordinal): the counter is advanced in stream order (by the
// call itself when bump != 0, e.g. once per rollout), so a hipGraph replay draws fresh numbers; the ordinal tells the
// launches between two bumps apart. Sixteen lanes per env.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/atr_policy.h"
#include "atr_sample.h"

namespace atr {

constexpr int kMaxR = 256;

// 16 lanes per env: each lane loads 2 x 16 B of the env's hidden row (coalesced 512 B per env) and the matching slices of
// the head weights straight from memory (3.5 KB, L2-resident and shared by every env: one round trip for everything —
// a first version staged them in LDS behind a workgroup barrier, i.e. two dependent round trips for a 5 us kernel),
// accumulates its partial logits, the 16 partials are summed on the DPP/permute network and sub-lane 0 draws the action.
__global__ __launch_bounds__(256) void k_sample_actions(const float *__restrict__ h, const float *__restrict__ w,
                                                        const float *__restrict__ b, long long *__restrict__ actions,
                                                        const unsigned long long *__restrict__ counter,
                                                        unsigned long long seed, unsigned ordinal, int n, int R, int A)
{
    const int sub = (int)(threadIdx.x & 15u);
    const int e = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 4);
    const bool valid = e < n;
    const int er = valid ? e : n - 1;                       // tail lanes shadow the last env (they never store)
    const int nj = R / 4;                                   // float4s per row; lane `sub` takes j = sub, sub + 16, ...
    const unsigned long long ctr = *counter;
    float logit[kMaxActions];
#pragma unroll
    for (int a = 0; a < kMaxActions; a++) logit[a] = 0.f;
    const float4 *hr = reinterpret_cast<const float4 *>(h + (size_t)er * R);
    const float4 *wr = reinterpret_cast<const float4 *>(w);
    for (int j0 = sub; j0 < nj; j0 += 32) {                 // two float4s of the row per trip: all their loads go out together
        const int j1 = j0 + 16;
        const bool two = j1 < nj;
        const float4 v0 = hr[j0], v1 = two ? hr[j1] : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 w0[kMaxActions], w1[kMaxActions];
#pragma unroll
        for (int a = 0; a < kMaxActions; a++) {
            w0[a] = a < A ? wr[a * nj + j0] : make_float4(0.f, 0.f, 0.f, 0.f);
            w1[a] = (a < A && two) ? wr[a * nj + j1] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int a = 0; a < kMaxActions; a++) {
            logit[a] = fmaf(v0.x, w0[a].x, fmaf(v0.y, w0[a].y, fmaf(v0.z, w0[a].z, fmaf(v0.w, w0[a].w, logit[a]))));
            if (two) logit[a] = fmaf(v1.x, w1[a].x, fmaf(v1.y, w1[a].y, fmaf(v1.z, w1[a].z, fmaf(v1.w, w1[a].w, logit[a]))));
        }
    }
#pragma unroll
    for (int a = 0; a < kMaxActions; a++)
        if (a < A) {
            float v = logit[a];
            v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
            logit[a] = v + b[a];
        } else {
            logit[a] = -INFINITY;
        }
    if (!valid || sub != 0) return;
    const int act = draw_action(logit, A, e, ctr, seed, ordinal);
    actions[e] = (long long)act;
}

__global__ void k_bump_counter(unsigned long long *counter) { *counter += 1ull; }

} // namespace atr

extern "C" int atr_sample_actions(const float *h, const float *w, const float *b, long long *actions,
                                  unsigned long long *counter, unsigned long long seed, unsigned ordinal, int bump,
                                  int n, int R, int A, void *stream)
{
    if (!h || !w || !b || !actions || !counter || n < 0 || R <= 0 || R > atr::kMaxR || (R & 3) || A < 1 || A > atr::kMaxActions)
        return -1;
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        if (bump) hipLaunchKernelGGL(atr::k_bump_counter, dim3(1), dim3(1), 0, st, counter);
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    hipLaunchKernelGGL(atr::k_sample_actions, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, st, h, w, b, actions,
                       counter, seed, ordinal, n, R, A);
    if (bump) hipLaunchKernelGGL(atr::k_bump_counter, dim3(1), dim3(1), 0, st, counter);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
