// policy_hip.hip — small fused policy-side kernels (C ABI in include/atr_policy.h).
//
// atr_sample_actions: the actor head of the rollout in ONE launch — logits = W h + b (n_actions x R, R <= 256),
// softmax, and one categorical draw per env by inverse CDF on a Philox4x32-10 uniform. It replaces
// actor_linear -> softmax -> torch.multinomial (the reference's sample_action, model.py:41-49), which on ROCm is a
// GEMM plus ~12 tiny launches (multinomial alone: min/max/nan asserts, exponential noise, divide, argmax ...).
// The stream position is (device-side counter, host-side ordinal): the counter is advanced in stream order (by the
// call itself when bump != 0, e.g. once per rollout), so a hipGraph replay draws fresh numbers; the ordinal tells the
// launches between two bumps apart. Sixteen lanes per env; the head weights sit in LDS.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/atr_policy.h"
#include "atr_sample.h"

namespace atr {

constexpr int kMaxR = 256;

// 16 lanes per env: each lane loads 2 x 16 B of the env's hidden row (coalesced 512 B per env), accumulates its
// partial logits, the 16 partials are summed on the DPP/permute network and sub-lane 0 draws the action.
__global__ __launch_bounds__(256) void k_sample_actions(const float *__restrict__ h, const float *__restrict__ w,
                                                        const float *__restrict__ b, long long *__restrict__ actions,
                                                        const unsigned long long *__restrict__ counter,
                                                        unsigned long long seed, unsigned ordinal, int n, int R, int A)
{
    __shared__ float ws[kMaxActions * kMaxR + kMaxActions];
    for (int i = (int)threadIdx.x; i < A * R; i += (int)blockDim.x) ws[i] = w[i];
    if ((int)threadIdx.x < A) ws[A * R + threadIdx.x] = b[threadIdx.x];
    __syncthreads();
    const int sub = (int)(threadIdx.x & 15u);
    const int e = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 4);
    const bool valid = e < n;
    float logit[kMaxActions];
#pragma unroll
    for (int a = 0; a < kMaxActions; a++) logit[a] = 0.f;
    if (valid) {
        const float4 *hr = reinterpret_cast<const float4 *>(h + (size_t)e * R);
        for (int j = sub; j < R / 4; j += 16) {
            const float4 v = hr[j];
#pragma unroll
            for (int a = 0; a < kMaxActions; a++)
                if (a < A) {
                    const float *wa = ws + a * R + 4 * j;
                    logit[a] = fmaf(v.x, wa[0], fmaf(v.y, wa[1], fmaf(v.z, wa[2], fmaf(v.w, wa[3], logit[a]))));
                }
        }
    }
#pragma unroll
    for (int a = 0; a < kMaxActions; a++)
        if (a < A) {
            float v = logit[a];
            v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
            logit[a] = v + ws[A * R + a];
        } else {
            logit[a] = -INFINITY;
        }
    if (!valid || sub != 0) return;
    const int act = draw_action(logit, A, e, *counter, seed, ordinal);
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
