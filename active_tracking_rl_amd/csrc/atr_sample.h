// atr_sample.h — Philox4x32-10 and the inverse-CDF categorical draw shared by the policy-side kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace atr {

constexpr int kMaxActions = 8;

__device__ __forceinline__ void philox4x32_10(uint32_t k0, uint32_t k1, uint32_t &c0, uint32_t &c1, uint32_t &c2, uint32_t &c3)
{
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

// One draw from Categorical(softmax(logit[0..A))) for row e: uniform keyed (seed; e, counter, ordinal).
__device__ __forceinline__ int draw_action(const float (&logit)[kMaxActions], int A, int e, unsigned long long c,
                                           unsigned long long seed, unsigned ordinal)
{
    float mx = logit[0];
#pragma unroll
    for (int a = 1; a < kMaxActions; a++) mx = fmaxf(mx, logit[a]);
    float p[kMaxActions], sum = 0.f;
#pragma unroll
    for (int a = 0; a < kMaxActions; a++) { p[a] = a < A ? __expf(logit[a] - mx) : 0.f; sum += p[a]; }
    uint32_t c0 = (uint32_t)e, c1 = (uint32_t)c, c2 = (uint32_t)(c >> 32), c3 = 0x5A3D0000u ^ ordinal;
    philox4x32_10((uint32_t)seed, (uint32_t)(seed >> 32), c0, c1, c2, c3);
    const float u = ((c0 >> 8) + 0.5f) * (1.0f / 16777216.0f) * sum;    // uniform in (0, sum)
    int act = A - 1;                                      // first a with u < cumulative(a)
    float acc = 0.f;
    bool found = false;
#pragma unroll
    for (int a = 0; a < kMaxActions; a++) {
        if (a < A) {
            acc += p[a];
            if (!found && u < acc) { act = a; found = true; }
        }
    }
    return act;
}

} // namespace atr
