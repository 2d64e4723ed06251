// pair_gemm_hip.hip — the rollout step's two small GEMM pairs as ONE launch each, for SMALL row counts (the shards of the
// strong-scaling form: 512 / 1024 / 2048 envs per GPU), where a library GEMM is 8-10 us of launch floor and load latency
// around 1-3 us of matrix-pipe work and the step issues four of them per player pair:
//
//     atr_pair_linear:   C[p] = act( A1[p] W1[p]^T [+ (k A2[p]) W2[p]^T] + bias[p] ),   p = 0, 1 (the two players)
//
//   * fc + ReLU of CNN_maze (perception.py:81,90 of the reference): A1 = the stem output [M, 512 | 1024], W1 = fc.weight
//     [256, 512 | 1024] (the tracker-aware target's encoder sees two frames: K differs per player), ReLU;
//   * both GEMMs of nn.LSTMCell (model.py:110,137,172,203): A1 = fc features [M, 256], W1 = weight_ih [512, 256], A2 = the
//     previous hidden state [M, 128] whose rows are scaled by the episode mask k = (done == 0) on their way in, W2 =
//     weight_hh [512, 128], bias = b_ih + b_hh: the gate pre-activations in one pass (no ig / hg pair to re-read).
// Layout: every operand is K-contiguous (activations row-major, nn.Linear weights [out, in]).
//
// One workgroup = one 32 x 32 output tile of one player, the contraction split over its four waves (so that a 512-row
// problem still has 256-512 workgroups = every CU of the chip, and a wave's chain is K/4 long): v_mfma_f32_32x32x2_f32
// (exact f32), operands fetched with coalesced 16-B loads (8 lanes cover one 128-B line of a row), parked in a
// wave-private LDS slice and read back in MFMA layout (the scheme of actor_step_hip.hip: 36-float row stride, no bank
// conflicts, no barrier), loads two K-blocks ahead of the MFMAs; the four partial tiles meet in LDS (one barrier), every
// thread adds four values + bias and stores 16 B. The column tiles of a row tile run on one XCD (workgroup b runs on XCD
// b % 8), so the activation rows come from memory once.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/atr_policy.h"

namespace atr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kPgBlk = 32;                 // K block
constexpr int kPgLs = kPgBlk + 4;          // LDS row stride (floats)
constexpr int kPgOp = 32 * kPgLs;          // one staged operand (32 rows x 32 k)

struct PairLinear {
    const float *a1[2], *w1[2], *a2[2], *w2[2], *bias[2];
    float *c[2];
    long long lda1[2], lda2[2], ldc[2];
    int k1[2], k2[2];
    const unsigned char *done;     // nullable [M]: A2's rows are scaled by (done == 0)
    int M, N, relu;
};

__global__ __launch_bounds__(256) void k_pair_linear(PairLinear g)
{
    __shared__ __attribute__((aligned(16))) float lds[4][2 * 2 * kPgOp];        // per wave: 2 buffers x {A, B}: 18 KB
    const int tid = (int)threadIdx.x, l = tid & 63, wave = tid >> 6;
    const int rt_n = (g.M + 31) >> 5, ct_n = g.N >> 5;
    const int per_player = ((rt_n + 7) >> 3) * 8 * ct_n;
    int b = (int)blockIdx.x;
    const int p = b >= per_player ? 1 : 0;
    b -= p * per_player;
    // (group of 8 row tiles) x (column tile) x (row tile within the group = XCD)
    const int grp = b / (8 * ct_n), rem = b - grp * 8 * ct_n;
    const int ct = rem >> 3, rt = grp * 8 + (rem & 7);
    if (rt >= rt_n) return;                                   // (whole workgroup: before any barrier)
    const int row0 = rt * 32, col0 = ct * 32;
    const float *A1 = g.a1[p], *W1 = g.w1[p], *A2 = g.a2[p], *W2 = g.w2[p];
    const long long lda1 = g.lda1[p], lda2 = g.lda2[p];
    const int k1 = g.k1[p], k2 = A2 ? g.k2[p] : 0;
    const int nb1 = k1 / kPgBlk, nb = nb1 + k2 / kPgBlk;
    const int bw0 = (nb * wave) >> 2, bw1 = (nb * (wave + 1)) >> 2;      // this wave's K blocks
    // staging: lane (sub-row rs = l >> 3, k4 = l & 7) fetches float4 X[rs + 8 i][32 blk + 4 k4], i = 0..3
    const int rs = l >> 3, k4 = l & 7;
    int ar[4];
#pragma unroll
    for (int i = 0; i < 4; i++) ar[i] = min(row0 + rs + 8 * i, g.M - 1);          // tail rows shadow the last one
    float4 sa0, sa1, sa2, sa3, sb0, sb1, sb2, sb3;
#define PG_LD(p_) (*reinterpret_cast<const float4 *>(p_))
#define PG_LOAD_BLOCK(blk)                                                                                             \
    do {                                                                                                               \
        const int b_ = (blk);                                                                                          \
        if (b_ < nb1) {                                                                                                \
            const float *pa_ = A1 + kPgBlk * b_ + 4 * k4, *pw_ = W1 + (size_t)(col0 + rs) * k1 + kPgBlk * b_ + 4 * k4; \
            sa0 = PG_LD(pa_ + ar[0] * lda1); sa1 = PG_LD(pa_ + ar[1] * lda1);                                          \
            sa2 = PG_LD(pa_ + ar[2] * lda1); sa3 = PG_LD(pa_ + ar[3] * lda1);                                          \
            sb0 = PG_LD(pw_); sb1 = PG_LD(pw_ + (size_t)8 * k1); sb2 = PG_LD(pw_ + (size_t)16 * k1);                   \
            sb3 = PG_LD(pw_ + (size_t)24 * k1);                                                                        \
        } else {                                                                                                       \
            const int o_ = kPgBlk * (b_ - nb1) + 4 * k4;                                                               \
            const float *pa_ = A2 + o_, *pw_ = W2 + (size_t)(col0 + rs) * k2 + o_;                                     \
            sa0 = PG_LD(pa_ + ar[0] * lda2); sa1 = PG_LD(pa_ + ar[1] * lda2);                                          \
            sa2 = PG_LD(pa_ + ar[2] * lda2); sa3 = PG_LD(pa_ + ar[3] * lda2);                                          \
            sb0 = PG_LD(pw_); sb1 = PG_LD(pw_ + (size_t)8 * k2); sb2 = PG_LD(pw_ + (size_t)16 * k2);                   \
            sb3 = PG_LD(pw_ + (size_t)24 * k2);                                                                        \
        }                                                                                                              \
    } while (0)
    float *mybuf = lds[wave];
    float *stp = mybuf + rs * kPgLs + 4 * k4;
#define PG_ST4(dst, v) (*reinterpret_cast<float4 *>(dst) = (v))
#define PG_STORE_BLOCK(i_)                                                                                             \
    do {                                                                                                               \
        float *d_ = stp + ((i_) & 1) * 2 * kPgOp;                                                                      \
        PG_ST4(d_, sa0); PG_ST4(d_ + 8 * kPgLs, sa1); PG_ST4(d_ + 16 * kPgLs, sa2); PG_ST4(d_ + 24 * kPgLs, sa3);      \
        d_ += kPgOp;                                                                                                   \
        PG_ST4(d_, sb0); PG_ST4(d_ + 8 * kPgLs, sb1); PG_ST4(d_ + 16 * kPgLs, sb2); PG_ST4(d_ + 24 * kPgLs, sb3);      \
    } while (0)
    const int jj = l & 31, kk = l >> 5;                       // MFMA lane coordinates: row / column index, K slot
    // the episode mask rides on operand A of the second term: (k h) W == k (h W)
    float akeep = 1.0f;
    if (g.done && A2) akeep = g.done[min(row0 + jj, g.M - 1)] == 0 ? 1.0f : 0.0f;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; r++) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
    const int nbw = bw1 - bw0;
    if (nbw > 0) {
        PG_LOAD_BLOCK(bw0);
        PG_STORE_BLOCK(0);
        if (nbw > 1) PG_LOAD_BLOCK(bw0 + 1);
        const float *rd = mybuf + jj * kPgLs + 4 * kk;
#pragma unroll 1
        for (int i = 0; i < nbw; i++) {
            if (i + 1 < nbw) PG_STORE_BLOCK(i + 1);           // block i + 1 (in registers since the previous trip) -> LDS
            if (i + 2 < nbw) PG_LOAD_BLOCK(bw0 + i + 2);      // block i + 2: memory -> registers, under this block's MFMAs
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            const float *src = rd + (i & 1) * 2 * kPgOp;
            const float sc = (bw0 + i) >= nb1 ? akeep : 1.0f;
#pragma unroll
            for (int cc = 0; cc < kPgBlk / 8; cc++) {
                float4 av = *reinterpret_cast<const float4 *>(src + 8 * cc);
                const float4 bv = *reinterpret_cast<const float4 *>(src + kPgOp + 8 * cc);
                av.x *= sc; av.y *= sc; av.z *= sc; av.w *= sc;
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc1, 0, 0, 0);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
#undef PG_LOAD_BLOCK
#undef PG_STORE_BLOCK
    // partial tile of this wave -> its own LDS slice as [row][col] (33-float rows), then every thread sums four partials
    constexpr int kPs = 33;
    float *part = mybuf;
#pragma unroll
    for (int r = 0; r < 16; r++) part[(8 * (r >> 2) + 4 * kk + (r & 3)) * kPs + jj] = acc0[r] + acc1[r];
    __syncthreads();
    const int orow = tid >> 3, oc = (tid & 7) * 4;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g.bias[p]) o = *reinterpret_cast<const float4 *>(g.bias[p] + col0 + oc);
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const float *s = lds[w] + orow * kPs + oc;
        o.x += s[0]; o.y += s[1]; o.z += s[2]; o.w += s[3];
    }
    if (g.relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
    if (row0 + orow < g.M) *reinterpret_cast<float4 *>(g.c[p] + (size_t)(row0 + orow) * g.ldc[p] + col0 + oc) = o;
}

} // namespace atr

using namespace atr;

extern "C" int atr_pair_linear(const atr_pair_linear_args *a, void *stream)
{
    if (!a || a->M <= 0 || a->N <= 0 || (a->N & 31)) return -1;
    PairLinear g;
    for (int p = 0; p < 2; p++) {
        if (!a->a1[p] || !a->w1[p] || !a->c[p] || a->k1[p] <= 0 || (a->k1[p] & 31)) return -1;
        if ((a->a2[p] != nullptr) != (a->w2[p] != nullptr)) return -1;
        if (a->a2[p] && (a->k2[p] <= 0 || (a->k2[p] & 31))) return -1;
        const int nb = (a->k1[p] + (a->a2[p] ? a->k2[p] : 0)) / kPgBlk;
        if (nb < 4) return -1;
        if ((a->lda1[p] & 3) || (a->a2[p] && (a->lda2[p] & 3)) || (a->ldc[p] & 3)) return -1;
        if (((uintptr_t)a->a1[p] | (uintptr_t)a->w1[p] | (uintptr_t)a->c[p] | (uintptr_t)a->a2[p] | (uintptr_t)a->w2[p] |
             (uintptr_t)a->bias[p]) & 15u)
            return -1;
        g.a1[p] = a->a1[p]; g.w1[p] = a->w1[p]; g.a2[p] = a->a2[p]; g.w2[p] = a->w2[p]; g.bias[p] = a->bias[p];
        g.c[p] = a->c[p]; g.lda1[p] = a->lda1[p]; g.lda2[p] = a->lda2[p]; g.ldc[p] = a->ldc[p];
        g.k1[p] = a->k1[p]; g.k2[p] = a->k2[p];
    }
    g.done = a->done; g.M = a->M; g.N = a->N; g.relu = a->relu;
    const int rt_n = (a->M + 31) / 32, ct_n = a->N / 32;
    const unsigned grid = 2u * (unsigned)(((rt_n + 7) / 8) * 8 * ct_n);
    hipLaunchKernelGGL(k_pair_linear, dim3(grid), dim3(256), 0, (hipStream_t)stream, g);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
