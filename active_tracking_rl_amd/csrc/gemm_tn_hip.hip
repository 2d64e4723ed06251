// gemm_tn_hip.hip — C[M,N] = X1^T X2 for two TALL row-major operands X1 [K,M], X2 [K,N] (K = T*N_envs = 81 920 rows,
// M, N in {128 .. 1024}): the weight-gradient GEMMs of the learner (dW_fc = dpre^T x, dW_ih = dG^T feat,
// dW_hh = h^T dG). The library kernels TunableOp picks run these long-K / small-output shapes at 74-110 TFLOP/s (28 for the
// M = 128 case as a plain mm); this kernel is built for exactly this shape class on the f32 matrix cores (96-101
// TFLOP/s on all of them, 2.5% of an A3C iteration saved; also tighter error than the library's split-K order):
//   * 128 x 128 output tile per workgroup (4 waves, each a 64 x 64 quadrant = 2 x 2 v_mfma_f32_32x32x2_f32 tiles,
//     64 accumulator VGPRs), the K range split over many workgroups (split-K) so that ~512 workgroups exist;
//   * both operands are consumed in their natural row-major layout: a 32-row chunk of X1 and of X2 is staged in LDS
//     (unpadded rows: the two k-rows an MFMA operand read touches are served in different LDS passes; padding them
//     apart was measured 3-10 % slower) and an MFMA operand is ONE ds_read_b32 per lane (A[i = m][k] = X1[k][m], B[k][j] = X2[k][n]: lane l reads row k0 + (l>>5),
//     column base + (l & 31)); the next chunk's global loads are in flight while the current one is multiplied;
//   * XCD-aware workgroup order: the output tiles of one K-slice run on the same XCD back to back, so each operand
//     slice is fetched from HBM once and re-read from that XCD's L2 by the other tiles;
//   * split-K partials are reduced in a fixed order by a second kernel (reproducible sums, no atomics).
// fp32 MFMA is an exact fmaf chain, so this is the reference's arithmetic type.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/atr_policy.h"

namespace atr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kTile = 128, kKC = 32, kLd = 128;   // tile side, K-chunk rows, LDS row length (floats)
constexpr int kGemmThreads = 256;

struct GemmLds { float a[2][kKC][kLd]; float b[2][kKC][kLd]; };   // 64 KB: two workgroups per CU

__global__ __launch_bounds__(kGemmThreads, 2) void k_gemm_tn(const float *__restrict__ x1, const float *__restrict__ x2,
                                                             float *__restrict__ partial, long long K, int M, int N,
                                                             int slices, int chunks_per_slice,
                                                             const float *__restrict__ row_scale,
                                                             float *__restrict__ colsum_partial)
{
    __shared__ __attribute__((aligned(16))) GemmLds s;
    const int tid = (int)threadIdx.x, l = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = N / kTile, tiles = (M / kTile) * tiles_n;
    // XCD-aware order: workgroup i runs on XCD i % 8; give each XCD whole K-slices (all their tiles back to back)
    const int i = (int)blockIdx.x, xcd = i & 7, in_xcd = i >> 3;
    const int slice = xcd + 8 * (in_xcd / tiles), tile = in_xcd % tiles;
    if (slice >= slices) return;
    const int m0 = (tile / tiles_n) * kTile, n0 = (tile % tiles_n) * kTile;
    const long long k_begin = (long long)slice * chunks_per_slice * kKC;
    long long k_end = k_begin + (long long)chunks_per_slice * kKC;
    if (k_end > K) k_end = K;
    // global -> LDS staging: thread t moves rows r, r + 8, r + 16, r + 24: 16 B at column c4 * 4 of each operand
    const int r = tid >> 5, c4 = tid & 31;
    // NB: predicated `if (in range) v = p[i]` loads, not `cond ? p[i] : zero` — the select form makes the compiler
    // merge the two sources into a generic pointer and emit flat_load, which also ticks lgkmcnt and so serialises
    // the global prefetch behind every LDS wait of the MFMA loop
    const float4 *g1 = reinterpret_cast<const float4 *>(x1 + m0 + c4 * 4), *g2 = reinterpret_cast<const float4 *>(x2 + n0 + c4 * 4);
    const long long ldm = M / 4, ldn = N / 4;                          // row strides in float4 units
    float4 ra0, ra1, rb0, rb1, ra2, ra3, rb2, rb3;
    // optional extras: row_scale[k] multiplies row k of X1 on its way into LDS (dW_hh needs the episode mask on h);
    // colsum_partial receives the column sums of X1 (the bias gradient that goes with a weight gradient) from the
    // workgroups of the first tile column, which add up their staged chunks
    const bool do_colsum = colsum_partial != nullptr && n0 == 0;
    float cs = 0.f;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
#define GEMM_TN_FETCH(k0_)                                                     \
    {                                                                          \
        const long long ka_ = (k0_) + r, kb_ = (k0_) + r + 8, kc_ = (k0_) + r + 16, kd_ = (k0_) + r + 24;  \
        ra0 = zero4; ra1 = zero4; rb0 = zero4; rb1 = zero4; ra2 = zero4; ra3 = zero4; rb2 = zero4; rb3 = zero4; \
        if (ka_ < k_end) { ra0 = g1[ka_ * ldm]; rb0 = g2[ka_ * ldn]; }         \
        if (kb_ < k_end) { ra1 = g1[kb_ * ldm]; rb1 = g2[kb_ * ldn]; }         \
        if (kc_ < k_end) { ra2 = g1[kc_ * ldm]; rb2 = g2[kc_ * ldn]; }         \
        if (kd_ < k_end) { ra3 = g1[kd_ * ldm]; rb3 = g2[kd_ * ldn]; }         \
        if (row_scale) {                                                       \
            const float s0_ = ka_ < k_end ? row_scale[ka_] : 0.f, s1_ = kb_ < k_end ? row_scale[kb_] : 0.f;   \
            const float s2_ = kc_ < k_end ? row_scale[kc_] : 0.f, s3_ = kd_ < k_end ? row_scale[kd_] : 0.f;   \
            ra0.x *= s0_; ra0.y *= s0_; ra0.z *= s0_; ra0.w *= s0_; ra1.x *= s1_; ra1.y *= s1_; ra1.z *= s1_; ra1.w *= s1_; \
            ra2.x *= s2_; ra2.y *= s2_; ra2.z *= s2_; ra2.w *= s2_; ra3.x *= s3_; ra3.y *= s3_; ra3.z *= s3_; ra3.w *= s3_; \
        }                                                                      \
    }
#define GEMM_TN_STAGE(buf_)                                                    \
    {                                                                          \
        *reinterpret_cast<float4 *>(&s.a[buf_][r][c4 * 4]) = ra0;              \
        *reinterpret_cast<float4 *>(&s.a[buf_][r + 8][c4 * 4]) = ra1;          \
        *reinterpret_cast<float4 *>(&s.b[buf_][r][c4 * 4]) = rb0;              \
        *reinterpret_cast<float4 *>(&s.b[buf_][r + 8][c4 * 4]) = rb1;          \
        *reinterpret_cast<float4 *>(&s.a[buf_][r + 16][c4 * 4]) = ra2;         \
        *reinterpret_cast<float4 *>(&s.a[buf_][r + 24][c4 * 4]) = ra3;         \
        *reinterpret_cast<float4 *>(&s.b[buf_][r + 16][c4 * 4]) = rb2;         \
        *reinterpret_cast<float4 *>(&s.b[buf_][r + 24][c4 * 4]) = rb3;         \
    }
    f32x16 acc00, acc01, acc10, acc11;
#pragma unroll
    for (int q = 0; q < 16; q++) { acc00[q] = 0.f; acc01[q] = 0.f; acc10[q] = 0.f; acc11[q] = 0.f; }
    GEMM_TN_FETCH(k_begin);
    GEMM_TN_STAGE(0);
    __syncthreads();
    int buf = 0;
    const int kr = l >> 5, col = l & 31;
    // (a second register set, i.e. two chunks of prefetch distance, was measured: no faster — with two workgroups per
    // CU the other one's MFMAs already cover the load latency)
    for (long long k0 = k_begin; k0 < k_end; k0 += kKC, buf ^= 1) {
        const bool more = k0 + kKC < k_end;
        if (more) GEMM_TN_FETCH(k0 + kKC);               // next chunk's loads fly under this chunk's MFMAs
        const float *pa = &s.a[buf][kr][wm * 64 + col], *pb = &s.b[buf][kr][wn * 64 + col];
        float a0 = pa[0], a1 = pa[32], b0 = pb[0], b1 = pb[32];
#pragma unroll
        for (int kp = 0; kp < kKC / 2; kp++) {
            float na0 = 0.f, na1 = 0.f, nb0 = 0.f, nb1 = 0.f;
            if (kp + 1 < kKC / 2) {                      // operands of the next k-pair are read under this one's MFMAs
                na0 = pa[(kp + 1) * 2 * kLd]; na1 = pa[(kp + 1) * 2 * kLd + 32];
                nb0 = pb[(kp + 1) * 2 * kLd]; nb1 = pb[(kp + 1) * 2 * kLd + 32];
            }
            acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc00, 0, 0, 0);
            acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc01, 0, 0, 0);
            acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc10, 0, 0, 0);
            acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc11, 0, 0, 0);
            a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
        }
        if (do_colsum) {                                 // thread = (column tid & 127, half of the chunk's rows)
            const float *pc = &s.a[buf][(tid >> 7) * (kKC / 2)][tid & 127];
#pragma unroll
            for (int q = 0; q < kKC / 2; q++) cs += pc[q * kLd];
        }
        if (more) GEMM_TN_STAGE(buf ^ 1);                // the other buffer was last read one iteration ago
        __syncthreads();
    }
    if (do_colsum) {
        float *red = &s.a[0][0][0];
        if (tid >= 128) red[tid - 128] = cs;
        __syncthreads();
        if (tid < 128) colsum_partial[(size_t)slice * M + m0 + tid] = cs + red[tid];
    }
#undef GEMM_TN_FETCH
#undef GEMM_TN_STAGE
    // C/D map of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    float *out = partial + (size_t)slice * M * N;
    auto store = [&](const f32x16 &acc, int mt, int nt) {
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int row = (q & 3) + 8 * (q >> 2) + 4 * kr;
            out[(size_t)(m0 + wm * 64 + mt * 32 + row) * N + n0 + wn * 64 + nt * 32 + col] = acc[q];
        }
    };
    store(acc00, 0, 0); store(acc01, 0, 1); store(acc10, 1, 0); store(acc11, 1, 1);
}

// C = sum over slices, fixed order
__global__ __launch_bounds__(256) void k_gemm_tn_reduce(const float *__restrict__ partial, float *__restrict__ c, int slices,
                                                        long long mn)
{
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= mn) return;
    float4 acc = *reinterpret_cast<const float4 *>(partial + i);
#pragma unroll 8
    for (int z = 1; z < slices; z++) {
        const float4 t = *reinterpret_cast<const float4 *>(partial + (size_t)z * mn + i);
        acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
    }
    *reinterpret_cast<float4 *>(c + i) = acc;
}

// one wavefront per column: lanes stride over the slices, then a fixed-order butterfly
__global__ __launch_bounds__(256) void k_gemm_tn_colsum(const float *__restrict__ partial, float *__restrict__ out, int slices, int M)
{
    const int col = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = (int)(threadIdx.x & 63u);
    if (col >= M) return;
    float acc = 0.f;
    for (int z = lane; z < slices; z += 64) acc += partial[(size_t)z * M + col];
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
    if (lane == 0) out[col] = acc;
}

static void gemm_tn_plan(long long K, int M, int N, int *slices, int *chunks_per_slice)
{
    const int tiles = (M / kTile) * (N / kTile);
    const long long chunks = (K + kKC - 1) / kKC;
    int s = (512 + tiles - 1) / tiles;                 // aim at ~512 workgroups (2 resident per CU, one round)
    s = ((s + 7) / 8) * 8;                             // whole multiples of the 8 XCDs
    if (s > chunks) s = (int)(((chunks + 7) / 8) * 8);
    if (s < 8) s = 8;
    *chunks_per_slice = (int)((chunks + s - 1) / s);
    *slices = s;
}

} // namespace atr

using namespace atr;

extern "C" long long atr_gemm_tn_workspace_floats(long long K, int M, int N)
{
    if (K <= 0 || M <= 0 || N <= 0 || M % kTile || N % kTile) return -1;
    int s, c;
    gemm_tn_plan(K, M, N, &s, &c);
    return (long long)s * M * N + (long long)s * M;
}

extern "C" int atr_gemm_tn(const float *x1, const float *x2, float *c, float *workspace, long long K, int M, int N,
                           const float *row_scale, float *colsum, void *stream)
{
    if (!x1 || !x2 || !c || !workspace || K <= 0 || M <= 0 || N <= 0 || M % kTile || N % kTile) return -1;
    int slices, cps;
    gemm_tn_plan(K, M, N, &slices, &cps);
    const int tiles = (M / kTile) * (N / kTile);
    hipStream_t st = (hipStream_t)stream;
    const long long mn = (long long)M * N;
    float *cs_partial = colsum ? workspace + (size_t)slices * mn : nullptr;
    hipLaunchKernelGGL(k_gemm_tn, dim3((unsigned)(slices * tiles)), dim3(kGemmThreads), 0, st, x1, x2, workspace, K, M, N,
                       slices, cps, row_scale, cs_partial);
    hipLaunchKernelGGL(k_gemm_tn_reduce, dim3((unsigned)((mn / 4 + 255) / 256)), dim3(256), 0, st, workspace, c, slices, mn);
    if (colsum)
        hipLaunchKernelGGL(k_gemm_tn_colsum, dim3((unsigned)((M * 64 + 255) / 256)), dim3(256), 0, st, cs_partial, colsum, slices, M);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
