// gemm_tn_hip.hip — C[M,N] = X1^T X2 for two TALL row-major operands X1 [K,M], X2 [K,N] (K = T*N_envs = 81 920 rows,
// M, N in {128 .. 1024}): the weight-gradient GEMMs of the learner (dW_fc = dpre^T x, dW_ih = dG^T feat,
// dW_hh = h^T dG). The library kernels TunableOp picks run these long-K / small-output shapes at 74-110 TFLOP/s (28 for the
// M = 128 case as a plain mm); this kernel is built for exactly this shape class on the f32 matrix cores (96-101
// TFLOP/s on all of them, 2.5% of an A3C iteration saved; also tighter error than the library's split-K order):
//   * 128 x 128 output tile per workgroup (4 waves, each a 64 x 64 quadrant = 2 x 2 v_mfma_f32_32x32x2_f32 tiles,
//     64 accumulator VGPRs), the K range split over many workgroups (split-K) so that ~512 workgroups exist;
//   * both operands are consumed in their natural row-major layout: a 16- or 32-row chunk of X1 and of X2 (by K: tn_kc) is staged in LDS
//     (unpadded rows: the two k-rows an MFMA operand read touches are served in different LDS passes; padding them
//     apart was measured 3-10 % slower) and an MFMA operand is ONE ds_read_b32 per lane (A[i = m][k] = X1[k][m], B[k][j] = X2[k][n]: lane l reads row k0 + (l>>5),
//     column base + (l & 31)); the next chunk's global loads are in flight while the current one is multiplied;
//   * XCD-aware workgroup order: the output tiles of one K-slice run on the same XCD back to back, so each operand
//     slice is fetched from HBM once and re-read from that XCD's L2 by the other tiles;
//   * split-K partials are reduced in a fixed order by a second kernel (reproducible sums, no atomics).
//   * GROUPED form (atr_gemm_tn_grouped): all weight gradients of one backward pass share K (the T*N rows), so up to 8 of
//     them go out as ONE launch + ONE reduction launch. Alone, the small ones (dW_hh: 4 output tiles) need 128 K-slices to
//     fill the chip — 3 chunks of work per workgroup and 33 MB of partials each; together the problems hold 40-48 tiles, 16
//     slices fill it (768 workgroups = 3 per CU, 20 chunks each), the partials shrink 8x and 14 launches become 2. The
//     reduction writes straight into the destinations the caller names (slices of the flat gradient bucket), the bias
//     gradient into up to two of them (bias_ih and bias_hh receive the same sums).
// fp32 MFMA is an exact fmaf chain, so this is the reference's arithmetic type.
// Where it stands (round 4): 124-127 TFLOP/s = 0.79-0.81 of the 157 TFLOP/s f32-MFMA peak on the learner's group at 4096 envs
// (1014-1035 us; round 3: 1184 us). What moved it: the operand prefetch kept above the MFMAs (107 -> 114-118), operands DMA'd
// straight into LDS (-> 124-127). Measured on the way (tools/gemm_tn_timeline.py, SQ counters): SQ_VALU_MFMA_BUSY_CYCLES =
// 0.765 of the kernel's cycles before the DMA path; the shader clock holds 2.4 GHz in steady state (s_memtime against the
// 100 MHz wall clock; the first launches after idle run at 1.9-2.0); equal workgroups take 287-390 us and the CUs finish
// 0.95 full; operands resident in L2 / MALL gain 3 %; LDS bank conflicts 0. Tried and dropped: cutting the last slices short to
// fill the ragged end (1-3 % slower: more partial tiles), wave-private staging with no barrier at all (15 % slower: twice the
// staging traffic costs more than the barriers), the LDS writes of the next chunk moved into the MFMA loop (no change).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/atr_policy.h"

namespace atr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef ATR_TN_DIRECT
#define ATR_TN_DIRECT 1    // 0: staging through registers only (A/B builds)
#endif
#ifndef ATR_TN_PROBE
#define ATR_TN_PROBE 0     // 1: probe build only (tools/gemm_tn_timeline.py): per-workgroup start / end stamps
#endif
constexpr int kTile = 128, kLd = 128;   // tile side, LDS row length (floats)
constexpr int kGemmThreads = 256;
// Rows per K chunk — a template parameter of the kernel: 32 (64 KB of LDS, 2 workgroups per CU) or 16 (32 KB, 4 per CU).
// The learner's group, 16 / 24 / 32 rows: 179 / 180 / 198 us at 512 envs, 309 / 309 / 334 at 1024, 571 / 570 / 575 at 2048,
// 1078 / 1083 / 1094 at 4096 (stand-alone); next to the other replica's rollout (pipelined schedule, 4096 envs) the 32-row
// form leaves it more room: 15.58-15.60 M env steps/s against 15.43-15.46. So: 16 rows up to kSmallK rows of K, 32 above.
constexpr long long kSmallK = 20 * 1024 + 1;
constexpr int tn_wg_per_cu(int kc) { return kc == 32 ? 2 : kc == 24 ? 3 : 4; }
// Co-run mode (atr_gemm_tn_set_corun; the pipelined schedule's learner, train.PipelinedIteration): the launch is padded with
// dynamic LDS until ONE workgroup fits a CU — 4 waves, one per SIMD, 32 KB of LDS — and the kernel takes ~1.4x as long, but
// the other replica's rollout on the second stream, a chain of short latency-bound kernels, keeps moving next to it instead
// of queueing behind co-resident dW workgroups that run for hundreds of microseconds each: the pipelined iteration at 4096
// envs 15.75 -> 16.6-16.75 M env steps/s, 11.3 -> 11.7 at 1024, 13.95 -> 14.5 at 2048 (two or three workgroups per CU: no
// gain at all; the same cap on k_stem_bwd: +1 % alone, a loss together with this one).
static thread_local int g_tn_corun = 0;     // per calling thread: another thread's eager launches keep their own mode
static int tn_kc(long long K)
{
    static const int forced = getenv("ATR_GEMM_TN_KC") ? atoi(getenv("ATR_GEMM_TN_KC")) : 0;   // (tuning experiments)
    if (forced == 16 || forced == 32) return forced;
    if (g_tn_corun) return 16;
    return K < kSmallK ? 16 : 32;
}
static int tn_wg_per_cu_now(int kc) { return g_tn_corun ? 1 : tn_wg_per_cu(kc); }
constexpr unsigned kCorunLdsPad = 50000;     // 32 KB static (16-row chunks) + this > half of a CU's 160 KB

template <int kKC> struct GemmLds { float a[2][kKC][kLd]; float b[2][kKC][kLd]; };

constexpr int kMaxProblems = 8;

struct TnProblem {
    const float *x1, *x2;          // [K, M], [K, N] row-major
    float *partial;                // [slices, M, N]
    float *cs_partial;             // [slices, M] or null
    const float *row_scale;        // nullable: row k of X1 is multiplied by row_scale[k - rs_shift] (1 for k < rs_shift)
    float *c;                      // [M, N] destination of the reduction
    float *cs_out0, *cs_out1;      // [M] destinations of the column sums (nullable)
    long long rs_shift;
    long long ld1, ld2;            // row strides of x1 / x2 (floats)
    int M, N;
    int tile_begin;                // first tile of this problem in the group's tile order
    int red_begin;                 // first block of this problem in the reduction launch
};

struct TnGroup {
    TnProblem p[kMaxProblems];
    long long K;
    int count, slices, chunks_per_slice, tiles_total;
    int red_blocks;                // matrix blocks of the reduction launch; column-sum blocks follow
};

#if ATR_TN_PROBE
__device__ unsigned long long g_tn_stamps[8192 * 4];
#endif

// kMixed = false: the host has checked that EVERY workgroup of the launch takes the direct path (no row factors anywhere in the
// group, K a whole number of chunks) and the register-staged path is compiled out.
template <int kKC, bool kMixed>
__global__ __launch_bounds__(kGemmThreads, tn_wg_per_cu(kKC)) void k_gemm_tn(const TnGroup g)
{
    __shared__ __attribute__((aligned(16))) GemmLds<kKC> s;
    const int tid = (int)threadIdx.x, l = tid & 63, wave = tid >> 6;
#if ATR_TN_PROBE
    const unsigned long long st_rt = wall_clock64(), st_cy = __builtin_readcyclecounter();
#endif
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware order: workgroup i runs on XCD i % 8; give each XCD whole K-slices (all their tiles, of every problem, back
    // to back): an operand slice is fetched from HBM once and re-read from that XCD's L2 by the other tiles — and by the other
    // problems that share it (dW_ih and dW_hh of a player both contract dG)
    const int i = (int)blockIdx.x, xcd = i & 7, in_xcd = i >> 3;
    const int slice = xcd + 8 * (in_xcd / g.tiles_total), tg = in_xcd % g.tiles_total;
    if (slice >= g.slices) return;
    int q = 0;
#pragma unroll
    for (int j = 1; j < kMaxProblems; j++)
        if (j < g.count && tg >= g.p[j].tile_begin) q = j;
    const float *__restrict__ x1 = g.p[q].x1, *__restrict__ x2 = g.p[q].x2;
    float *__restrict__ partial = g.p[q].partial, *__restrict__ colsum_partial = g.p[q].cs_partial;
    const float *__restrict__ row_scale = g.p[q].row_scale;
    const long long rs_shift = g.p[q].rs_shift, K = g.K;
    const int M = g.p[q].M, N = g.p[q].N, tile = tg - g.p[q].tile_begin, chunks_per_slice = g.chunks_per_slice;
    const int tiles_n = N / kTile;
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
    const long long ldm = g.p[q].ld1 / 4, ldn = g.p[q].ld2 / 4;        // row strides in float4 units
    float4 ra0, ra1, rb0, rb1, ra2, ra3, rb2, rb3;
    float rs0 = 1.f, rs1 = 1.f, rs2 = 1.f, rs3 = 1.f;
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
        if (kKC >= 24 && kc_ < k_end) { ra2 = g1[kc_ * ldm]; rb2 = g2[kc_ * ldn]; }         \
        if (kKC == 32 && kd_ < k_end) { ra3 = g1[kd_ * ldm]; rb3 = g2[kd_ * ldn]; }         \
        if (row_scale) {      /* the factors ride along with the chunk; applied on the way into LDS (GEMM_TN_STAGE) */     \
            rs0 = 1.f; rs1 = 1.f; rs2 = 1.f; rs3 = 1.f;        /* (rows past k_end hold zeros already) */               \
            if (ka_ >= rs_shift && ka_ < k_end) rs0 = row_scale[ka_ - rs_shift];                                       \
            if (kb_ >= rs_shift && kb_ < k_end) rs1 = row_scale[kb_ - rs_shift];                                       \
            if (kKC >= 24 && kc_ >= rs_shift && kc_ < k_end) rs2 = row_scale[kc_ - rs_shift];                         \
            if (kKC == 32 && kd_ >= rs_shift && kd_ < k_end) rs3 = row_scale[kd_ - rs_shift];                         \
        }                                                                      \
    }
#define GEMM_TN_STAGE(buf_)                                                    \
    {                                                                          \
        if (row_scale) {                                                       \
            ra0.x *= rs0; ra0.y *= rs0; ra0.z *= rs0; ra0.w *= rs0; ra1.x *= rs1; ra1.y *= rs1; ra1.z *= rs1; ra1.w *= rs1; \
            ra2.x *= rs2; ra2.y *= rs2; ra2.z *= rs2; ra2.w *= rs2; ra3.x *= rs3; ra3.y *= rs3; ra3.z *= rs3; ra3.w *= rs3; \
        }                                                                      \
        *reinterpret_cast<float4 *>(&s.a[buf_][r][c4 * 4]) = ra0;              \
        *reinterpret_cast<float4 *>(&s.a[buf_][r + 8][c4 * 4]) = ra1;          \
        *reinterpret_cast<float4 *>(&s.b[buf_][r][c4 * 4]) = rb0;              \
        *reinterpret_cast<float4 *>(&s.b[buf_][r + 8][c4 * 4]) = rb1;          \
        if (kKC >= 24) {                                                       \
        *reinterpret_cast<float4 *>(&s.a[buf_][(r + 16) % kKC][c4 * 4]) = ra2;         \
        *reinterpret_cast<float4 *>(&s.b[buf_][(r + 16) % kKC][c4 * 4]) = rb2;         \
        }                                                                      \
        if (kKC == 32) {                                                       \
        *reinterpret_cast<float4 *>(&s.a[buf_][(r + 24) % kKC][c4 * 4]) = ra3;         \
        *reinterpret_cast<float4 *>(&s.b[buf_][(r + 24) % kKC][c4 * 4]) = rb3;         \
        }                                                                      \
    }
    f32x16 acc00, acc01, acc10, acc11;
#pragma unroll
    for (int q = 0; q < 16; q++) { acc00[q] = 0.f; acc01[q] = 0.f; acc10[q] = 0.f; acc11[q] = 0.f; }
    const int kr = l >> 5, col = l & 31;
    auto compute = [&](int buf) {
        const float *pa = &s.a[buf][kr][wm * 64 + col], *pb = &s.b[buf][kr][wn * 64 + col];
        float a0 = pa[0], a1 = pa[32], b0 = pb[0], b1 = pb[32];
#pragma unroll
        for (int kp = 0; kp < kKC / 2; kp++) {
            float na0 = 0.f, na1 = 0.f, nb0 = 0.f, nb1 = 0.f;
            if (kp + 1 < kKC / 2) {                      // operands of the next k-pair are read under this one's MFMAs
                na0 = pa[(kp + 1) * 2 * kLd]; na1 = pa[(kp + 1) * 2 * kLd + 32];
                nb0 = pb[(kp + 1) * 2 * kLd]; nb1 = pb[(kp + 1) * 2 * kLd + 32];
            }
            // keep these reads ABOVE the MFMAs: left alone, the scheduler sinks them below and waits for them at once
            // (ds_read, s_waitcnt lgkmcnt(0), 4 MFMAs per k-pair: the matrix pipe idles for an LDS latency every 256 cycles —
            // 107 -> 114 TFLOP/s on the learner's group at 4096 envs)
            __builtin_amdgcn_sched_barrier(0);
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
    };
    // Whole chunks and no row factors (the learner's own groups at every BASELINE size): the operands go from global memory
    // STRAIGHT into LDS (global_load_lds_dwordx4: lane l of a wave lands at base + 16 l, i.e. a wave fills two 128-float rows
    // per instruction) — no staging registers, no ds_write, nothing for the wave to do between issuing the loads of chunk
    // c + 1 (into the buffer everybody left at the last barrier) and the barrier that ends chunk c.
    // (a mixed launch with 16-row chunks keeps to the register-staged path: both paths together do not fit its 128 VGPRs)
    const bool direct = !kMixed || (kKC == 32 && ATR_TN_DIRECT && row_scale == nullptr && k_end > k_begin && (k_end - k_begin) % kKC == 0);
    if (direct) {
        auto dma = [&](long long k0, int buf) {
#pragma unroll
            for (int j = 0; j < kKC / 8; j++) {
                const long long kk = k0 + r + 8 * j;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g1 + kk * ldm),
                                                 (__attribute__((address_space(3))) void *)(&s.a[buf][2 * wave + 8 * j][0]), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g2 + kk * ldn),
                                                 (__attribute__((address_space(3))) void *)(&s.b[buf][2 * wave + 8 * j][0]), 16, 0, 0);
            }
        };
        if (k_begin < k_end) dma(k_begin, 0);            // (a slice past the end of K — the planner's last ones when the chunk
        __builtin_amdgcn_s_waitcnt(0x0F70);              // count does not divide — loads nothing and stores zeros); vmcnt(0)
        __syncthreads();
        int buf = 0;
        for (long long k0 = k_begin; k0 < k_end; k0 += kKC, buf ^= 1) {
            if (k0 + kKC < k_end) dma(k0 + kKC, buf ^ 1);
            compute(buf);
            __builtin_amdgcn_s_waitcnt(0x0F70);          // the next chunk has landed (issued a whole chunk of MFMAs ago)
            __syncthreads();
        }
    } else if (kMixed) {
        GEMM_TN_FETCH(k_begin);
        GEMM_TN_STAGE(0);
        __syncthreads();
        int buf = 0;
        // (a second register set, i.e. two chunks of prefetch distance, was measured: no faster — with several workgroups per
        // CU the other one's MFMAs already cover the load latency)
        for (long long k0 = k_begin; k0 < k_end; k0 += kKC, buf ^= 1) {
            const bool more = k0 + kKC < k_end;
            if (more) GEMM_TN_FETCH(k0 + kKC);           // next chunk's loads fly under this chunk's MFMAs
            compute(buf);
            if (more) GEMM_TN_STAGE(buf ^ 1);            // the other buffer was last read one iteration ago
            __syncthreads();
        }
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
            // (nontemporal: the partials are read once, by the reduction launch; kept out of L2 they are also not a pile of
            // dirty lines that every kernel boundary of a concurrent stream has to write back — the pipelined schedule)
            __builtin_nontemporal_store(acc[q], &out[(size_t)(m0 + wm * 64 + mt * 32 + row) * N + n0 + wn * 64 + nt * 32 + col]);
        }
    };
    store(acc00, 0, 0); store(acc01, 0, 1); store(acc10, 1, 0); store(acc11, 1, 1);
#if ATR_TN_PROBE
    if (tid == 0 && blockIdx.x < 8192) {
        unsigned long long *o = &g_tn_stamps[(size_t)blockIdx.x * 4];
        o[0] = st_rt; o[1] = wall_clock64(); o[2] = __builtin_readcyclecounter() - st_cy;
        o[3] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);
    }
#endif
}

// C = sum over slices, fixed order — all problems of a group in one launch: blocks [red_begin, ...) of a problem cover its
// M*N outputs (1024 per block), the blocks after red_blocks its column sums (one wavefront per column: lanes stride over the
// slices, then a fixed-order butterfly; 4 columns per block)
__global__ __launch_bounds__(256) void k_gemm_tn_reduce(const TnGroup g)
{
    const int b = (int)blockIdx.x;
    if (b < g.red_blocks) {
        int q = 0;
#pragma unroll
        for (int j = 1; j < kMaxProblems; j++)
            if (j < g.count && b >= g.p[j].red_begin) q = j;
        const long long mn = (long long)g.p[q].M * g.p[q].N;
        const long long i = ((long long)(b - g.p[q].red_begin) * 256 + threadIdx.x) * 4;
        if (i >= mn) return;
        const float *__restrict__ partial = g.p[q].partial;
        float4 acc = *reinterpret_cast<const float4 *>(partial + i);
#pragma unroll 8
        for (int z = 1; z < g.slices; z++) {
            const float4 t = *reinterpret_cast<const float4 *>(partial + (size_t)z * mn + i);
            acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
        }
        *reinterpret_cast<float4 *>(g.p[q].c + i) = acc;
        return;
    }
    int cb = b - g.red_blocks;                       // column-sum blocks: problem by problem, M / 4 blocks each
    for (int q = 0; q < g.count; q++) {
        if (!g.p[q].cs_partial) continue;
        const int nb = g.p[q].M / 4;
        if (cb >= nb) { cb -= nb; continue; }
        const int col = cb * 4 + (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63u), M = g.p[q].M;
        const float *__restrict__ partial = g.p[q].cs_partial;
        float acc = 0.f;
        for (int z = lane; z < g.slices; z += 64) acc += partial[(size_t)z * M + col];
        for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
        if (lane == 0) {
            if (g.p[q].cs_out0) g.p[q].cs_out0[col] = acc;
            if (g.p[q].cs_out1) g.p[q].cs_out1[col] = acc;
        }
        return;
    }
}

// slices (a multiple of the 8 XCDs) for `tiles` output tiles in all: the estimate is rounds of 256-CU occupancy x rows per
// workgroup, plus a fixed cost per round (prologue / epilogue of a workgroup ~ 64 rows of work)
struct TnPlan { int slices, chunks_per_slice; };

static TnPlan gemm_tn_plan(long long K, int tiles)
{
    const int kKC = tn_kc(K), kWgPerCu = tn_wg_per_cu_now(kKC);
    // kWgPerCu workgroups fit a CU (LDS) and run best together (one's MFMAs cover the others' load latency): count
    // rounds of 256 * kWgPerCu co-resident workgroups at that efficiency, a tail of <= 256 as a round of lone workgroups at solo
    // efficiency; every workgroup pays a fixed prologue / epilogue worth ~64 rows
    const long long chunks = (K + kKC - 1) / kKC;
    const double eff_pair = 0.85, eff_solo = 0.6;
    int best = 8;
    double best_cost = 1e30;
    static const int forced = getenv("ATR_GEMM_TN_SLICES") ? atoi(getenv("ATR_GEMM_TN_SLICES")) : 0;   // (tuning experiments)
    for (int s = 8; s <= 512; s += 8) {
        if (s > chunks && s > 8) break;
        const long long cps = (chunks + s - 1) / s, wgs = (long long)tiles * s;
        const double rows = (double)cps * kKC + 64.0;
        const long long round = 256 * kWgPerCu;
        const long long full = wgs / round, rem = wgs - full * round;
        double cost = (double)full * kWgPerCu * rows / eff_pair;
        if (rem > 256) cost += (double)((rem + 255) / 256) * rows / eff_pair;
        else if (rem > 0) cost += rows / eff_solo;
        if (cost < best_cost - 1e-9) { best_cost = cost; best = s; }
    }
    if (forced >= 8 && forced % 8 == 0 && forced <= chunks) best = forced;
    TnPlan p;
    p.chunks_per_slice = (int)((chunks + best - 1) / best);
    p.slices = best;
    // (Equal workgroups do not take equal time — 300-510 us for the same 80 chunks at 4096 envs, tools/gemm_tn_timeline.py:
    // power management, the neighbour on the CU — and every CU gets the same number of them, so the CUs finish ~7 % apart.
    // Cutting the last 8 slices into 2-4x as many short ones to fill that ragged end was measured: 1-3 % SLOWER, the extra
    // partial tiles cost more than the tail returns.)
    return p;
}

static int group_tiles(const atr_gemm_tn_problem *pr, int count)
{
    int tiles = 0;
    for (int q = 0; q < count; q++) {
        if (!pr[q].x1 || !pr[q].x2 || !pr[q].c || pr[q].M <= 0 || pr[q].N <= 0 || pr[q].M % kTile || pr[q].N % kTile) return -1;
        if ((pr[q].ld1 && (pr[q].ld1 < pr[q].M || pr[q].ld1 % 4)) || (pr[q].ld2 && (pr[q].ld2 < pr[q].N || pr[q].ld2 % 4))) return -1;
        tiles += (pr[q].M / kTile) * (pr[q].N / kTile);
    }
    return tiles;
}

} // namespace atr

using namespace atr;

#if ATR_TN_PROBE
extern "C" int atr_tn_stamps(unsigned long long *host, int wgs)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_tn_stamps), (size_t)wgs * 4 * sizeof(unsigned long long));
}
#endif

extern "C" long long atr_gemm_tn_grouped_workspace_floats(const atr_gemm_tn_problem *problems, int count, long long K)
{
    if (!problems || count < 1 || count > kMaxProblems || K <= 0) return -1;
    const int tiles = group_tiles(problems, count);
    if (tiles < 0) return -1;
    const int s = gemm_tn_plan(K, tiles).slices;
    long long total = 0;
    for (int q = 0; q < count; q++) total += (long long)s * problems[q].M * problems[q].N + (long long)s * problems[q].M;
    return total;
}

extern "C" int atr_gemm_tn_grouped(const atr_gemm_tn_problem *problems, int count, long long K, float *workspace, void *stream)
{
    if (!problems || !workspace || count < 1 || count > kMaxProblems || K <= 0) return -1;
    const int tiles = group_tiles(problems, count);
    if (tiles < 0) return -1;
    TnGroup g;
    const TnPlan plan = gemm_tn_plan(K, tiles);
    g.slices = plan.slices; g.chunks_per_slice = plan.chunks_per_slice;
    g.K = K; g.count = count; g.tiles_total = tiles;
    float *ws = workspace;
    int tile_begin = 0, red_begin = 0, cs_blocks = 0;
    for (int q = 0; q < kMaxProblems; q++) {
        TnProblem &d = g.p[q];
        if (q >= count) { d = g.p[0]; d.tile_begin = 1 << 30; d.red_begin = 1 << 30; continue; }
        const atr_gemm_tn_problem &p = problems[q];
        const long long mn = (long long)p.M * p.N;
        d.x1 = p.x1; d.x2 = p.x2; d.c = p.c; d.row_scale = p.row_scale; d.rs_shift = p.row_scale_shift; d.M = p.M; d.N = p.N;
        d.ld1 = p.ld1 ? p.ld1 : p.M; d.ld2 = p.ld2 ? p.ld2 : p.N;
        d.cs_out0 = p.colsum0; d.cs_out1 = p.colsum1;
        d.partial = ws; ws += (size_t)g.slices * mn;
        const bool cs = p.colsum0 || p.colsum1;
        d.cs_partial = cs ? ws : nullptr;
        if (cs) { ws += (size_t)g.slices * p.M; cs_blocks += p.M / 4; }
        d.tile_begin = tile_begin; tile_begin += (p.M / kTile) * (p.N / kTile);
        d.red_begin = red_begin; red_begin += (int)((mn / 4 + 255) / 256);
    }
    g.red_blocks = red_begin;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)(g.slices * tiles)), block(kGemmThreads);
    const int kc = tn_kc(K);
    static const unsigned forced_pad = getenv("ATR_GEMM_TN_LDS_PAD") ? (unsigned)atoi(getenv("ATR_GEMM_TN_LDS_PAD")) : 0u;   // (occupancy experiments)
    const unsigned lds_pad = forced_pad ? forced_pad : (g_tn_corun && kc == 16 ? kCorunLdsPad : 0u);
    static bool attr_set = false;
    if (lds_pad && !attr_set) {              // (static + dynamic LDS beyond 64 KB needs the opt-in)
        const int lim = 96 * 1024;
        if (hipFuncSetAttribute((const void *)k_gemm_tn<16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lim) != hipSuccess ||
            hipFuncSetAttribute((const void *)k_gemm_tn<16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lim) != hipSuccess ||
            hipFuncSetAttribute((const void *)k_gemm_tn<32, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lim) != hipSuccess ||
            hipFuncSetAttribute((const void *)k_gemm_tn<32, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lim) != hipSuccess)
            return -2;
        attr_set = true;
    }
    bool all_direct = ATR_TN_DIRECT && K % kc == 0;
    for (int q = 0; q < count; q++) all_direct = all_direct && problems[q].row_scale == nullptr;
    if (kc == 16) {
        if (all_direct) hipLaunchKernelGGL((k_gemm_tn<16, false>), grid, block, lds_pad, st, g);
        else hipLaunchKernelGGL((k_gemm_tn<16, true>), grid, block, lds_pad, st, g);
    } else {
        if (all_direct) hipLaunchKernelGGL((k_gemm_tn<32, false>), grid, block, lds_pad, st, g);
        else hipLaunchKernelGGL((k_gemm_tn<32, true>), grid, block, lds_pad, st, g);
    }
    hipLaunchKernelGGL(k_gemm_tn_reduce, dim3((unsigned)(red_begin + cs_blocks)), dim3(256), 0, st, g);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int atr_gemm_tn_set_corun(int on)
{
    const int was = g_tn_corun;
    g_tn_corun = on ? 1 : 0;
    return was;
}

extern "C" long long atr_gemm_tn_workspace_floats(long long K, int M, int N)
{
    atr_gemm_tn_problem p = {};
    float dummy = 0.f;
    p.x1 = p.x2 = &dummy; p.c = &dummy; p.M = M; p.N = N; p.colsum0 = &dummy;
    return atr_gemm_tn_grouped_workspace_floats(&p, 1, K);
}

extern "C" int atr_gemm_tn(const float *x1, const float *x2, float *c, float *workspace, long long K, int M, int N,
                           const float *row_scale, float *colsum, void *stream)
{
    atr_gemm_tn_problem p = {};
    p.x1 = x1; p.x2 = x2; p.c = c; p.M = M; p.N = N; p.row_scale = row_scale; p.row_scale_shift = 0; p.colsum0 = colsum;
    return atr_gemm_tn_grouped(&p, 1, K, workspace, stream);
}
