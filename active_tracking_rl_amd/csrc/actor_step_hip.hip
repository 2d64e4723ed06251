// actor_step_hip.hip — the actor's LSTM step of the rollout as ONE f32-MFMA kernel: both GEMMs of nn.LSTMCell and the
// cell itself (model.py:110,133-145,172,190-209 of the reference; torch.nn.LSTMCell gate order i, f, g, o):
//
//     gates = f W_ih^T + (k h_prev) W_hh^T + (b_ih + b_hh) [+ emb[a_in]]      k = episode mask of the previous step
//     c' = sigm(f) (k c_prev) + sigm(i) tanh(g),   h' = sigm(o) tanh(c')
//
// Before: input GEMM (library, 12.7 us at 4096 rows) + half of a batched hidden GEMM (6 us) + the cell kernel (12.2 us)
// per player and step, with ig / hg (2 x 8 MB) written and re-read in between. Here the K = F + R = 384 contraction runs
// straight into the cell:
//   * v_mfma_f32_32x32x2_f32 (exact f32). Measured on this chip (scratch_exp/mfma_rate.hip): the 32x32x2 form sustains
//     145-155 TFLOP/s with two accumulators per wave, the 16x16x4 form only 97-118 with four (a first version of this
//     kernel on 16x16x4 tiles stalled at 63-70 TFLOP/s).
//   * a wave owns 32 rows x 16 hidden units = two 32x32 output tiles whose 32 columns are [gate i | gate f] and
//     [gate g | gate o] of those 16 units: 1024 wave tiles at 4096 rows = one per SIMD of the chip, 384 MFMAs each, on
//     four accumulators (tile x parity of the MFMA step) so that same-accumulator MFMAs are three others apart.
//   * operands: per 32-wide K block a wave fetches its A rows ([f | h_prev], 32 x 128 B) and its two weight tiles
//     (2 x 32 gate columns x 128 B) with COALESCED loads — 8 lanes x 16 B cover one 128-B line, 8 lines per load
//     instruction — parks them in its own LDS slice (wave-private: no barrier, DS ops of a wave are ordered) and reads
//     them back in the MFMA operand layout (lane = row / column, K slot), conflict-free with a 36-float row stride.
//     Loads run two blocks ahead of the MFMAs. MFMA step t of an 8-wide K chunk consumes component t of the A and B
//     float4s, so the k <-> slot assignment is the same permutation on both sides. Reading the operands in MFMA layout
//     straight from memory (32 B used per 128-B line and instruction) made the L1 tag pipe, not the matrix pipe, set the
//     pace: 24 us per 4096-row call against 11.7 us of MFMA time. The 8 unit slices of a row block run on one XCD
//     (block b runs on XCD b % 8), so the rows are fetched from memory once and re-read out of that XCD's L2; the four
//     waves of a workgroup share their weight tiles through the CU's L1.
//   * lanes j < 16 end up with gates (i, g), lanes j + 16 with (f, o) of the same unit: 16 ds_bpermute exchanges later
//     every lane holds all four gates of 8 (row, unit) pairs and evaluates the cell in registers; h', c' and the
//     activated gates (for the backward pass) are stored once.
// The actor head + categorical draw stay a separate 5 us launch (atr_sample_actions): a row's 128 units are spread over
// 8 workgroups here.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/atr_policy.h"

#ifndef ATR_EXP
#define ATR_EXP 0
#endif

namespace atr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kF = 256, kR = 128, kK = kF + kR;          // feature width, hidden width, contraction length
constexpr int kUnits = 16;                                // hidden units per workgroup (x 4 gates = two 32-column tiles)
constexpr int kRowsPerWg = 128;                           // 4 waves x one 32-row tile
constexpr int kChunks = kK / 8, kChunksIh = kF / 8;       // 48 K-chunks of 8 (4 MFMA steps each); the first 32 are W_ih
constexpr int kBlk = 32, kBlocks = kK / kBlk, kBlocksIh = kF / kBlk;   // K blocks of 32 (4 chunks); the first 8 are W_ih
constexpr int kLs = kBlk + 4;                             // LDS row stride (floats): 16-lane float4 reads hit 64 distinct banks
constexpr int kOpFloats = 32 * kLs;                       // one staged operand (32 rows / columns x 32 k)

// sigmoid / tanh on the hardware exp and reciprocal (v_exp_f32, v_rcp_f32): ~1e-7 relative, far inside the 2e-5 the
// summation order of a 384-term fp32 dot product already costs
__device__ __forceinline__ float sigm(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_fast(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

struct ActorStep {
    const float *f, *h_prev, *c_prev;
    const unsigned char *done;       // nullable
    const float *w_ih, *w_hh, *bias;
    const float *emb;                // nullable [n_in, 4R]
    const long long *act_in;
    float *h_out, *c_out, *acts;     // acts nullable
    int N;
};

__global__ __launch_bounds__(256, 1) void k_actor_step(ActorStep a)
{
    __shared__ __attribute__((aligned(16))) float lds[4][2 * 3 * kOpFloats];   // per wave: 2 buffers x {A, B0, B1}: 27 KB
    const int tid = (int)threadIdx.x, l = tid & 63, wave = tid >> 6;
    const int jj = l & 31, kk = l >> 5;                    // MFMA lane coordinates: row / column index, K slot
    // workgroup -> (row block, unit slice): the 8 slices of a row block on one XCD (see the header)
    const int b = (int)blockIdx.x;
    const int rb = (b >> 6) * 8 + (b & 7), sl = (b >> 3) & 7;
    const int u0 = sl * kUnits;
    const int row0 = rb * kRowsPerWg + wave * 32;
    if (row0 >= a.N) return;                               // no barrier anywhere: a wave may leave alone
    // ATR_EXP == 1 (probe build only, tools/actor_step_timeline.py): s_memtime stamps of this wave
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0;
    if (ATR_EXP == 1) ts0 = __builtin_readcyclecounter();

    // ---- per-row inputs of the cell, first half: the tracker-action indices (their dependent embedding loads follow
    // after the operand prefetch, so nothing here drains the load queue). This lane ends up with rows
    // 8 * (r >> 2) + 4 * kk + (r & 3), r in its half: lanes jj < 16 take r = 0..7, lanes jj >= 16 take r = 8..15.
    const int half = jj >> 4;
    const int u = u0 + (jj & 15);
    const bool has_emb = a.emb != nullptr, has_done = a.done != nullptr;
    const float *embp = has_emb ? a.emb : a.bias;          // without the embedding the bias row stands in (scaled by 0)
    const unsigned char *donep = has_done ? a.done : reinterpret_cast<const unsigned char *>(a.bias);
    const long long *ainp = has_emb ? a.act_in : reinterpret_cast<const long long *>(a.bias);
    int rws[8];
    long long ain[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int r = 8 * half + i;
        rws[i] = min(row0 + 8 * (r >> 2) + 4 * kk + (r & 3), a.N - 1);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) ain[i] = ainp[has_emb ? rws[i] : 0];

    // ---- coalesced staging loads: lane (sub-row rs = l >> 3, k4 = l & 7) fetches float4 X[rs + 8 i][32 blk + 4 k4],
    // i = 0..3, for X = A rows, weight tile 0, weight tile 1
    const int rs = l >> 3, k4 = l & 7;
    const float *pf[4], *ph[4], *pwi[2][4], *pwh[2][4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int ar = min(row0 + rs + 8 * i, a.N - 1);    // tail rows shadow the last one (never stored)
        pf[i] = a.f + (size_t)ar * kF + 4 * k4;
        ph[i] = a.h_prev + (size_t)ar * kR + 4 * k4;
        const int cj = rs + 8 * i;                         // column of the tile: gate 2 T + (cj >> 4), unit u0 + (cj & 15)
#pragma unroll
        for (int T = 0; T < 2; T++) {
            const size_t col = (size_t)(2 * T + (cj >> 4)) * kR + u0 + (cj & 15);
            pwi[T][i] = a.w_ih + col * kF + 4 * k4;
            pwh[T][i] = a.w_hh + col * kR + 4 * k4;
        }
    }
    // (written out per operand with constant indices: with lambdas over a 3 x 4 array the compiler kept the staging
    // registers in memory — promoted to LDS — and every load became load -> wait -> ds_write)
    float4 sa0, sa1, sa2, sa3, sb0, sb1, sb2, sb3, sc0, sc1, sc2, sc3;
#define ATR_LOAD_BLOCK(blk)                                                                                            \
    do {                                                                                                               \
        if ((blk) < kBlocksIh) {                                                                                       \
            const int o_ = kBlk * (blk);                                                                               \
            sa0 = *reinterpret_cast<const float4 *>(pf[0] + o_); sa1 = *reinterpret_cast<const float4 *>(pf[1] + o_);   \
            sa2 = *reinterpret_cast<const float4 *>(pf[2] + o_); sa3 = *reinterpret_cast<const float4 *>(pf[3] + o_);   \
            sb0 = *reinterpret_cast<const float4 *>(pwi[0][0] + o_); sb1 = *reinterpret_cast<const float4 *>(pwi[0][1] + o_); \
            sb2 = *reinterpret_cast<const float4 *>(pwi[0][2] + o_); sb3 = *reinterpret_cast<const float4 *>(pwi[0][3] + o_); \
            sc0 = *reinterpret_cast<const float4 *>(pwi[1][0] + o_); sc1 = *reinterpret_cast<const float4 *>(pwi[1][1] + o_); \
            sc2 = *reinterpret_cast<const float4 *>(pwi[1][2] + o_); sc3 = *reinterpret_cast<const float4 *>(pwi[1][3] + o_); \
        } else {                                                                                                       \
            const int o_ = kBlk * ((blk) - kBlocksIh);                                                                 \
            sa0 = *reinterpret_cast<const float4 *>(ph[0] + o_); sa1 = *reinterpret_cast<const float4 *>(ph[1] + o_);   \
            sa2 = *reinterpret_cast<const float4 *>(ph[2] + o_); sa3 = *reinterpret_cast<const float4 *>(ph[3] + o_);   \
            sb0 = *reinterpret_cast<const float4 *>(pwh[0][0] + o_); sb1 = *reinterpret_cast<const float4 *>(pwh[0][1] + o_); \
            sb2 = *reinterpret_cast<const float4 *>(pwh[0][2] + o_); sb3 = *reinterpret_cast<const float4 *>(pwh[0][3] + o_); \
            sc0 = *reinterpret_cast<const float4 *>(pwh[1][0] + o_); sc1 = *reinterpret_cast<const float4 *>(pwh[1][1] + o_); \
            sc2 = *reinterpret_cast<const float4 *>(pwh[1][2] + o_); sc3 = *reinterpret_cast<const float4 *>(pwh[1][3] + o_); \
        }                                                                                                              \
    } while (0)
    float *mybuf = lds[wave];
    float *stp = mybuf + rs * kLs + 4 * k4;
#define ATR_ST4(dst, v) (*reinterpret_cast<float4 *>(dst) = (v))
#define ATR_STORE_BLOCK(blk)                                                                                           \
    do {                                                                                                               \
        float *d_ = stp + ((blk) & 1) * 3 * kOpFloats;                                                                 \
        ATR_ST4(d_, sa0); ATR_ST4(d_ + 8 * kLs, sa1); ATR_ST4(d_ + 16 * kLs, sa2); ATR_ST4(d_ + 24 * kLs, sa3);        \
        d_ += kOpFloats;                                                                                               \
        ATR_ST4(d_, sb0); ATR_ST4(d_ + 8 * kLs, sb1); ATR_ST4(d_ + 16 * kLs, sb2); ATR_ST4(d_ + 24 * kLs, sb3);        \
        d_ += kOpFloats;                                                                                               \
        ATR_ST4(d_, sc0); ATR_ST4(d_ + 8 * kLs, sc1); ATR_ST4(d_ + 16 * kLs, sc2); ATR_ST4(d_ + 24 * kLs, sc3);        \
    } while (0)
    ATR_LOAD_BLOCK(0);

    // ---- per-row inputs, second half. Raw loads only (results are first touched after the main loop): any arithmetic on
    // a loaded value here makes the compiler wait for it on the spot.
    float bias[4], cp[8], eb[8][4];
    unsigned char dn[8];
#pragma unroll
    for (int g = 0; g < 4; g++) bias[g] = a.bias[g * kR + u];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        cp[i] = a.c_prev[(size_t)rws[i] * kR + u];
        dn[i] = donep[has_done ? rws[i] : 0];
    }
    const int arow = min(row0 + jj, a.N - 1);              // the row this lane supplies as MFMA operand A
    const unsigned char adone = donep[has_done ? arow : 0];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const float *e = embp + (has_emb ? (size_t)ain[i] * (4 * kR) : (size_t)0) + u;
#pragma unroll
        for (int g = 0; g < 4; g++) eb[i][g] = e[g * kR];
    }
    __builtin_amdgcn_sched_barrier(0);
    ATR_STORE_BLOCK(0);
    ATR_LOAD_BLOCK(1);

    // Four accumulators (tile T x parity of the MFMA step): same-accumulator MFMAs are three others apart.
    f32x16 acc[2][2];
#pragma unroll
    for (int T = 0; T < 2; T++)
#pragma unroll
        for (int p2 = 0; p2 < 2; p2++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[T][p2][r] = 0.0f;

    const float akeep = (has_done && adone != 0) ? 0.0f : 1.0f;
    const float *rd = mybuf + jj * kLs + 4 * kk;           // MFMA layout: lane (row / column jj, K slot kk)
    if (ATR_EXP == 1) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); ts1 = __builtin_readcyclecounter(); }
#pragma unroll
    for (int blk = 0; blk < kBlocks; blk++) {
        if (blk + 1 < kBlocks) ATR_STORE_BLOCK(blk + 1);   // block blk + 1 (in registers since the previous iteration) -> LDS
        if (blk + 2 < kBlocks) ATR_LOAD_BLOCK(blk + 2);    // block blk + 2: memory -> registers, under this block's MFMAs
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_sched_barrier(0);
        const float *src = rd + (blk & 1) * 3 * kOpFloats;
#pragma unroll
        for (int cc = 0; cc < kBlk / 8; cc++) {
            float4 av = *reinterpret_cast<const float4 *>(src + 8 * cc);
            const float4 b0 = *reinterpret_cast<const float4 *>(src + kOpFloats + 8 * cc);
            const float4 b1 = *reinterpret_cast<const float4 *>(src + 2 * kOpFloats + 8 * cc);
            if (blk >= kBlocksIh) {            // (k h) W == k (h W): the mask rides on A
                av.x *= akeep; av.y *= akeep; av.z *= akeep; av.w *= akeep;
            }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b0.x, acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b1.x, acc[1][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b0.y, acc[0][1], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b1.y, acc[1][1], 0, 0, 0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b0.z, acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b1.z, acc[1][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b0.w, acc[0][1], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b1.w, acc[1][1], 0, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    f32x16 accs[2];
#pragma unroll
    for (int T = 0; T < 2; T++)
#pragma unroll
        for (int r = 0; r < 16; r++) accs[T][r] = acc[T][0][r] + acc[T][1][r];
    if (ATR_EXP == 1) { asm volatile("" :: "v"(accs[0][0]), "v"(accs[1][15])); ts2 = __builtin_readcyclecounter(); }

    // ---- gather the four gates of each (row, unit): lane jj < 16 holds (i, g), its partner jj + 16 holds (f, o).
    // Exchange k: the low lane sends its row-8+k values and receives the partner's row-k values, and vice versa.
    float gi[8], gf[8], gg[8], go[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const float s0 = half ? accs[0][i] : accs[0][8 + i], s1 = half ? accs[1][i] : accs[1][8 + i];
        const float r0 = __shfl_xor(s0, 16, 64), r1 = __shfl_xor(s1, 16, 64);
        gi[i] = half ? r0 : accs[0][i];
        gf[i] = half ? accs[0][8 + i] : r0;
        gg[i] = half ? r1 : accs[1][i];
        go[i] = half ? accs[1][8 + i] : r1;
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int r = 8 * half + i;
        const int row = row0 + 8 * (r >> 2) + 4 * kk + (r & 3);
        const float es = has_emb ? 1.0f : 0.0f, kp = (has_done && dn[i] != 0) ? 0.0f : 1.0f;
        const float si = sigm(gi[i] + bias[0] + es * eb[i][0]), sf = sigm(gf[i] + bias[1] + es * eb[i][1]);
        const float tg = tanh_fast(gg[i] + bias[2] + es * eb[i][2]), so = sigm(go[i] + bias[3] + es * eb[i][3]);
        const float cn = sf * (kp * cp[i]) + si * tg;
        if (row < a.N) {
            a.c_out[(size_t)row * kR + u] = cn;
            a.h_out[(size_t)row * kR + u] = so * tanh_fast(cn);
            if (a.acts) {
                float *ac = a.acts + (size_t)row * (4 * kR) + u;
                ac[0] = si; ac[kR] = sf; ac[2 * kR] = tg; ac[3 * kR] = so;
            }
        }
    }
    if (ATR_EXP == 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ts3 = __builtin_readcyclecounter();
        if (l == 0 && sl == 0 && a.acts) {      // parked in the gate store of the wave's first row (the probe reads them back)
            unsigned *dbg = reinterpret_cast<unsigned *>(a.acts + (size_t)row0 * (4 * kR));
            dbg[0] = (unsigned)ts0; dbg[1] = (unsigned)ts1; dbg[2] = (unsigned)ts2; dbg[3] = (unsigned)ts3;
        }
    }
}

} // namespace atr

using namespace atr;

extern "C" int atr_actor_step(const float *f, const float *h_prev, const float *c_prev, const unsigned char *done,
                              const float *w_ih, const float *w_hh, const float *bias, const float *emb,
                              const long long *act_in, float *h_out, float *c_out, float *acts, int N, int F, int R,
                              void *stream)
{
    if (!f || !h_prev || !c_prev || !w_ih || !w_hh || !bias || !h_out || !c_out || N <= 0) return -1;
    if (F != kF || R != kR) return -1;                       // the maze policies' LSTMCell(256 -> 128)
    if (emb && !act_in) return -1;
    ActorStep a;
    a.f = f; a.h_prev = h_prev; a.c_prev = c_prev; a.done = done; a.w_ih = w_ih; a.w_hh = w_hh; a.bias = bias;
    a.emb = emb; a.act_in = act_in; a.h_out = h_out; a.c_out = c_out; a.acts = acts; a.N = N;
    static_assert(kR / kUnits == 8, "8 unit slices (one per workgroup of an XCD-local group of 8)");
    const unsigned nrb = (unsigned)((N + kRowsPerWg - 1) / kRowsPerWg);
    const unsigned grid = ((nrb + 7u) / 8u) * 64u;
    hipLaunchKernelGGL(k_actor_step, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
