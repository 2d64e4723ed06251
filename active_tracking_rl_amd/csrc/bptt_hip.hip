// bptt_hip.hip — back-propagation through time of the two players' LSTMCells over a whole rollout as ONE launch.
//
// The learner's recurrence backward was 2 launches per step and player pair (atr_lstm_cell_backward, then a batched GEMM
// dG_t W_hh feeding step t - 1): 40 dependent launches of 5-12 us per iteration, with dG_t written by one kernel and re-read by
// the next. But the recurrence is independent PER ROW (env): dh_{t-1}[n] = dG_t[n] W_hh. So one workgroup takes 16 rows of
// one player through all T steps:
//   * W_hh [4R, R] = [512, 128] stays in REGISTERS for the whole launch as MFMA B operands: wave w of the 8 owns output
//     units [16 w, 16 w + 16) = 128 B-operand VGPRs of v_mfma_f32_16x16x4_f32 (exact f32), loaded once;
//   * the accumulator layout of that MFMA (lane = unit column, 4 rows) IS the ownership of the element-wise cell backward:
//     the gradient arriving through the hidden GEMM (dh_{t-1} contribution) and the cell-state carry never leave the
//     registers of the thread that needs them at step t - 1;
//   * per step: every thread evaluates the cell backward of its 4 (row, unit) pairs (the expressions of k_lstm_cell_bwd),
//     stores the four gate gradients (the learner's weight-gradient GEMMs read dG afterwards) and parks them in a
//     double-buffered LDS tile [16 rows x 512]; ONE barrier; then 128 MFMAs per wave contract the tile with the wave's
//     weight slice (A operands by conflict-free ds_read_b128; the k <-> MFMA-slot assignment is the same permutation on both
//     operands); the next step's activations are fetched from memory under those MFMAs.
// Bounds: 2 x 16 x 512 x 128 flop per step and workgroup = 3.4 us on a CU's four SIMDs (the floor at small batches: 68 us per
// 20-step rollout however few envs), and the stored activations + dG stream (~11 KB per row and step) at large batches.
// Reference semantics: torch.nn.LSTMCell's backward under the truncated-BPTT rollout of player_util.py:104-161 (the episode
// masks k_t cut the recurrence where an env finished: train.py:73-79).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/atr_policy.h"
#include "atr_cell.h"

namespace atr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kBR = 128;                    // hidden units
constexpr int kBK = 4 * kBR;                // contraction length: the four gates
constexpr int kBRows = 16;                  // rows per workgroup (one MFMA M tile)
constexpr int kBLd = kBK + 4;               // LDS row stride (floats): 16-B reads of 8 lanes hit 32 distinct banks

struct Bptt {
    const float *dh[2];        // per player: dL/dh_t from the heads [T, N, R] (nullable: zero)
    const float *keep;         // [T, N] float episode masks
    const float *acts;         // + p * acts_ps + (t * N + n) * 4R : activated gates (i, f, g, o) — or, PRE, what the rollout's
    long long acts_ps;         //   gate GEMM wrote: the pre-activations without bias (activated here: same expressions)
    // PRE only: per player b_ih + b_hh [4R]; the tracker-action embedding projected through W_ih [n_act, 4R] added to player
    // emb_player's pre-activations, row = the tracker's action of that step (int64, act + t * act_ts + n)
    const float *bias[2];
    const float *emb;
    const long long *act;
    long long act_ts;
    int emb_player, n_act;
    const float *c_all;        // + p * c_ps + (t * N + n) * R : cell states, slot t = before step t, slot t + 1 = after
    long long c_ps;
    const float *whh[2];       // per player: weight_hh [4R, R] (nn.LSTMCell layout)
    float *dg;                 // + p * dg_ps + (t * N + n) * 4R : pre-activation gate gradients (out)
    long long dg_ps;
    float *dh0, *dc0;          // [P, N, R]: gradient into the rollout's initial hidden / cell state (out)
    // PRE + emb only (nullable): per row tile of player emb_player, the column sums of dG over the tile's rows BY THE TRACKER'S
    // ACTION of the row — [tiles][4][4R] (n_act == 4). S = their sum over the tiles is all the embedding needs of the backward
    // pass: d fc_action_tracker = S W_ih, and the target's dW_ih = dG^T (f + E[a]) = dG^T f + S^T E (atr_embed_fold) — the
    // learner then neither materialises f + E[a] nor gathers dL/df by action (two passes over [T N, F] each)
    float *act_sums;
    int P, T, N;
};

template <bool PRE>
__global__ __launch_bounds__(512, 1) void k_lstm_bptt(Bptt a)
{
    extern __shared__ __attribute__((aligned(16))) float tileA[];      // [2][kBRows][kBLd] (+ PRE: the embedding table, row actions)
    const int tid = (int)threadIdx.x, l = tid & 63, w = tid >> 6;
    const int col = l & 15, q = l >> 4;
    const int tiles = (a.N + kBRows - 1) / kBRows;
    const int p = (int)blockIdx.x / tiles, rt = (int)blockIdx.x - p * tiles;
    const int row0 = rt * kBRows;
    const int u = 16 * w + col;                                         // this lane's hidden unit
    // ---- this wave's slice of W_hh as MFMA B operands: step s = 4 g + t, slot q  <->  k = 16 g + 4 q + t
    const float *W = a.whh[p];
    float B[128];
#pragma unroll
    for (int g = 0; g < 32; g++)
#pragma unroll
        for (int t = 0; t < 4; t++) B[4 * g + t] = W[(size_t)(16 * g + 4 * q + t) * kBR + u];
    // ---- the 4 (row, unit) pairs of this thread: rows row0 + 4 q + i — the rows of its MFMA accumulator
    int rows[4];
    bool ok[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { rows[i] = row0 + 4 * q + i; ok[i] = rows[i] < a.N; rows[i] = min(rows[i], a.N - 1); }
    const float *acts = a.acts + (size_t)p * a.acts_ps;
    const float *call = a.c_all + (size_t)p * a.c_ps;
    const float *dhp = a.dh[p];
    float *dg = a.dg + (size_t)p * a.dg_ps;
    // PRE: the rollout kept the gate GEMM's output instead of the activated gates (k_act_step then writes 4R floats per row and
    // step less); the activations are recomputed here exactly as k_act_step computed them: ((pre + bias) + emb[a_tracker]),
    // sigmoidf_ / tanhf_ of atr_cell.h
    float bi = 0.f, bf = 0.f, bg = 0.f, bo = 0.f;
    float *embL = tileA + (size_t)2 * kBRows * kBLd;
    const bool with_emb = PRE && a.emb != nullptr && p == a.emb_player;
    if (PRE) {
        const float *b = a.bias[p];
        bi = b[u]; bf = b[kBR + u]; bg = b[2 * kBR + u]; bo = b[3 * kBR + u];
        if (with_emb)
            for (int i = tid; i < a.n_act * kBK; i += 512) embL[i] = a.emb[i];
        __syncthreads();
    }
    int ai[4] = {0, 0, 0, 0};
    // column sums of dG by the row's tracker action: thread tid owns COLUMN tid of the [16 x 512] tile every step parks in LDS
    const bool sums = PRE && with_emb && a.act_sums != nullptr;
    float *actL = embL + (size_t)kMaxActions * kBK;       // [2][16][4]: one-hot of the tile rows' actions (zeros: no such row)
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};          // dG_{t+1} W_hh for this thread's pairs (gradient arriving through h_t)
    float dcc[4] = {0.f, 0.f, 0.f, 0.f};       // dc_{t+1} f_{t+1}
    // staged inputs of one step (fetched one step ahead, under the MFMAs)
    float gi[4], gf[4], gg[4], go[4], cv[4], cpv[4], dhh[4], ko[4], ki[4];
#define BPTT_FETCH(t_)                                                                                                 \
    do {                                                                                                               \
        const int tt_ = (t_);                                                                                          \
        _Pragma("unroll") for (int i = 0; i < 4; i++) {                                                                \
            const size_t r_ = (size_t)tt_ * a.N + rows[i];                                                             \
            const float *ac_ = acts + r_ * kBK + u;                                                                    \
            gi[i] = ac_[0]; gf[i] = ac_[kBR]; gg[i] = ac_[2 * kBR]; go[i] = ac_[3 * kBR];                              \
            cv[i] = call[(r_ + a.N) * kBR + u];                  /* c after step t: slot t + 1 */                      \
            cpv[i] = call[r_ * kBR + u];                         /* c before step t */                                 \
            dhh[i] = dhp ? dhp[r_ * kBR + u] : 0.0f;                                                                   \
            if (with_emb) ai[i] = (int)a.act[(size_t)tt_ * a.act_ts + rows[i]];                                        \
            ko[i] = a.keep[r_];                                                                                        \
            ki[i] = tt_ > 0 ? a.keep[r_ - a.N] : 1.0f;                                                                 \
        }                                                                                                              \
    } while (0)
    BPTT_FETCH(a.T - 1);
#pragma unroll 1
    for (int t = a.T - 1; t >= 0; t--) {
        float *At = tileA + (size_t)(t & 1) * kBRows * kBLd;
        const bool has_next = t < a.T - 1;
        // ---- cell backward of this thread's 4 pairs (k_lstm_cell_bwd's expressions)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (PRE) {
                float xi = gi[i] + bi, xf = gf[i] + bf, xg = gg[i] + bg, xo = go[i] + bo;
                if (with_emb) {
                    const float *e_ = embL + ai[i] * kBK + u;
                    xi = e_[0] + xi; xf = e_[kBR] + xf; xg = e_[2 * kBR] + xg; xo = e_[3 * kBR] + xo;
                }
                gi[i] = sigmoidf_(xi); gf[i] = sigmoidf_(xf); gg[i] = tanhf_(xg); go[i] = sigmoidf_(xo);
            }
            float dh = dhh[i], dcn = 0.0f;
            if (has_next) { dh = fmaf(ko[i], acc[i], dh); dcn = ko[i] * dcc[i]; }
            const float tc = tanhf_(cv[i]);
            const float dc = dcn + dh * go[i] * (1.0f - tc * tc);
            const float d_o = dh * tc * go[i] * (1.0f - go[i]);
            const float d_i = dc * gg[i] * gi[i] * (1.0f - gi[i]);
            const float d_g = dc * gi[i] * (1.0f - gg[i] * gg[i]);
            const float d_f = dc * (ki[i] * cpv[i]) * gf[i] * (1.0f - gf[i]);
            dcc[i] = dc * gf[i];
            float *ar = At + (4 * q + i) * kBLd + u;
            ar[0] = d_i; ar[kBR] = d_f; ar[2 * kBR] = d_g; ar[3 * kBR] = d_o;
            if (ok[i]) {
                float *o = dg + ((size_t)t * a.N + rows[i]) * kBK + u;
                __builtin_nontemporal_store(d_i, o); __builtin_nontemporal_store(d_f, o + kBR);      // streamed out: the
                __builtin_nontemporal_store(d_g, o + 2 * kBR); __builtin_nontemporal_store(d_o, o + 3 * kBR);  // GEMMs read dG later
            }
        }
        if (sums && w == 0 && col == 0) {                  // (one lane per row group: rows 4 q + i of the tile)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int ra = ok[i] ? ai[i] : -1;
                *reinterpret_cast<float4 *>(actL + ((t & 1) * kBRows + 4 * q + i) * 4) =
                    make_float4(ra == 0 ? 1.f : 0.f, ra == 1 ? 1.f : 0.f, ra == 2 ? 1.f : 0.f, ra == 3 ? 1.f : 0.f);
            }
        }
        __syncthreads();                                   // the tile of step t is complete (the other buffer: step t + 1's
                                                           // reads finished before this step's writes began two barriers ago)
        if (t > 0) BPTT_FETCH(t - 1);                      // next step's inputs: in flight under the MFMAs
        if (sums) {                                        // this thread's column of the tile, by action (VALU under the MFMAs)
            const float *colp = At + tid;
            const float *ar = actL + (t & 1) * kBRows * 4;
#pragma unroll 4                                           // (fully unrolled, the 16 mask quads are hoisted: spills at 256 VGPRs)
            for (int r = 0; r < kBRows; r++) {             // (fmaf(1, v, s) = s + v, fmaf(0, v, s) = s: exact either way)
                const float v = colp[r * kBLd];
                const float4 m = *reinterpret_cast<const float4 *>(ar + 4 * r);
                s0 = fmaf(m.x, v, s0); s1 = fmaf(m.y, v, s1); s2 = fmaf(m.z, v, s2); s3 = fmaf(m.w, v, s3);
            }
        }
        // ---- dh_{t-1} contribution: acc = dG_t[rows, :] W_hh[:, units of this wave]
        acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
        const float *rd = At + col * kBLd + 4 * q;         // MFMA A layout: lane (row col, K slot q)
        // the A operands run kBDepth reads ahead of the MFMAs that consume them (a ring of registers: an LDS read issued right
        // before its use costs its full latency on a wave that has the SIMD to itself)
        constexpr int kBDepth = 6;
        float4 ring[kBDepth];
#pragma unroll
        for (int g = 0; g < kBDepth; g++) ring[g] = *reinterpret_cast<const float4 *>(rd + 16 * g);
#pragma unroll
        for (int g = 0; g < 32; g++) {
            const float4 av = ring[g % kBDepth];
            if (g + kBDepth < 32) ring[g % kBDepth] = *reinterpret_cast<const float4 *>(rd + 16 * (g + kBDepth));
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, B[4 * g + 0], acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, B[4 * g + 1], acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, B[4 * g + 2], acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, B[4 * g + 3], acc2, 0, 0, 0);
        }
        // (scheduling directives for the block above: kBDepth LDS reads, then 4 MFMAs : 1 LDS read, then the remaining MFMAs — left
        // to itself the compiler issues every read right before the MFMAs that consume it)
        __builtin_amdgcn_sched_group_barrier(0x100, kBDepth, 0);
#pragma unroll
        for (int g = 0; g < 32 - kBDepth; g++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * kBDepth, 0);
        acc += acc2;
    }
#undef BPTT_FETCH
    if (sums) {
        float *o = a.act_sums + (size_t)rt * 4 * kBK + tid;
        o[0] = s0; o[kBK] = s1; o[2 * kBK] = s2; o[3 * kBK] = s3;
    }
    // ---- gradient into the rollout's initial state
#pragma unroll
    for (int i = 0; i < 4; i++)
        if (ok[i]) {
            const size_t o = ((size_t)p * a.N + rows[i]) * kBR + u;
            a.dh0[o] = acc[i];
            a.dc0[o] = dcc[i];
        }
}

} // namespace atr

using namespace atr;

static int bptt_launch(const Bptt &a, bool pre, void *stream)
{
    const size_t lds = (size_t)2 * kBRows * kBLd * sizeof(float) +
                       (pre ? (size_t)kMaxActions * kBK * sizeof(float) + 2 * kBRows * 4 * sizeof(float) : 0);
    static bool attr_set[2] = {false, false};
    if (!attr_set[pre ? 1 : 0]) {
        const void *fn = pre ? (const void *)k_lstm_bptt<true> : (const void *)k_lstm_bptt<false>;
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -2;
        attr_set[pre ? 1 : 0] = true;
    }
    const unsigned grid = (unsigned)(a.P * ((a.N + kBRows - 1) / kBRows));
    if (pre) hipLaunchKernelGGL(k_lstm_bptt<true>, dim3(grid), dim3(512), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(k_lstm_bptt<false>, dim3(grid), dim3(512), lds, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int atr_lstm_bptt(const float *dh0_heads, const float *dh1_heads, const float *keep, const float *acts,
                             long long acts_pstride, const float *c_all, long long c_pstride, const float *whh0,
                             const float *whh1, float *dg, long long dg_pstride, float *dh_init, float *dc_init, int P, int T,
                             int N, int R, void *stream)
{
    if (!keep || !acts || !c_all || !whh0 || !dg || !dh_init || !dc_init || P < 1 || P > 2 || (P == 2 && !whh1) || T < 1 ||
        N < 1 || R != kBR)
        return -1;
    Bptt a;
    a.dh[0] = dh0_heads; a.dh[1] = dh1_heads; a.keep = keep; a.acts = acts; a.acts_ps = acts_pstride; a.c_all = c_all;
    a.c_ps = c_pstride; a.whh[0] = whh0; a.whh[1] = whh1; a.dg = dg; a.dg_ps = dg_pstride; a.dh0 = dh_init; a.dc0 = dc_init;
    a.P = P; a.T = T; a.N = N;
    a.bias[0] = a.bias[1] = nullptr; a.emb = nullptr; a.act = nullptr; a.act_ts = 0; a.emb_player = -1; a.n_act = 0;
    a.act_sums = nullptr;
    return bptt_launch(a, false, stream);
}

// atr_lstm_bptt over a rollout that stored the gate GEMM's OUTPUT (pre-activations without bias) instead of the activated gates.
extern "C" int atr_lstm_bptt_pre2(const float *dh0_heads, const float *dh1_heads, const float *keep, const float *pre,
                                  long long pre_pstride, const float *bias0, const float *bias1, const float *emb, int emb_player,
                                  int n_act, const long long *act_tracker, long long act_tstride, const float *c_all,
                                  long long c_pstride, const float *whh0, const float *whh1, float *dg, long long dg_pstride,
                                  float *dh_init, float *dc_init, float *act_sums, int P, int T, int N, int R, void *stream);

extern "C" int atr_lstm_bptt_pre(const float *dh0_heads, const float *dh1_heads, const float *keep, const float *pre,
                                 long long pre_pstride, const float *bias0, const float *bias1, const float *emb, int emb_player,
                                 int n_act, const long long *act_tracker, long long act_tstride, const float *c_all,
                                 long long c_pstride, const float *whh0, const float *whh1, float *dg, long long dg_pstride,
                                 float *dh_init, float *dc_init, int P, int T, int N, int R, void *stream)
{
    return atr_lstm_bptt_pre2(dh0_heads, dh1_heads, keep, pre, pre_pstride, bias0, bias1, emb, emb_player, n_act, act_tracker,
                              act_tstride, c_all, c_pstride, whh0, whh1, dg, dg_pstride, dh_init, dc_init, nullptr, P, T, N, R,
                              stream);
}

extern "C" long long atr_lstm_bptt_act_sums_floats(int N) { return (long long)((N + kBRows - 1) / kBRows) * 4 * kBK; }

extern "C" int atr_lstm_bptt_pre2(const float *dh0_heads, const float *dh1_heads, const float *keep, const float *pre,
                                  long long pre_pstride, const float *bias0, const float *bias1, const float *emb, int emb_player,
                                  int n_act, const long long *act_tracker, long long act_tstride, const float *c_all,
                                  long long c_pstride, const float *whh0, const float *whh1, float *dg, long long dg_pstride,
                                  float *dh_init, float *dc_init, float *act_sums, int P, int T, int N, int R, void *stream)
{
    if (act_sums && (!emb || n_act != 4)) return -1;     // (the by-action sums are built for the four-move action table)
    if (!keep || !pre || !bias0 || !c_all || !whh0 || !dg || !dh_init || !dc_init || P < 1 || P > 2 || (P == 2 && (!whh1 || !bias1)) ||
        T < 1 || N < 1 || R != kBR)
        return -1;
    if (emb && (!act_tracker || n_act < 1 || n_act > kMaxActions || emb_player < 0 || emb_player >= P)) return -1;
    Bptt a;
    a.dh[0] = dh0_heads; a.dh[1] = dh1_heads; a.keep = keep; a.acts = pre; a.acts_ps = pre_pstride; a.c_all = c_all;
    a.c_ps = c_pstride; a.whh[0] = whh0; a.whh[1] = whh1; a.dg = dg; a.dg_ps = dg_pstride; a.dh0 = dh_init; a.dc0 = dc_init;
    a.P = P; a.T = T; a.N = N;
    a.bias[0] = bias0; a.bias[1] = bias1; a.emb = emb; a.act = act_tracker; a.act_ts = act_tstride;
    a.emb_player = emb ? emb_player : -1; a.n_act = emb ? n_act : 0;
    a.act_sums = act_sums;
    return bptt_launch(a, true, stream);
}
