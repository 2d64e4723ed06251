// gate_cell_hip.hip — the rollout step's LSTMCell product with the cell as its epilogue (round 6).
//
// Reference: model.py:116,137,165,196 (nn.LSTMCell of both players), player_util.py:44-67 (one actor step). Since round 4 the
// step's two LSTMCell GEMMs are ONE product over rows [fc features (256) | k h_prev (128)], K = 384, against [W_ih | W_hh]
// [512, 384] per player — a library kernel (hipBLASLt) that wrote the gate pre-activations [2, N, 512] to memory, from where the
// step's last kernel (k_act_step) read them straight back: 33.5 MB per step at 4096 envs crossing HBM twice for nothing. This
// kernel is that product on the f32 matrix cores with the cell where the accumulators are:
//   * one workgroup = 128 rows x 64 gate columns = ALL FOUR gates of 16 hidden units ([i | f | g | o] x 16: the W rows
//     gate * 128 + u0 + 0..15), 4 waves each 32 rows x 64 columns = two v_mfma_f32_32x32x2_f32 accumulators; 512 workgroups at
//     4096 envs x 2 players = two per CU (72 KB of LDS each), one's MFMAs under the other's waits;
//   * operands DMA'd global -> LDS (global_load_lds_dwordx4: a wave instruction fills 8 rows x 128 B = one 32-float K chunk of
//     8 rows; 8 lanes cover one 128-byte line, so the global side is fully coalesced), three K chunks in flight around the one
//     being multiplied; the 16-byte pieces of a row are XOR-swizzled by (row >> 1) & 7 as they land (the lane -> global piece
//     assignment is free), which makes the MFMA-side ds_read_b128 conflict-free in all four of its 16-lane groups;
//   * an MFMA operand is FOUR consecutive k of one row per lane (one ds_read_b128 feeds four MFMAs): lane half h of the wave takes
//     k = 8 j + 4 h + s for MFMA s of 8-k block j on BOTH operands, i.e. a fixed permutation of the summation order;
//   * epilogue: the accumulators hold [i | f] and [g | o] of 16 units in the two 16-lane halves of a 32-column tile;
//     v_permlane16_swap_b32 (gfx950) exchanges them between lane rows so that every lane ends up with all four gates of 8
//     (row, unit) pairs — no LDS round trip — and evaluates the cell with the expressions of atr_cell.h (what k_act_step did):
//     ((acc + bias)), sigmoid / tanh on v_exp_f32 / v_rcp_f32, c' = f (k c) + i g, h' = o tanh(c').
// The gate tensor is still WRITTEN (once, streamed: the learner's BPTT re-activates it, atr_lstm_bptt_pre) but never read back
// by the rollout for a player whose cell ran here. The tracker-aware target's cell needs the tracker's action of the same step
// (fc_action_tracker(one_hot(a)), model.py:193-194) — drawn from the tracker's fresh hidden row AFTER its cell — so for that
// player (cell[p] == 0) the kernel stops at the pre-activations and k_act_step does its cell as before.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/atr_policy.h"
#include "atr_cell.h"

#ifndef ATR_GC_PRIO
#define ATR_GC_PRIO 0      // 1: the two workgroups of a CU alternate s_setprio per K chunk (measured: no effect on who finishes first)
#endif

namespace atr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kGcRows = 128, kGcCols = 64, kGcKc = 32;          // tile rows, tile columns (4 gates x 16 units), K chunk
constexpr int kGcStages = 3;
constexpr int kGcStageFloats = (kGcRows + kGcCols) * kGcKc;      // A chunk [128][32] then B chunk [64][32]
constexpr int kGcThreads = 256;

struct GateCell {
    const float *a[2];          // per player: rows [N, K] (row stride lda): [fc features | k h_prev]
    const float *w[2];          // per player: [4R, K] = [W_ih | W_hh], row-major
    const float *bias[2];       // per player: b_ih + b_hh [4R]
    float *pre[2];              // per player: [N, 4R] the product WITHOUT bias (the learner's store), nullable
    const float *c_prev[2];     // [N, R]
    float *h_out[2], *c_out[2]; // [N, R]
    const unsigned char *done_prev;    // [N] nullable: the previous step's done flags (k = done == 0 masks c_prev)
    long long lda;
    int cell[2];                // per player: run the cell here (else only `pre` is written)
    int N, K, rb_total;         // rb_total = ceil(N / 128)
    unsigned long long *probe;  // nullable: [workgroups][4] wall-clock stamps (tools/gate_cell_bench.py --timeline)
};

__device__ __forceinline__ unsigned long long gc_clock() { return wall_clock64(); }

__global__ __launch_bounds__(kGcThreads, 2) void k_gate_cell(const GateCell g)
{
    extern __shared__ __attribute__((aligned(16))) float gc_lds[];       // [kGcStages][kGcStageFloats]
    const int tid = (int)threadIdx.x, l = tid & 63, w = tid >> 6;
    // workgroup -> (player, row block, column block): the 8 column blocks of a row block run on ONE XCD back to back (workgroup
    // i goes to XCD i % 8), so a row block's operand rows come from HBM once per XCD and W stays in that XCD's L2.
    // Player-major: ALL of player 0's workgroups come first. Two workgroups share a CU, and the hardware issues the OLDER
    // wave first: the workgroup dispatched first gets the matrix pipe whenever it wants it and leaves its main loop ~5 us before
    // its CU mate (measured: 21.2 against 26.3 us at 4096 rows) — so the players whose epilogue is the long one (the cell:
    // player 0 always, when anybody) go first, and their cells run under the other workgroup's last K chunks.
    const int i = (int)blockIdx.x, xcd = i & 7, j = i >> 3;
    const int per_player = 8 * ((g.rb_total + 7) / 8);            // (row-block slot, column block) pairs per XCD and player
    const int p = j >= per_player ? 1 : 0, jj = j - p * per_player;
    const int cb = jj & 7, rb = xcd + 8 * (jj >> 3);
    if (rb >= g.rb_total) return;
    const unsigned long long t0 = g.probe ? gc_clock() : 0ull;
    const int m0 = rb * kGcRows, u0 = cb * 16;
    const int N = g.N, K = g.K, nch = K / kGcKc;
    const float *__restrict__ A = g.a[p];
    const float *__restrict__ W = g.w[p];
    // ---- DMA addressing: an instruction of this wave fills 8 consecutive tile rows (1 KB); lane l lands on row r8 + (l >> 3),
    // 16-byte slot l & 7, and must bring piece (l & 7) ^ ((row >> 1) & 7) of that row's K chunk
    const int dr = l >> 3, ds = l & 7;
    const float *ga[4];
    const float *gb[2];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int row = w * 32 + 8 * q + dr;                       // tile row of A
        int gr = m0 + row;
        if (gr >= N) gr = N - 1;                                   // (rows past N repeat the last row; never stored)
        ga[q] = A + (size_t)gr * g.lda + 4 * (ds ^ ((row >> 1) & 7));
    }
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int n = w * 16 + 8 * q + dr;                         // tile column = B row n: gate n >> 4, unit u0 + (n & 15)
        gb[q] = W + (size_t)((n >> 4) * 128 + u0 + (n & 15)) * K + 4 * (ds ^ ((n >> 1) & 7));
    }
    auto dma = [&](int c, int stage) {
        float *sa = gc_lds + (size_t)stage * kGcStageFloats, *sb = sa + kGcRows * kGcKc;
#pragma unroll
        for (int q = 0; q < 4; q++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(ga[q] + c * kGcKc),
                                             (__attribute__((address_space(3))) void *)(sa + (w * 32 + 8 * q) * kGcKc), 16, 0, 0);
#pragma unroll
        for (int q = 0; q < 2; q++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gb[q] + c * kGcKc),
                                             (__attribute__((address_space(3))) void *)(sb + (w * 16 + 8 * q) * kGcKc), 16, 0, 0);
    };
    // ---- MFMA-side addressing: lane (h = l >> 5, r = l & 31) reads piece 2 jb + h of rows w * 32 + r (A), r and 32 + r (B).
    // The LDS reads of the main loop are spelled as inline assembly: the compiler knows global_load_lds writes LDS and puts
    // `s_waitcnt vmcnt(0)` in front of every LDS read it can see — i.e. it would drain the two chunks in flight before the
    // first operand read of every chunk (checked in the ISA) — so it is not shown them; their lgkmcnt waits are placed by hand.
    const int h = l >> 5, r = l & 31;
    const int arow = w * 32 + r;
    const int asw = (arow >> 1) & 7, bsw0 = (r >> 1) & 7, bsw1 = ((32 + r) >> 1) & 7;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)gc_lds;
    uint32_t oa[4], ob0[4], ob1[4];
#pragma unroll
    for (int jb = 0; jb < 4; jb++) {
        oa[jb] = lds0 + (uint32_t)(arow * kGcKc + 4 * ((2 * jb + h) ^ asw)) * 4u;
        ob0[jb] = lds0 + (uint32_t)((kGcRows + r) * kGcKc + 4 * ((2 * jb + h) ^ bsw0)) * 4u;
        ob1[jb] = lds0 + (uint32_t)((kGcRows + 32 + r) * kGcKc + 4 * ((2 * jb + h) ^ bsw1)) * 4u;
    }
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    auto lds_read = [](uint32_t addr) {
        f32x4 v;
        asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
        return v;
    };
    // what the cell will read, requested before the main loop (a dependent round trip after it otherwise): this lane's 8
    // (row, unit) pairs are registers 2 s + odd of the accumulator map, unit u0 + (lane & 15)
    const int col = l & 31, gate_lo = col >> 4, un = col & 15;
    const int rowq = m0 + w * 32 + 4 * h;                          // + (q & 3) + 8 * (q >> 2)
    const int odd = (l >> 4) & 1;
    float bi = 0.f, bf = 0.f, bg = 0.f, bo = 0.f, cpv[8];
    unsigned dv[8];             // (raw done bytes: turned into the mask only in the epilogue — an ALU use here would wait for them)
#pragma unroll
    for (int s = 0; s < 8; s++) { cpv[s] = 0.f; dv[s] = 0u; }
    // (issued BEHIND the first two chunks' DMA: vector memory operations retire in order, so loads put in front of it hold up
    // the first `vmcnt` wait — measured: the cell workgroups then start ~1 us late and stay behind their CU mates for the whole
    // loop; inside the loop, under a condition, the compiler drains vmcnt to 0 in front of them. The loop's waits stay correct
    // whatever else is in flight: vmcnt(6) leaves the six NEWEST operations outstanding, and at least six — the next chunk's
    // DMA — are newer than the chunk awaited.)
    auto prefetch_cell = [&]() {
        const float *bs = g.bias[p];
        bi = bs[u0 + un]; bf = bs[128 + u0 + un]; bg = bs[256 + u0 + un]; bo = bs[384 + u0 + un];
#pragma unroll
        for (int s = 0; s < 8; s++) {
            const int q = 2 * s + odd;
            int row = rowq + (q & 3) + 8 * (q >> 2);
            if (row >= N) row = N - 1;
            cpv[s] = g.c_prev[p][(size_t)row * 128 + u0 + un];
            if (g.done_prev) dv[s] = g.done_prev[row];
        }
    };
    f32x16 acc0, acc1;
#pragma unroll
    for (int q = 0; q < 16; q++) { acc0[q] = 0.f; acc1[q] = 0.f; }
    dma(0, 0);
    if (nch > 1) dma(1, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (g.cell[p]) prefetch_cell();          // (straight-line code behind the first two chunks' DMA: see above)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int c = 0; c < nch; c++) {
        // chunk c has landed: everything this wave issued but the 6 loads of chunk c + 1 (vmcnt counts per wave, in order);
        // then the barrier: everybody else's part of it has landed too, and all waves are done reading chunk c - 1.
        // (a bare s_barrier: __syncthreads() is a fence over all memory and would wait for vmcnt(0))
        if (c + 1 < nch) asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (c + 2 < nch) dma(c + 2, (c + 2) % kGcStages);
#if ATR_GC_PRIO
        // the two workgroups of a CU take turns at the head of the issue arbiter, chunk by chunk (left alone the one dispatched
        // first wins every arbitration, leaves its loop ~5 us early and its mate finishes alone at a lone wave's efficiency)
        if (((c ^ p) & 1) != 0) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
#endif
        const uint32_t so = (uint32_t)((c % kGcStages) * kGcStageFloats) * 4u;
        f32x4 a = lds_read(oa[0] + so), b0 = lds_read(ob0[0] + so), b1 = lds_read(ob1[0] + so);
#pragma unroll
        for (int jb = 0; jb < 4; jb++) {
            // this block's operands have arrived (the asm ties the wait to the registers the MFMAs below read; the scheduling
            // barrier keeps it BELOW the previous block's MFMAs — it was hoisted above them otherwise, right behind its reads)
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b0), "+v"(b1));
            f32x4 na = a, nb0 = b0, nb1 = b1;
            if (jb + 1 < 4) {                // the next 8-k block's operands are read under this block's MFMAs
                na = lds_read(oa[jb + 1] + so); nb0 = lds_read(ob0[jb + 1] + so); nb1 = lds_read(ob1[jb + 1] + so);
            }
            __builtin_amdgcn_sched_barrier(0);      // (keep those reads ABOVE the MFMAs: left alone the scheduler sinks them below)
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, acc1, 0, 0, 0);
            a = na; b0 = nb0; b1 = nb1;
        }
    }
#if ATR_GC_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    const unsigned long long t1 = g.probe ? gc_clock() : 0ull;
    // ---- epilogue. C/D map of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    float *pre = g.pre[p];
    if (pre) {      // the product itself (no bias): streamed out, read a whole rollout later by the learner — or, for a player
                    // whose cell does not run here, by this step's k_act_step (plain stores then: it is read at once)
        float *o0 = pre + (size_t)gate_lo * 128 + u0 + un, *o1 = o0 + 256;      // acc0: gates i / f, acc1: gates g / o
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int row = rowq + (q & 3) + 8 * (q >> 2);
            if (row < N) {
                if (g.cell[p]) {
                    __builtin_nontemporal_store(acc0[q], o0 + (size_t)row * 512);
                    __builtin_nontemporal_store(acc1[q], o1 + (size_t)row * 512);
                } else {
                    o0[(size_t)row * 512] = acc0[q];
                    o1[(size_t)row * 512] = acc1[q];
                }
            }
        }
    }
    if (g.cell[p]) {
        float *ho = g.h_out[p], *co = g.c_out[p];
        // lane row (lane >> 4) even: keeps register 2 s (its i / g), receives f / o of the same (row, unit) from lane + 16;
        // lane row odd: keeps register 2 s + 1's f / o ... after the swap BOTH lane rows hold all four gates of one pair:
        //   swap(X = acc[2 s], Y = acc[2 s + 1]):  X = [even lane rows: X's own | odd lane rows: Y's even-lane-row value]
        //                                          Y = [even lane rows: X's odd-lane-row value | odd lane rows: Y's own]
        // with even lane rows = columns 0..15 (gate i resp. g) and odd = columns 16..31 (gate f resp. o): afterwards, in an even
        // lane row, X = i(reg 2 s), Y = f(reg 2 s); in an odd lane row, X = i(reg 2 s + 1), Y = f(reg 2 s + 1).
#pragma unroll
        for (int s = 0; s < 8; s++) {
            auto x0 = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc0[2 * s]), __float_as_uint(acc0[2 * s + 1]), false, false);
            auto x1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc1[2 * s]), __float_as_uint(acc1[2 * s + 1]), false, false);
            const float pi = __uint_as_float(x0[0]), pf = __uint_as_float(x0[1]);
            const float pg = __uint_as_float(x1[0]), po = __uint_as_float(x1[1]);
            const int q = 2 * s + odd;
            const int row = rowq + (q & 3) + 8 * (q >> 2);
            if (row < N) {
                const float k = dv[s] == 0u ? 1.0f : 0.0f, cv = cpv[s];
                // (atr_cell.h cell4 / k_act_step: pre = 1 * bias + acc, then the gates, c' = f (k c) + i g, h' = o tanh(c'))
                const float gi = sigmoidf_(fmaf(1.0f, bi, pi)), gf = sigmoidf_(fmaf(1.0f, bf, pf));
                const float gg = tanhf_(fmaf(1.0f, bg, pg)), go = sigmoidf_(fmaf(1.0f, bo, po));
                const float cn = gf * (k * cv) + gi * gg;
                const float hn = go * tanhf_(cn);
                co[(size_t)row * 128 + u0 + un] = cn;
                ho[(size_t)row * 128 + u0 + un] = hn;
            }
        }
    }
    if (g.probe && tid == 0) {
        unsigned long long *o = g.probe + (size_t)blockIdx.x * 4;
        o[0] = t0; o[1] = t1; o[2] = gc_clock();
        o[3] = (unsigned long long)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7);      // HW_REG_XCC_ID
    }
}

} // namespace atr

using namespace atr;

extern "C" int atr_gate_cell(const atr_gate_cell_args *args, void *stream)
{
    if (!args) return -1;
    const atr_gate_cell_args &a = *args;
    if (a.N < 1 || a.K < kGcKc || a.K % kGcKc || a.R != 128 || a.lda < a.K || (a.lda & 3)) return -1;
    GateCell g;
    for (int p = 0; p < 2; p++) {
        if (!a.a[p] || !a.w[p]) return -1;
        if (((uintptr_t)a.a[p] | (uintptr_t)a.w[p]) & 15u) return -1;
        if (a.cell[p] && (!a.bias[p] || !a.c_prev[p] || !a.h_out[p] || !a.c_out[p])) return -1;
        if (!a.cell[p] && !a.pre[p]) return -1;
        g.a[p] = a.a[p]; g.w[p] = a.w[p]; g.bias[p] = a.bias[p]; g.pre[p] = a.pre[p]; g.c_prev[p] = a.c_prev[p];
        g.h_out[p] = a.h_out[p]; g.c_out[p] = a.c_out[p]; g.cell[p] = a.cell[p] ? 1 : 0;
    }
    g.done_prev = a.done_prev; g.lda = a.lda; g.N = a.N; g.K = a.K; g.rb_total = (a.N + kGcRows - 1) / kGcRows;
    g.probe = (unsigned long long *)a.probe;
    const size_t lds = (size_t)kGcStages * kGcStageFloats * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void *)k_gate_cell, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -2;
        attr_set = true;
    }
    // row blocks are dealt to the XCDs round-robin: 8 * ceil(rb_total / 8) row-block slots x 16 (column block, player) pairs
    const unsigned grid = (unsigned)(((g.rb_total + 7) / 8) * 8 * 16);
    hipLaunchKernelGGL(k_gate_cell, dim3(grid), dim3(kGcThreads), lds, (hipStream_t)stream, g);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int atr_gate_cell_workgroups(int N) { return ((((N + kGcRows - 1) / kGcRows) + 7) / 8) * 8 * 16; }
