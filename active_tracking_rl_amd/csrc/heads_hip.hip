// heads_hip.hip — the A3C heads and loss of one player over all T*N stored steps as two HIP kernels (C ABI in
// include/atr_policy.h). Restates, batched: critic/actor heads (model.py:24-52,120-126 of the reference), the
// train branch of sample_action (softmax, log-softmax, entropy, log-prob of the taken action) and the loss terms of
// Agent.optimize (player_util.py:118-154): value_loss += 0.5 (R - V)^2, policy_loss -= logp gae + w_ent entropy,
// loss = mean_n sum_t (policy_loss + 0.5 value_loss) [+ L1(R_pred, r_tracker) for the tracker-aware target].
//
//   atr_heads_values    v = h W_c^T + b_c for every row (the n-step returns / GAE need the detached values first)
//   atr_heads_loss      per row: logits, softmax statistics, the loss terms and their analytic gradients; writes
//                       dL/dh, accumulates dL/d(actor, critic, aux weights) and the loss sums in registers, one
//                       partial record per workgroup, then a fixed-order reduction (reproducible).
// This replaces ~100 tiny PyTorch launches per A3C iteration (two skinny GEMMs per player, softmax / log_softmax /
// gather / mul / sum / neg, the loss arithmetic and all of their backward nodes).
// Layout: a row's R = 4 * L hidden units are held by L lanes of one wave (L = 16, 32 or 64; R = 128 -> 32 lanes).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/atr_policy.h"

namespace atr {

constexpr int kHeadsMaxA = 8;
constexpr int kHeadsBlock = 256;

__device__ __forceinline__ float4 h_ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ float dot4(const float4 &a, const float4 &b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w))); }
// sum over the L = 16 / 32 / 64 lanes that hold one row: within a 16-lane DPP row on the crossbar (quad permutes, then
// row rotations by 4 and 8), across rows through the permute network
__device__ __forceinline__ float row_sum(float v, int L)
{
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E /* quad_perm [2,3,0,1] */, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124 /* row_ror:4 */, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128 /* row_ror:8 */, 0xf, 0xf, true));
    if (L > 16) v += __shfl_xor(v, 16, 64);
    if (L > 32) v += __shfl_xor(v, 32, 64);
    return v;
}

// values[row * vstride + voff] = h[row] . w_c + b_c
__global__ __launch_bounds__(kHeadsBlock) void k_heads_values(const float *__restrict__ h, const float *__restrict__ wc,
                                                              const float *__restrict__ bc, float *__restrict__ values,
                                                              long long rows, int R, int vstride, int voff,
                                                              const float *__restrict__ h1, const float *__restrict__ wc1,
                                                              const float *__restrict__ bc1, int voff1)
{
    // (h1 != null: a second player's rows in the same launch — the upper half of the grid)
    const int half = h1 ? (int)(gridDim.x >> 1) : (int)gridDim.x;
    const bool second = h1 && (int)blockIdx.x >= half;
    if (second) { h = h1; wc = wc1; bc = bc1; voff = voff1; }
    const int bid = second ? (int)blockIdx.x - half : (int)blockIdx.x;
    const int L = R >> 2;
    const long long total = rows * L;
    for (long long idx = (long long)bid * blockDim.x + threadIdx.x; idx < total; idx += (long long)half * blockDim.x) {
        const int j = (int)(idx % L) * 4;
        const long long row = idx / L;
        const float v = row_sum(dot4(h_ld4(h + row * R + j), h_ld4(wc + j)), L);
        if (j == 0) values[row * vstride + voff] = v + bc[0];
    }
}

struct HeadsLoss {
    const float *h;            // [rows, R]
    const long long *actions;  // row r reads actions[(r / act_n) * act_tstride + r % act_n] (a flat vector: act_n >= rows)
    long long act_n, act_tstride;
    const float *ret, *gae, *val;   // [rows * stride + off]: n-step return, GAE term, value (as written by k_heads_values)
    int stride, off;
    const float *r_aux;        // nullable: reward the aux head predicts, [rows * aux_stride + aux_off]
    int aux_stride, aux_off;
    const float *wa, *ba, *wc, *waux, *baux;   // actor [A,R],[A]; critic [R]; aux [R],[1] (nullable)
    float scale;               // 1/N if this player's loss is trained, else 0 (statistics are still produced)
    float scale_aux;           // 1/N if the aux loss is part of the objective
    float w_ent;
    float *dh;                 // [rows, R]
    float *partial;            // [grid, rec] records: dWa [A*R] | dWc [R] | dWaux [R] | dba [A] | dbc | dbaux | sums[4]
    long long rows;
    int R, A;
    int grid;                  // workgroups of this player
};

// up to two players (different weights, hidden sequences, loss coefficients) in one launch: workgroups [0, p[0].grid) are
// player 0's, the rest player 1's
struct HeadsLossMulti { HeadsLoss p[2]; };

// NAx: compile-time size of the per-action arrays (4 = every registered id, 8 = the 'Moore' table): with 8 the unused half of
// the gradient accumulators and weight rows costs 40 VGPRs, i.e. a wave per SIMD of rows in flight
template <int NAx>
__global__ __launch_bounds__(kHeadsBlock) void k_heads_loss(const HeadsLossMulti g)
{
    const int pi = (int)blockIdx.x >= g.p[0].grid ? 1 : 0;
    const HeadsLoss &a = g.p[pi];
    const int bid = (int)blockIdx.x - (pi ? g.p[0].grid : 0), nblk = a.grid;
    __shared__ float red[kHeadsBlock / 16][(kHeadsMaxA + 2) * 4 + 1];   // per row-slot partials during the block reduction
    const int L = a.R >> 2, A = a.A;
    const int slot = (int)threadIdx.x / L, slots = kHeadsBlock / L;
    const int j = ((int)threadIdx.x % L) * 4;
    float4 gwa[NAx], gwc = make_float4(0.f, 0.f, 0.f, 0.f), gwx = gwc;
    float gba[NAx], gbc = 0.f, gbx = 0.f, s_pol = 0.f, s_val = 0.f, s_ent = 0.f, s_aux = 0.f;
#pragma unroll
    for (int i = 0; i < NAx; i++) { gwa[i] = make_float4(0.f, 0.f, 0.f, 0.f); gba[i] = 0.f; }
    float4 wa[NAx];
#pragma unroll
    for (int i = 0; i < NAx; i++) wa[i] = i < A ? h_ld4(a.wa + i * a.R + j) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 wc = h_ld4(a.wc + j);
    const float4 wx = a.waux ? h_ld4(a.waux + j) : make_float4(0.f, 0.f, 0.f, 0.f);
    // the next row's operands are fetched under this row's arithmetic (a row is a ~5 us dependent chain of loads, butterflies
    // and transcendentals; a CU holds only a few dozen rows at a time)
    const long long rstride = (long long)nblk * slots;
    long long row = (long long)bid * slots + slot;
    float4 hv_n = make_float4(0.f, 0.f, 0.f, 0.f);
    float v_n = 0.f, ret_n = 0.f, gae_n = 0.f, raux_n = 0.f;
    int act_n_ = 0;
    auto fetch = [&](long long r) {
        if (r < a.rows) {
            hv_n = h_ld4(a.h + r * a.R + j);
            const long long s_ = r * a.stride + a.off;
            v_n = a.val[s_]; ret_n = a.ret[s_]; gae_n = a.gae[s_];
            const unsigned an_ = (unsigned)(a.act_n > 0x7fffffffLL ? 0x7fffffffLL : a.act_n), at_ = (unsigned)r / an_;   // (rows < 2^31)
            act_n_ = (int)a.actions[(long long)at_ * a.act_tstride + (long long)((unsigned)r - at_ * an_)];
            if (a.r_aux) raux_n = a.r_aux[r * a.aux_stride + a.aux_off];
        }
    };
    fetch(row);
    for (; row < a.rows; row += rstride) {
        const float4 hv = hv_n;
        const float v = v_n, ret = ret_n, gae = gae_n, raux = raux_n;
        const int act = act_n_;
        fetch(row + rstride);
        float z[NAx];
#pragma unroll
        for (int i = 0; i < NAx; i++) z[i] = i < A ? row_sum(dot4(hv, wa[i]), L) + a.ba[i] : -INFINITY;
        const float pred = a.waux ? row_sum(dot4(hv, wx), L) + a.baux[0] : 0.f;
        // softmax statistics (the train branch of sample_action)
        float mx = z[0];
#pragma unroll
        for (int i = 1; i < NAx; i++) mx = fmaxf(mx, z[i]);
        float p[NAx], se = 0.f;
#pragma unroll
        for (int i = 0; i < NAx; i++) { p[i] = i < A ? expf(z[i] - mx) : 0.f; se += p[i]; }
        const float lse = mx + logf(se);
        float ent = 0.f, logp_a = 0.f;
#pragma unroll
        for (int i = 0; i < NAx; i++) {
            if (i < A) {
                p[i] = p[i] / se;
                const float lp = z[i] - lse;
                ent -= lp * p[i];
                if (i == act) logp_a = lp;
            }
        }
        // loss terms and their gradients
        const float dlogp = -a.scale * gae, dent = -a.scale * a.w_ent;
        const float dv = a.scale * 0.5f * (v - ret);
        float dpred = 0.f, aux_abs = 0.f;
        if (a.r_aux) {
            const float diff = pred - raux;
            aux_abs = fabsf(diff);
            dpred = a.scale_aux * (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f));
        }
        float dz[NAx];
#pragma unroll
        for (int i = 0; i < NAx; i++)
            dz[i] = i < A ? dlogp * ((i == act ? 1.f : 0.f) - p[i]) - dent * p[i] * ((z[i] - lse) + ent) : 0.f;
        float4 dh = make_float4(dv * wc.x + dpred * wx.x, dv * wc.y + dpred * wx.y, dv * wc.z + dpred * wx.z, dv * wc.w + dpred * wx.w);
#pragma unroll
        for (int i = 0; i < NAx; i++)
            if (i < A) {
                dh.x = fmaf(dz[i], wa[i].x, dh.x); dh.y = fmaf(dz[i], wa[i].y, dh.y);
                dh.z = fmaf(dz[i], wa[i].z, dh.z); dh.w = fmaf(dz[i], wa[i].w, dh.w);
                gwa[i].x = fmaf(dz[i], hv.x, gwa[i].x); gwa[i].y = fmaf(dz[i], hv.y, gwa[i].y);
                gwa[i].z = fmaf(dz[i], hv.z, gwa[i].z); gwa[i].w = fmaf(dz[i], hv.w, gwa[i].w);
                gba[i] += dz[i];
            }
        {
            float *dp = a.dh + row * a.R + j;             // streamed out (read by the BPTT launch)
            __builtin_nontemporal_store(dh.x, dp); __builtin_nontemporal_store(dh.y, dp + 1);
            __builtin_nontemporal_store(dh.z, dp + 2); __builtin_nontemporal_store(dh.w, dp + 3);
        }
        gwc.x = fmaf(dv, hv.x, gwc.x); gwc.y = fmaf(dv, hv.y, gwc.y); gwc.z = fmaf(dv, hv.z, gwc.z); gwc.w = fmaf(dv, hv.w, gwc.w);
        gwx.x = fmaf(dpred, hv.x, gwx.x); gwx.y = fmaf(dpred, hv.y, gwx.y); gwx.z = fmaf(dpred, hv.z, gwx.z); gwx.w = fmaf(dpred, hv.w, gwx.w);
        gbc += dv; gbx += dpred;
        s_pol += -logp_a * gae - a.w_ent * ent;
        s_val += 0.5f * (ret - v) * (ret - v);
        s_ent += ent;
        s_aux += aux_abs;
    }
    // block reduction over the row slots (fixed order), one record per workgroup
    __shared__ float4 wred[kHeadsBlock];
    const int rec = (A + 2) * a.R + (A + 2) + 4;
    float *out = a.partial + (size_t)bid * rec;
#pragma unroll
    for (int q = 0; q < NAx + 2; q++) {
        if (q < A + 2) {                      // uniform: weight vector q = actor row q | critic | aux
            float4 g = gwx;
            if (q < NAx && q < A) g = gwa[q < NAx ? q : 0];
            else if (q == A) g = gwc;
            __syncthreads();
            wred[threadIdx.x] = g;
            __syncthreads();
            if (slot == 0) {
                float4 acc = wred[threadIdx.x];
                for (int sl = 1; sl < slots; sl++) {
                    const float4 t = wred[sl * L + (int)threadIdx.x];
                    acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
                }
                *reinterpret_cast<float4 *>(out + q * a.R + j) = acc;
            }
        }
    }
    __syncthreads();
    if (j == 0) {   // row-lane 0 of every slot holds that slot's scalar partials
#pragma unroll
        for (int i = 0; i < NAx; i++) red[slot][i] = gba[i];
        red[slot][kHeadsMaxA] = gbc; red[slot][kHeadsMaxA + 1] = gbx;
        red[slot][kHeadsMaxA + 2] = s_pol; red[slot][kHeadsMaxA + 3] = s_val;
        red[slot][kHeadsMaxA + 4] = s_ent; red[slot][kHeadsMaxA + 5] = s_aux;
    }
    __syncthreads();
    if (threadIdx.x < (unsigned)(A + 2 + 4)) {
        const int q = (int)threadIdx.x;
        const int src = q < A ? q : kHeadsMaxA + (q - A);
        float acc = 0.f;
        for (int sl = 0; sl < slots; sl++) acc += red[sl][src];
        out[(A + 2) * a.R + q] = acc;
    }
}

// out[0..rec) = column sums of the records in a fixed order: 16 columns x 64 record slices per block; blocks [0, nb) are
// player 0's, [nb, 2 nb) player 1's. The block that holds the four loss sums (columns rec-4 .. rec-1, one block when they do
// not straddle a multiple of 16) also finishes: out[rec] = this player's contribution to the objective, and the statistics
// (policy, value, entropy, |aux error| sums x stats_scale) go to stats_out — no separate launches for either.
struct HeadsReduce {
    const float *partial[2];
    float *out[2];
    float *stats_out[2];       // nullable: 4 floats
    float scale[2], scale_aux[2];
    int nrec[2];
    float stats_scale;
    int rec, nb, fold;
};

__global__ __launch_bounds__(1024) void k_heads_reduce(const HeadsReduce f)
{
    __shared__ float red[64][17];
    __shared__ float tail[16];
    const int pi = (int)blockIdx.x >= f.nb ? 1 : 0, b = (int)blockIdx.x - pi * f.nb;
    const float *__restrict__ partial = f.partial[pi];
    float *__restrict__ out = f.out[pi];
    const int nrec = f.nrec[pi], rec = f.rec;
    const int jl = (int)(threadIdx.x & 15u), slice = (int)(threadIdx.x >> 4);
    const int j = b * 16 + jl;
    float acc = 0.f;
    if (j < rec)
        for (int r = slice; r < nrec; r += 64) acc += partial[(size_t)r * rec + j];
    red[slice][jl] = acc;
    __syncthreads();
    if (slice == 0) {
        acc = 0.f;
        for (int q = 0; q < 64; q++) acc += red[q][jl];
        if (j < rec) out[j] = acc;
        tail[jl] = acc;
    }
    __syncthreads();
    if (f.fold && threadIdx.x == 0 && b == (rec - 4) / 16) {
        const int o = (rec - 4) & 15;
        out[rec] = f.scale[pi] * (tail[o] + 0.5f * tail[o + 1]) + f.scale_aux[pi] * tail[o + 3];
        if (f.stats_out[pi])
            for (int q = 0; q < 4; q++) f.stats_out[pi][q] = tail[o + q] * f.stats_scale;
    }
}
// (the straddling case) out[rec] = this player's contribution to the objective
__global__ void k_heads_finish(float *out, int rec, float scale, float scale_aux, float *stats_out, float stats_scale)
{
    out[rec] = scale * (out[rec - 4] + 0.5f * out[rec - 3]) + scale_aux * out[rec - 1];
    if (stats_out)
        for (int q = 0; q < 4; q++) stats_out[q] = out[rec - 4 + q] * stats_scale;
}

static int heads_grid(long long rows, int R)
{
    const int slots = kHeadsBlock / (R / 4);
    long long b = (rows + slots - 1) / slots;
    if (b > 512) b = 512;
    return (int)(b < 1 ? 1 : b);
}

} // namespace atr

using namespace atr;

extern "C" int atr_heads_values2(const float *h, const float *wc, const float *bc, int voff, const float *h1, const float *wc1,
                                 const float *bc1, int voff1, float *values, long long rows, int R, int vstride, void *stream)
{
    const int L = R / 4;
    if (!h || !wc || !bc || !values || rows < 0 || R <= 0 || (R & 3) || (L != 16 && L != 32 && L != 64) || vstride < 1 ||
        voff < 0 || voff >= vstride)
        return -1;
    if (h1 && (!wc1 || !bc1 || voff1 < 0 || voff1 >= vstride)) return -1;
    if (rows == 0) return 0;
    long long blocks = (rows * L + kHeadsBlock - 1) / kHeadsBlock;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_heads_values, dim3((unsigned)(h1 ? 2 * blocks : blocks)), dim3(kHeadsBlock), 0, (hipStream_t)stream, h, wc,
                       bc, values, rows, R, vstride, voff, h1, wc1, bc1, voff1);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int atr_heads_values(const float *h, const float *wc, const float *bc, float *values, long long rows, int R,
                                int vstride, int voff, void *stream)
{
    return atr_heads_values2(h, wc, bc, voff, nullptr, nullptr, nullptr, 0, values, rows, R, vstride, stream);
}

extern "C" long long atr_heads_workspace_floats(long long rows, int R, int A)
{
    if (R <= 0 || (R & 3) || A < 1) return -1;
    return (long long)heads_grid(rows, R) * ((A + 2) * R + (A + 2) + 4);
}

static int heads_fill(HeadsLoss &a, const atr_heads_loss_args &p)
{
    const int L = p.R / 4;
    if (!p.h || !p.actions || !p.ret || !p.gae || !p.val || !p.wa || !p.ba || !p.wc || !p.dh || !p.grads_and_sums ||
        !p.workspace || p.rows <= 0 || p.R <= 0 || (p.R & 3) || (L != 16 && L != 32 && L != 64) || p.A < 1 || p.A > kHeadsMaxA ||
        p.stride < 1 || p.off < 0 || p.off >= p.stride || ((p.waux != nullptr) != (p.baux != nullptr)) || (p.r_aux && !p.waux) ||
        p.act_n < 1 || p.rows >= (1LL << 31))
        return -1;
    a.h = p.h; a.actions = p.actions; a.act_n = p.act_n; a.act_tstride = p.act_tstride; a.ret = p.ret; a.gae = p.gae;
    a.val = p.val; a.stride = p.stride; a.off = p.off; a.r_aux = p.r_aux; a.aux_stride = p.aux_stride; a.aux_off = p.aux_off;
    a.wa = p.wa; a.ba = p.ba; a.wc = p.wc; a.waux = p.waux; a.baux = p.baux; a.scale = p.scale; a.scale_aux = p.scale_aux;
    a.w_ent = p.w_ent; a.dh = p.dh; a.partial = p.workspace; a.rows = p.rows; a.R = p.R; a.A = p.A;
    a.grid = heads_grid(p.rows, p.R);
    return 0;
}

extern "C" int atr_heads_loss_multi(const atr_heads_loss_args *players, int count, float stats_scale, void *stream)
{
    if (!players || count < 1 || count > 2) return -1;
    HeadsLossMulti g;
    HeadsReduce f;
    for (int i = 0; i < 2; i++) {
        const atr_heads_loss_args &p = players[i < count ? i : 0];
        if (heads_fill(g.p[i], p) != 0) return -1;
        if (i >= count) g.p[i].grid = 0;
        f.partial[i] = p.workspace; f.out[i] = p.grads_and_sums; f.stats_out[i] = p.stats_out; f.scale[i] = p.scale;
        f.scale_aux[i] = p.scale_aux; f.nrec[i] = g.p[i].grid;
    }
    if (count == 2 && (players[0].R != players[1].R || players[0].A != players[1].A)) return -1;
    const int rec = (players[0].A + 2) * players[0].R + (players[0].A + 2) + 4;
    f.stats_scale = stats_scale; f.rec = rec; f.nb = (rec + 15) / 16; f.fold = (rec - 4) / 16 == (rec - 1) / 16 ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    if (players[0].A <= 4)
        hipLaunchKernelGGL(k_heads_loss<4>, dim3((unsigned)(g.p[0].grid + g.p[1].grid)), dim3(kHeadsBlock), 0, st, g);
    else
        hipLaunchKernelGGL(k_heads_loss<kHeadsMaxA>, dim3((unsigned)(g.p[0].grid + g.p[1].grid)), dim3(kHeadsBlock), 0, st, g);
    hipLaunchKernelGGL(k_heads_reduce, dim3((unsigned)(f.nb * count)), dim3(1024), 0, st, f);
    if (!f.fold)
        for (int i = 0; i < count; i++)
            hipLaunchKernelGGL(k_heads_finish, dim3(1), dim3(1), 0, st, players[i].grads_and_sums, rec, players[i].scale,
                               players[i].scale_aux, players[i].stats_out, stats_scale);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int atr_heads_loss(const float *h, const long long *actions, const float *ret, const float *gae,
                              const float *val, int stride, int off, const float *r_aux, int aux_stride, int aux_off,
                              const float *wa, const float *ba, const float *wc, const float *waux, const float *baux,
                              float scale, float scale_aux, float w_ent, float *dh, float *grads_and_sums,
                              float *workspace, long long rows, int R, int A, void *stream)
{
    atr_heads_loss_args p = {};
    p.h = h; p.actions = actions; p.act_n = rows > 0 ? rows : 1; p.act_tstride = 0; p.ret = ret; p.gae = gae; p.val = val;
    p.stride = stride; p.off = off; p.r_aux = r_aux; p.aux_stride = aux_stride; p.aux_off = aux_off; p.wa = wa; p.ba = ba;
    p.wc = wc; p.waux = waux; p.baux = baux; p.scale = scale; p.scale_aux = scale_aux; p.w_ent = w_ent; p.dh = dh;
    p.grads_and_sums = grads_and_sums; p.workspace = workspace; p.stats_out = nullptr; p.rows = rows; p.R = R; p.A = A;
    return atr_heads_loss_multi(&p, 1, 1.0f, stream);
}
