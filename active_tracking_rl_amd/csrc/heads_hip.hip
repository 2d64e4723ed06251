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
                                                              long long rows, int R, int vstride, int voff)
{
    const int L = R >> 2;
    const long long total = rows * L;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(idx % L) * 4;
        const long long row = idx / L;
        const float v = row_sum(dot4(h_ld4(h + row * R + j), h_ld4(wc + j)), L);
        if (j == 0) values[row * vstride + voff] = v + bc[0];
    }
}

struct HeadsLoss {
    const float *h;            // [rows, R]
    const long long *actions;  // [rows]
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
};

__global__ __launch_bounds__(kHeadsBlock) void k_heads_loss(HeadsLoss a)
{
    __shared__ float red[kHeadsBlock / 16][(kHeadsMaxA + 2) * 4 + 1];   // per row-slot partials during the block reduction
    const int L = a.R >> 2, A = a.A;
    const int slot = (int)threadIdx.x / L, slots = kHeadsBlock / L;
    const int j = ((int)threadIdx.x % L) * 4;
    float4 gwa[kHeadsMaxA], gwc = make_float4(0.f, 0.f, 0.f, 0.f), gwx = gwc;
    float gba[kHeadsMaxA], gbc = 0.f, gbx = 0.f, s_pol = 0.f, s_val = 0.f, s_ent = 0.f, s_aux = 0.f;
#pragma unroll
    for (int i = 0; i < kHeadsMaxA; i++) { gwa[i] = make_float4(0.f, 0.f, 0.f, 0.f); gba[i] = 0.f; }
    float4 wa[kHeadsMaxA];
#pragma unroll
    for (int i = 0; i < kHeadsMaxA; i++) wa[i] = i < A ? h_ld4(a.wa + i * a.R + j) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 wc = h_ld4(a.wc + j);
    const float4 wx = a.waux ? h_ld4(a.waux + j) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (long long row = (long long)blockIdx.x * slots + slot; row < a.rows; row += (long long)gridDim.x * slots) {
        const float4 hv = h_ld4(a.h + row * a.R + j);
        float z[kHeadsMaxA];
#pragma unroll
        for (int i = 0; i < kHeadsMaxA; i++) z[i] = i < A ? row_sum(dot4(hv, wa[i]), L) + a.ba[i] : -INFINITY;
        const float pred = a.waux ? row_sum(dot4(hv, wx), L) + a.baux[0] : 0.f;
        const long long s = row * a.stride + a.off;
        const float v = a.val[s], ret = a.ret[s], gae = a.gae[s];
        const int act = (int)a.actions[row];
        // softmax statistics (the train branch of sample_action)
        float mx = z[0];
#pragma unroll
        for (int i = 1; i < kHeadsMaxA; i++) mx = fmaxf(mx, z[i]);
        float p[kHeadsMaxA], se = 0.f;
#pragma unroll
        for (int i = 0; i < kHeadsMaxA; i++) { p[i] = i < A ? expf(z[i] - mx) : 0.f; se += p[i]; }
        const float lse = mx + logf(se);
        float ent = 0.f, logp_a = 0.f;
#pragma unroll
        for (int i = 0; i < kHeadsMaxA; i++) {
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
            const float diff = pred - a.r_aux[row * a.aux_stride + a.aux_off];
            aux_abs = fabsf(diff);
            dpred = a.scale_aux * (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f));
        }
        float dz[kHeadsMaxA];
#pragma unroll
        for (int i = 0; i < kHeadsMaxA; i++)
            dz[i] = i < A ? dlogp * ((i == act ? 1.f : 0.f) - p[i]) - dent * p[i] * ((z[i] - lse) + ent) : 0.f;
        float4 dh = make_float4(dv * wc.x + dpred * wx.x, dv * wc.y + dpred * wx.y, dv * wc.z + dpred * wx.z, dv * wc.w + dpred * wx.w);
#pragma unroll
        for (int i = 0; i < kHeadsMaxA; i++)
            if (i < A) {
                dh.x = fmaf(dz[i], wa[i].x, dh.x); dh.y = fmaf(dz[i], wa[i].y, dh.y);
                dh.z = fmaf(dz[i], wa[i].z, dh.z); dh.w = fmaf(dz[i], wa[i].w, dh.w);
                gwa[i].x = fmaf(dz[i], hv.x, gwa[i].x); gwa[i].y = fmaf(dz[i], hv.y, gwa[i].y);
                gwa[i].z = fmaf(dz[i], hv.z, gwa[i].z); gwa[i].w = fmaf(dz[i], hv.w, gwa[i].w);
                gba[i] += dz[i];
            }
        *reinterpret_cast<float4 *>(a.dh + row * a.R + j) = dh;
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
    float *out = a.partial + (size_t)blockIdx.x * rec;
#pragma unroll
    for (int q = 0; q < kHeadsMaxA + 2; q++) {
        if (q < A + 2) {                      // uniform: weight vector q = actor row q | critic | aux
            float4 g = gwx;
            if (q < kHeadsMaxA && q < A) g = gwa[q < kHeadsMaxA ? q : 0];
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
        for (int i = 0; i < kHeadsMaxA; i++) red[slot][i] = gba[i];
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

// out[0..rec) = column sums of the records in a fixed order: 16 columns x 64 record slices per block
__global__ __launch_bounds__(1024) void k_heads_reduce(const float *__restrict__ partial, int nrec, int rec, float *__restrict__ out)
{
    __shared__ float red[64][17];
    const int jl = (int)(threadIdx.x & 15u), slice = (int)(threadIdx.x >> 4);
    const int j = (int)blockIdx.x * 16 + jl;
    float acc = 0.f;
    if (j < rec)
        for (int r = slice; r < nrec; r += 64) acc += partial[(size_t)r * rec + j];
    red[slice][jl] = acc;
    __syncthreads();
    if (slice == 0 && j < rec) {
        acc = 0.f;
        for (int q = 0; q < 64; q++) acc += red[q][jl];
        out[j] = acc;
    }
}
// out[rec] = this player's contribution to the objective
__global__ void k_heads_finish(float *out, int rec, float scale, float scale_aux)
{
    out[rec] = scale * (out[rec - 4] + 0.5f * out[rec - 3]) + scale_aux * out[rec - 1];
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

extern "C" int atr_heads_values(const float *h, const float *wc, const float *bc, float *values, long long rows, int R,
                                int vstride, int voff, void *stream)
{
    const int L = R / 4;
    if (!h || !wc || !bc || !values || rows < 0 || R <= 0 || (R & 3) || (L != 16 && L != 32 && L != 64) || vstride < 1 ||
        voff < 0 || voff >= vstride)
        return -1;
    if (rows == 0) return 0;
    long long blocks = (rows * L + kHeadsBlock - 1) / kHeadsBlock;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_heads_values, dim3((unsigned)blocks), dim3(kHeadsBlock), 0, (hipStream_t)stream, h, wc, bc, values,
                       rows, R, vstride, voff);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" long long atr_heads_workspace_floats(long long rows, int R, int A)
{
    if (R <= 0 || (R & 3) || A < 1) return -1;
    return (long long)heads_grid(rows, R) * ((A + 2) * R + (A + 2) + 4);
}

extern "C" int atr_heads_loss(const float *h, const long long *actions, const float *ret, const float *gae,
                              const float *val, int stride, int off, const float *r_aux, int aux_stride, int aux_off,
                              const float *wa, const float *ba, const float *wc, const float *waux, const float *baux,
                              float scale, float scale_aux, float w_ent, float *dh, float *grads_and_sums,
                              float *workspace, long long rows, int R, int A, void *stream)
{
    const int L = R / 4;
    if (!h || !actions || !ret || !gae || !val || !wa || !ba || !wc || !dh || !grads_and_sums || !workspace || rows <= 0 ||
        R <= 0 || (R & 3) || (L != 16 && L != 32 && L != 64) || A < 1 || A > kHeadsMaxA || stride < 1 || off < 0 ||
        off >= stride || ((waux != nullptr) != (baux != nullptr)) || (r_aux && !waux))
        return -1;
    HeadsLoss a;
    a.h = h; a.actions = actions; a.ret = ret; a.gae = gae; a.val = val; a.stride = stride; a.off = off;
    a.r_aux = r_aux; a.aux_stride = aux_stride; a.aux_off = aux_off; a.wa = wa; a.ba = ba; a.wc = wc; a.waux = waux;
    a.baux = baux; a.scale = scale; a.scale_aux = scale_aux; a.w_ent = w_ent; a.dh = dh; a.partial = workspace;
    a.rows = rows; a.R = R; a.A = A;
    const int grid = heads_grid(rows, R), rec = (A + 2) * R + (A + 2) + 4;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_heads_loss, dim3((unsigned)grid), dim3(kHeadsBlock), 0, st, a);
    hipLaunchKernelGGL(k_heads_reduce, dim3((unsigned)((rec + 15) / 16)), dim3(1024), 0, st, workspace, grid, rec, grads_and_sums);
    hipLaunchKernelGGL(k_heads_finish, dim3(1), dim3(1), 0, st, grads_and_sums, rec, scale, scale_aux);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
