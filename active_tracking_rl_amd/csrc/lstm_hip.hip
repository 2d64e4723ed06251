// lstm_hip.hip — the recurrent half of the A3C learner/actor as fused element-wise HIP kernels (C ABI in
// include/atr_policy.h). The policies' LSTMCell(256 -> 128) runs once per env step for thousands of envs; the GEMMs
// stay with rocBLAS/hipBLASLt, everything between them is one launch per step here:
//
//   atr_lstm_cell_forward   gates = ig + k * hg (the episode-boundary mask k of the PREVIOUS step folded in: row
//                           scaling commutes with the hidden GEMM, so k * (h W) == (k h) W bit for bit), PyTorch's
//                           (i, f, g, o) chunk order, c' = f * (k c) + i g, h' = o tanh(c'). Optionally stores the gate
//                           activations for the backward pass. Replaces _thnn_fused_lstm_cell + 2 mask multiplies
//                           (+ the done -> float conversions in the rollout).
//   atr_lstm_cell_backward  one step of back-propagation through time: (dh from the heads, dh/dc arriving from step
//                           t+1, masked by this step's k) -> pre-activation gate gradients + dc for step t-1.
//   atr_gae_returns         the n-step returns and GAE terms of player_util.py:118-141 of the reference for all
//                           (env, agent) pairs in one launch instead of ~12 tiny launches per rollout step.
// HBM-bound streaming kernels: one thread per 4 hidden units (16-byte accesses), grid-stride.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/atr_policy.h"
#include "atr_cell.h"
#include "atr_sample.h"

namespace atr {

struct CellFwd {
    const float *ig[2];        // per player: [N, 4R] pre-activations from the input projection (bias included)
    const float *hg;           // [P, N, 4R]  h_prev W_hh^T
    const float *c_prev;       // + p * c_prev_ps + n * R
    long long c_prev_ps;
    const float *keep;         // [N] float mask of the previous step (nullable)
    const unsigned char *done; // [N] done flags of the previous step (nullable; k = done == 0)
    float *h_out, *c_out;      // + p * ps + n * R
    long long h_ps, c_ps;
    float *acts;               // nullable: + p * acts_ps + n * 4R, activated gates (i, f, g, o)
    long long acts_ps;
    int P, N, R;
    // actor extras (k_lstm_cell_fwd<true>, P == 1): tracker-action embedding added to the gates, and the actor head +
    // categorical draw on the fresh hidden row (R/4 lanes of one wave hold a row)
    const float *emb;          // nullable: [n_in, 4R] rows added to the pre-activations, row = act_in[n]
    const long long *act_in;
    const float *actor_w[2], *actor_b[2];   // per player: [A, R], [A]
    long long *actions_out;    // + p * N
    const float *bias[2];      // nullable, per player [4R]: added to the pre-activations (ig then comes from a bias-free bmm)
    const unsigned long long *counter;
    unsigned long long seed;
    unsigned ordinal;
    int A;
};

template <bool ACT> __global__ __launch_bounds__(256) void k_lstm_cell_fwd(CellFwd a)
{
    const int rq = a.R >> 2;
    const long long total = (long long)a.P * a.N * rq;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(idx % rq) * 4;
        const long long row = idx / rq;
        const int n = (int)(row % a.N), p = (int)(row / a.N);
        const float k = a.keep ? a.keep[n] : (a.done ? (a.done[n] == 0 ? 1.0f : 0.0f) : 1.0f);
        const float *ig = a.ig[p] + (long long)n * 4 * a.R + j;
        const float *hg = a.hg + ((long long)p * a.N + n) * 4 * a.R + j;
        float4 aw[kMaxActions];                 // ACT: this lane's slice of the actor head, fetched with everything else
        if (ACT) {
#pragma unroll
            for (int q = 0; q < kMaxActions; q++) aw[q] = q < a.A ? ld4(a.actor_w[p] + q * a.R + j) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float4 pi = fma4(k, ld4(hg), ld4(ig));
        float4 pf = fma4(k, ld4(hg + a.R), ld4(ig + a.R));
        float4 pg = fma4(k, ld4(hg + 2 * a.R), ld4(ig + 2 * a.R));
        float4 po = fma4(k, ld4(hg + 3 * a.R), ld4(ig + 3 * a.R));
        if (ACT && a.bias[p]) {
            const float *bb = a.bias[p] + j;
            pi = fma4(1.0f, ld4(bb), pi); pf = fma4(1.0f, ld4(bb + a.R), pf);
            pg = fma4(1.0f, ld4(bb + 2 * a.R), pg); po = fma4(1.0f, ld4(bb + 3 * a.R), po);
        }
        if (ACT && a.emb) {
            const float *e = a.emb + a.act_in[n] * 4 * a.R + j;
            pi = fma4(1.0f, ld4(e), pi); pf = fma4(1.0f, ld4(e + a.R), pf);
            pg = fma4(1.0f, ld4(e + 2 * a.R), pg); po = fma4(1.0f, ld4(e + 3 * a.R), po);
        }
        const float4 cp = ld4(a.c_prev + p * a.c_prev_ps + (long long)n * a.R + j);
        const CellOut o = cell4(pi, pf, pg, po, cp, k);
        const float4 gi = o.gi, gf = o.gf, gg = o.gg, go = o.go, c = o.c, h = o.h;
        st4(a.h_out + p * a.h_ps + (long long)n * a.R + j, h);
        st4(a.c_out + p * a.c_ps + (long long)n * a.R + j, c);
        if (a.acts) {
            float *ac = a.acts + p * a.acts_ps + (long long)n * 4 * a.R + j;
            st4(ac, gi); st4(ac + a.R, gf); st4(ac + 2 * a.R, gg); st4(ac + 3 * a.R, go);
        }
        if (ACT) {      // actor head on the fresh row: partial logits over this lane's 4 units, summed over the row's lanes
            float logit[kMaxActions];
            head_logits(h, aw, a.A, rq, logit);
            if (j == 0) {
#pragma unroll
                for (int q = 0; q < kMaxActions; q++) logit[q] = q < a.A ? logit[q] + a.actor_b[p][q] : -INFINITY;
                // player p draws under ordinal + p: the same numbers as two one-player launches with consecutive ordinals
                a.actions_out[(long long)p * a.N + n] = (long long)draw_action(logit, a.A, n, *a.counter, a.seed, a.ordinal + (unsigned)p);
            }
        }
    }
}

struct CellBwd {
    const float *dh_out;       // grad of this step's h from the heads: + p * dh_ps + n * R
    long long dh_ps;
    const float *dh_next;      // nullable: [P, N, R] dG_{t+1} W_hh (unmasked)
    float *dc_carry;           // [P, N, R]: in = dc_{t+1} f_{t+1} (unmasked, ignored if !has_next); out = dc_t f_t
    const float *keep_out;     // nullable [N]: this step's mask k_t (applies to what arrives from t+1)
    const float *keep_in;      // nullable [N]: k_{t-1} (applied to c_prev and the hidden gates at this step)
    const float *acts;         // + p * acts_ps + n * 4R
    long long acts_ps;
    const float *c;            // this step's cell state: + p * c_ps + n * R
    long long c_ps;
    const float *c_prev;       // + p * c_prev_ps + n * R
    long long c_prev_ps;
    float *dg;                 // out: pre-activation gate grads (= grad of ig): + p * dg_ps + n * 4R
    long long dg_ps;
    int has_next;
    int P, N, R;
};

__global__ __launch_bounds__(256) void k_lstm_cell_bwd(CellBwd a)
{
    const int rq = a.R >> 2;
    const long long total = (long long)a.P * a.N * rq;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(idx % rq) * 4;
        const long long row = idx / rq;
        const int n = (int)(row % a.N), p = (int)(row / a.N);
        const long long o1 = ((long long)p * a.N + n) * a.R + j;
        const float ko = a.keep_out ? a.keep_out[n] : 1.0f, ki = a.keep_in ? a.keep_in[n] : 1.0f;
        float4 dh = ld4(a.dh_out + p * a.dh_ps + (long long)n * a.R + j);
        float4 dcn = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.has_next) {
            dh = fma4(ko, ld4(a.dh_next + o1), dh);
            const float4 t = ld4(a.dc_carry + o1);
            dcn = make_float4(ko * t.x, ko * t.y, ko * t.z, ko * t.w);
        }
        const float *ac = a.acts + p * a.acts_ps + (long long)n * 4 * a.R + j;
        const float4 gi = ld4(ac), gf = ld4(ac + a.R), gg = ld4(ac + 2 * a.R), go = ld4(ac + 3 * a.R);
        const float4 c = ld4(a.c + p * a.c_ps + (long long)n * a.R + j);
        const float4 cp = ld4(a.c_prev + p * a.c_prev_ps + (long long)n * a.R + j);
        float di[4], df[4], dgg[4], dov[4], dcp[4];
        const float dhv[4] = {dh.x, dh.y, dh.z, dh.w}, dcnv[4] = {dcn.x, dcn.y, dcn.z, dcn.w};
        const float iv[4] = {gi.x, gi.y, gi.z, gi.w}, fv[4] = {gf.x, gf.y, gf.z, gf.w};
        const float gv[4] = {gg.x, gg.y, gg.z, gg.w}, ov[4] = {go.x, go.y, go.z, go.w};
        const float cv[4] = {c.x, c.y, c.z, c.w}, cpv[4] = {cp.x, cp.y, cp.z, cp.w};
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const float tc = tanhf_(cv[u]);     // the forward's tanh (atr_cell.h)
            const float dc = dcnv[u] + dhv[u] * ov[u] * (1.0f - tc * tc);
            dov[u] = dhv[u] * tc * ov[u] * (1.0f - ov[u]);
            di[u] = dc * gv[u] * iv[u] * (1.0f - iv[u]);
            dgg[u] = dc * iv[u] * (1.0f - gv[u] * gv[u]);
            df[u] = dc * (ki * cpv[u]) * fv[u] * (1.0f - fv[u]);
            dcp[u] = dc * fv[u];
        }
        float *dg = a.dg + p * a.dg_ps + (long long)n * 4 * a.R + j;
        st4(dg, make_float4(di[0], di[1], di[2], di[3]));
        st4(dg + a.R, make_float4(df[0], df[1], df[2], df[3]));
        st4(dg + 2 * a.R, make_float4(dgg[0], dgg[1], dgg[2], dgg[3]));
        st4(dg + 3 * a.R, make_float4(dov[0], dov[1], dov[2], dov[3]));
        st4(a.dc_carry + o1, make_float4(dcp[0], dcp[1], dcp[2], dcp[3]));
    }
}

// player_util.py:118-141: R_t = gamma R_{t+1} nd_t + r_t; delta_t = r_t + gamma v_{t+1} nd_t - v_t;
// gae_t = gae_{t+1} gamma tau nd_t + delta_t, for every (env, agent), newest step first.
__global__ __launch_bounds__(256) void k_gae(const float *__restrict__ rew, const float *__restrict__ val,
                                             const float *__restrict__ nd, float gamma, float tau,
                                             float *__restrict__ ret, float *__restrict__ gae, int T, int N, int A)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= N * A) return;
    const int n = i / A;
    const long long step = (long long)N * A;
    float r_run = val[(long long)T * step + i], g_run = 0.0f;
    for (int t = T - 1; t >= 0; t--) {
        const float k = nd[(long long)t * N + n], r = rew[t * step + i];
        const float v1 = val[(t + 1) * step + i], v0 = val[t * step + i];
        r_run = gamma * r_run * k + r;
        const float delta = r + gamma * v1 * k - v0;
        g_run = g_run * gamma * tau * k + delta;
        ret[t * step + i] = r_run;
        gae[t * step + i] = g_run;
    }
}

static unsigned grid_for(long long work)
{
    long long b = (work + 255) / 256;
    if (b > 8192) b = 8192;
    return (unsigned)(b < 1 ? 1 : b);
}

} // namespace atr

using namespace atr;

extern "C" int atr_lstm_cell_forward(const float *ig0, const float *ig1, const float *hg, const float *c_prev,
                                     long long c_prev_pstride, const float *keep, const unsigned char *done,
                                     float *h_out, long long h_pstride, float *c_out, long long c_pstride, float *acts,
                                     long long acts_pstride, int P, int N, int R, void *stream)
{
    if (!ig0 || !hg || !c_prev || !h_out || !c_out || P < 1 || P > 2 || (P == 2 && !ig1) || N < 0 || R <= 0 || (R & 3))
        return -1;
    if (N == 0) return 0;
    CellFwd a;
    a.ig[0] = ig0; a.ig[1] = ig1; a.hg = hg; a.c_prev = c_prev; a.c_prev_ps = c_prev_pstride; a.keep = keep; a.done = done;
    a.h_out = h_out; a.c_out = c_out; a.h_ps = h_pstride; a.c_ps = c_pstride; a.acts = acts; a.acts_ps = acts_pstride;
    a.P = P; a.N = N; a.R = R;
    a.emb = nullptr; a.act_in = nullptr; a.actor_w[0] = a.actor_w[1] = nullptr; a.actor_b[0] = a.actor_b[1] = nullptr;
    a.actions_out = nullptr; a.bias[0] = a.bias[1] = nullptr;
    a.counter = nullptr; a.seed = 0; a.ordinal = 0; a.A = 0;
    hipLaunchKernelGGL(k_lstm_cell_fwd<false>, dim3(grid_for((long long)P * N * (R / 4))), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int atr_lstm_cell_forward_act(const float *ig, const float *hg, const float *c_prev, const unsigned char *done,
                                         float *h_out, float *c_out, float *acts, const float *emb,
                                         const long long *act_in, const float *actor_w, const float *actor_b, int A,
                                         long long *actions_out, const unsigned long long *counter,
                                         unsigned long long seed, unsigned ordinal, int N, int R, void *stream)
{
    const int rq = R / 4;
    if (!ig || !hg || !c_prev || !h_out || !c_out || !actor_w || !actor_b || !actions_out || !counter || N < 0 || R <= 0 ||
        (R & 3) || (rq != 16 && rq != 32 && rq != 64) || A < 1 || A > kMaxActions || ((emb != nullptr) != (act_in != nullptr)))
        return -1;
    if (N == 0) return 0;
    CellFwd a;
    a.ig[0] = ig; a.ig[1] = nullptr; a.hg = hg; a.c_prev = c_prev; a.c_prev_ps = 0; a.keep = nullptr; a.done = done;
    a.h_out = h_out; a.c_out = c_out; a.h_ps = 0; a.c_ps = 0; a.acts = acts; a.acts_ps = 0; a.P = 1; a.N = N; a.R = R;
    a.emb = emb; a.act_in = act_in; a.actor_w[0] = actor_w; a.actor_w[1] = nullptr; a.actor_b[0] = actor_b; a.actor_b[1] = nullptr;
    a.actions_out = actions_out; a.bias[0] = a.bias[1] = nullptr;
    a.counter = counter; a.seed = seed; a.ordinal = ordinal; a.A = A;
    hipLaunchKernelGGL(k_lstm_cell_fwd<true>, dim3(grid_for((long long)N * rq)), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int atr_lstm_cell_forward_act1(const float *ig, const float *hg, const float *bias, const float *c_prev,
                                         const unsigned char *done, float *h_out, float *c_out, float *acts,
                                         const float *emb, const long long *act_in, const float *actor_w,
                                         const float *actor_b, int A, long long *actions_out,
                                         const unsigned long long *counter, unsigned long long seed, unsigned ordinal,
                                         int N, int R, void *stream)
{
    const int rq = R / 4;
    if (!ig || !hg || !c_prev || !h_out || !c_out || !actor_w || !actor_b || !actions_out || !counter || N < 0 || R <= 0 ||
        (R & 3) || (rq != 16 && rq != 32 && rq != 64) || A < 1 || A > kMaxActions || ((emb != nullptr) != (act_in != nullptr)))
        return -1;
    if (N == 0) return 0;
    CellFwd a;
    a.ig[0] = ig; a.ig[1] = nullptr; a.hg = hg; a.c_prev = c_prev; a.c_prev_ps = 0; a.keep = nullptr; a.done = done;
    a.h_out = h_out; a.c_out = c_out; a.h_ps = 0; a.c_ps = 0; a.acts = acts; a.acts_ps = 0; a.P = 1; a.N = N; a.R = R;
    a.emb = emb; a.act_in = act_in; a.actor_w[0] = actor_w; a.actor_w[1] = nullptr; a.actor_b[0] = actor_b; a.actor_b[1] = nullptr;
    a.actions_out = actions_out; a.bias[0] = bias; a.bias[1] = nullptr;
    a.counter = counter; a.seed = seed; a.ordinal = ordinal; a.A = A;
    hipLaunchKernelGGL(k_lstm_cell_fwd<true>, dim3(grid_for((long long)N * rq)), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int atr_lstm_cell_forward_act2(const float *ig, const float *hg, const float *bias0, const float *bias1,
                                          const float *c_prev, long long c_prev_pstride, const unsigned char *done,
                                          float *h_out, long long h_pstride, float *c_out, long long c_pstride, float *acts,
                                          long long acts_pstride, const float *actor_w0, const float *actor_b0,
                                          const float *actor_w1, const float *actor_b1, int A, long long *actions_out,
                                          const unsigned long long *counter, unsigned long long seed, unsigned ordinal,
                                          int N, int R, void *stream)
{
    const int rq = R / 4;
    if (!ig || !hg || !c_prev || !h_out || !c_out || !actor_w0 || !actor_b0 || !actor_w1 || !actor_b1 || !actions_out ||
        !counter || N < 0 || R <= 0 || (R & 3) || (rq != 16 && rq != 32 && rq != 64) || A < 1 || A > kMaxActions)
        return -1;
    if (N == 0) return 0;
    CellFwd a;
    a.ig[0] = ig; a.ig[1] = ig + (size_t)N * 4 * R; a.hg = hg; a.c_prev = c_prev; a.c_prev_ps = c_prev_pstride;
    a.keep = nullptr; a.done = done; a.h_out = h_out; a.c_out = c_out; a.h_ps = h_pstride; a.c_ps = c_pstride;
    a.acts = acts; a.acts_ps = acts_pstride; a.P = 2; a.N = N; a.R = R;
    a.emb = nullptr; a.act_in = nullptr; a.actor_w[0] = actor_w0; a.actor_w[1] = actor_w1; a.actor_b[0] = actor_b0;
    a.actor_b[1] = actor_b1; a.actions_out = actions_out; a.bias[0] = bias0; a.bias[1] = bias1;
    a.counter = counter; a.seed = seed; a.ordinal = ordinal; a.A = A;
    hipLaunchKernelGGL(k_lstm_cell_fwd<true>, dim3(grid_for((long long)2 * N * rq)), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int atr_lstm_cell_backward(const float *dh_out, long long dh_pstride, const float *dh_next, float *dc_carry,
                                      const float *keep_out, const float *keep_in, const float *acts,
                                      long long acts_pstride, const float *c, long long c_pstride, const float *c_prev,
                                      long long c_prev_pstride, float *dg, long long dg_pstride, int has_next, int P,
                                      int N, int R, void *stream)
{
    if (!dh_out || !dc_carry || !acts || !c || !c_prev || !dg || (has_next && !dh_next) || P < 1 || P > 2 || N < 0 ||
        R <= 0 || (R & 3))
        return -1;
    if (N == 0) return 0;
    CellBwd a;
    a.dh_out = dh_out; a.dh_ps = dh_pstride; a.dh_next = dh_next; a.dc_carry = dc_carry; a.keep_out = keep_out;
    a.keep_in = keep_in; a.acts = acts; a.acts_ps = acts_pstride; a.c = c; a.c_ps = c_pstride; a.c_prev = c_prev;
    a.c_prev_ps = c_prev_pstride; a.dg = dg; a.dg_ps = dg_pstride; a.has_next = has_next; a.P = P; a.N = N; a.R = R;
    hipLaunchKernelGGL(k_lstm_cell_bwd, dim3(grid_for((long long)P * N * (R / 4))), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int atr_gae_returns(const float *rewards, const float *values, const float *notdone, float gamma, float tau,
                               float *returns, float *gae, int T, int N, int A, void *stream)
{
    if (!rewards || !values || !notdone || !returns || !gae || T < 1 || N < 0 || A < 1) return -1;
    if (N == 0) return 0;
    hipLaunchKernelGGL(k_gae, dim3((unsigned)((N * A + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rewards, values,
                       notdone, gamma, tau, returns, gae, T, N, A);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
