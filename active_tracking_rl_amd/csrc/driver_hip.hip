// driver_hip.hip — the rollout driver's bookkeeping between the big kernels, one launch each (C ABI in
// include/atr_policy.h). In the reference these are a few Python statements per worker (player_util.py:98-106 LSTM state
// reset / detach, train.py:73-76 episode-length counters, shared_optim.py:80-134 SharedAdam.step); batched over N envs
// as tensor expressions each of them became a string of 5-20 tiny launches, which at ~5 us per launch added up to
// ~0.3 ms of a 6.7 ms iteration.
//
//   atr_rollout_begin   LSTM state [N,A,R] -> slot 0 of the per-player rollout cache [A,T+1,N,R]; the current observation
//                       -> row 0 of the rollout's observation store
//   atr_rollout_end     masked final LSTM state of slot T -> [N,A,R]; episode-length counters; the done flags of the whole
//                       rollout as the float keep-mask the learner's kernels read
//   atr_adam_step       SharedAdam.step (Adam + AMSGrad, float64 step / beta powers on the device) over the flat bucket,
//                       or torch.optim.Adam's form of it; atr_rmsprop_step: SharedRMSprop / torch.optim.RMSprop
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../../include/atr_policy.h"

namespace atr {

struct RolloutBegin {
    const float *hxs, *cxs;          // [N, A, R]
    float *h0, *c0;                  // slot 0 of player 0; player p at + p * pstride
    long long pstride;
    const uint32_t *obs_src;         // nullable: current observation, obs_words dwords
    uint32_t *obs_dst;
    long long obs_words;
    int N, A, R;
};

// The per-rollout constants of the actor (the weights do not change inside a rollout), made in the rollout's first launch
// instead of ~8 tiny ones (two bias adds, the tracker-action embedding table and its projection through W_ih, the
// concatenated LSTMCell weight, the draw counter's bump, the first rows' hidden columns): at 512 envs those were 2 % of an
// iteration. All optional (null = not wanted).
struct RolloutConsts {
    const float *w_ih[2], *w_hh[2], *b_ih[2], *b_hh[2];   // nn.LSTMCell parameters of the two players ([4R,F], [4R,R], [4R], [4R])
    float *bsum;                     // [2, 4R] = b_ih + b_hh
    float *w_cat;                    // [2, 4R, F + R] = [W_ih | W_hh] (the one-GEMM LSTMCell's weight)
    const float *fa_w, *fa_b;        // fc_action_tracker [F, A], [F] (tracker-aware target)
    float *emb_ih;                   // [A, 4R] = (fa_w^T + fa_b) W_ih[1]^T
    unsigned long long *counter;     // the action sampler's stream counter: += 1
    float *fh0;                      // slot 0 of the [features | k h] rows of player 0: hidden columns <- hxs; player p at + p * fh_pstride
    long long fh_pstride, fh_ld;     // (row stride; the hidden columns start at fh_ld - R)
    int F, A_act;
};

__global__ __launch_bounds__(256) void k_rollout_begin(RolloutBegin a, RolloutConsts k)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long nthreads = (long long)gridDim.x * blockDim.x;
    const long long nstate = (long long)a.N * a.A * (a.R / 4);
    if (i < nstate) {
        const int r4 = (int)(i % (a.R / 4));
        const long long na = i / (a.R / 4);
        const int p = (int)(na % a.A);
        const long long n = na / a.A;
        const float4 h = reinterpret_cast<const float4 *>(a.hxs)[i], c = reinterpret_cast<const float4 *>(a.cxs)[i];
        const long long o = (long long)p * a.pstride + n * a.R + 4 * r4;
        *reinterpret_cast<float4 *>(a.h0 + o) = h;
        *reinterpret_cast<float4 *>(a.c0 + o) = c;
        if (k.fh0) *reinterpret_cast<float4 *>(k.fh0 + (long long)p * k.fh_pstride + n * k.fh_ld + (k.fh_ld - a.R) + 4 * r4) = h;
    }
    if (a.obs_src != nullptr)
        for (long long j = i; j < a.obs_words; j += nthreads) a.obs_dst[j] = a.obs_src[j];
    const int G = 4 * a.R;                       // gate rows
    if (k.counter && i == 0) *k.counter += 1ull;
    if (k.bsum && i < 2LL * G / 4) {
        const int p = (int)(i / (G / 4)), j4 = (int)(i % (G / 4));
        const float4 x = reinterpret_cast<const float4 *>(k.b_ih[p])[j4], y = reinterpret_cast<const float4 *>(k.b_hh[p])[j4];
        reinterpret_cast<float4 *>(k.bsum)[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
    }
    if (k.w_cat) {
        const int K4 = (k.F + a.R) / 4, F4 = k.F / 4;
        const long long total = 2LL * G * K4;
        for (long long j = i; j < total; j += nthreads) {
            const int c4 = (int)(j % K4);
            const long long row = j / K4;             // p * G + g
            const int p = (int)(row / G), g = (int)(row % G);
            const float4 v = c4 < F4 ? reinterpret_cast<const float4 *>(k.w_ih[p] + (long long)g * k.F)[c4]
                                     : reinterpret_cast<const float4 *>(k.w_hh[p] + (long long)g * a.R)[c4 - F4];
            reinterpret_cast<float4 *>(k.w_cat)[j] = v;
        }
    }
    // emb_ih[a][g] = sum_c (fa_w[c][a] + fa_b[c]) * W_ih[1][g][c]: one WAVE per gate row g (coalesced reads of the row, the
    // A embedding rows from an LDS table, butterfly sums), the first G / 4 workgroups
    if (k.emb_ih && (int)blockIdx.x < G / 4) {
        __shared__ float etab[8 * 256];
        const int F = k.F, A = k.A_act;
        for (int t = (int)threadIdx.x; t < A * F; t += (int)blockDim.x) {
            const int act = t / F, c = t - act * F;
            etab[t] = k.fa_w[c * A + act] + k.fa_b[c];
        }
        __syncthreads();
        const int lane = (int)threadIdx.x & 63, g = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6);
        const float *wr = k.w_ih[1] + (long long)g * F;
        float acc[8];
#pragma unroll
        for (int a_ = 0; a_ < 8; a_++) acc[a_] = 0.f;
        for (int c = lane; c < F; c += 64) {
            const float w = wr[c];
#pragma unroll
            for (int a_ = 0; a_ < 8; a_++)
                if (a_ < A) acc[a_] = fmaf(etab[a_ * F + c], w, acc[a_]);
        }
#pragma unroll
        for (int a_ = 0; a_ < 8; a_++) {
            float v = acc[a_];
            for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
            if (a_ < A && lane == 0) k.emb_ih[a_ * G + g] = v;
        }
    }
}

struct RolloutEnd {
    const float *hT, *cT;            // slot T of player 0; player p at + p * pstride
    long long pstride;
    const uint8_t *dones;            // [T, N]; row T-1 masks the final state
    float *hxs, *cxs;                // [N, A, R]
    int *eps_len;                    // [N], in place
    float *keep;                     // [T, N] = (dones == 0)
    int T, N, A, R;
    const uint32_t *obs_src;         // optional: the observation after the last step (obs_words dwords) -> obs_dst, and
    uint32_t *obs_dst;               // dones[T-1] -> done_dst [N]: what the next rollout starts from, published here instead
    long long obs_words;             // of by three separate copy launches
    uint8_t *done_dst;
};

__global__ __launch_bounds__(256) void k_rollout_end(RolloutEnd a)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long nstate = (long long)a.N * a.A * (a.R / 4);
    if (i < nstate) {
        const int r4 = (int)(i % (a.R / 4));
        const long long na = i / (a.R / 4);
        const int p = (int)(na % a.A);
        const long long n = na / a.A;
        const float k = a.dones[(long long)(a.T - 1) * a.N + n] == 0 ? 1.f : 0.f;
        const long long o = (long long)p * a.pstride + n * a.R + 4 * r4;
        float4 h = *reinterpret_cast<const float4 *>(a.hT + o), c = *reinterpret_cast<const float4 *>(a.cT + o);
        h.x *= k; h.y *= k; h.z *= k; h.w *= k;
        c.x *= k; c.y *= k; c.z *= k; c.w *= k;
        reinterpret_cast<float4 *>(a.hxs)[i] = h;
        reinterpret_cast<float4 *>(a.cxs)[i] = c;
    }
    if (i < (long long)a.T * a.N) a.keep[i] = a.dones[i] == 0 ? 1.f : 0.f;
    if (a.obs_src)
        for (long long j = i; j < a.obs_words; j += (long long)gridDim.x * blockDim.x) a.obs_dst[j] = a.obs_src[j];
    if (a.done_dst && i < a.N) a.done_dst[i] = a.dones[(long long)(a.T - 1) * a.N + i];
    if (i < a.N) {
        // train.py:73-76 per env: the counter restarts at a done and counts the steps since; over T stored steps that is
        // eps_len * [no done in the rollout] + the number of steps after the last done
        int run = 0;
        bool alive = true;
#pragma unroll 4
        for (int t = a.T - 1; t >= 0; t--) {
            alive = alive && a.dones[(long long)t * a.N + i] == 0;
            run += alive ? 1 : 0;
        }
        a.eps_len[i] = (alive ? a.eps_len[i] : 0) + run;
    }
}

// state: [0] step, [1] beta1^step, [2] beta2^step (float64, as SharedAdam keeps them); scal: [0] step size, [1] the factor
// on sqrt(second moment). torch_eps = 0: SharedAdam (shared_optim.py:127-129,168-173 of the reference): eps is added to
// the raw sqrt(v) and both bias corrections sit in the step size; 1: torch.optim.Adam, which the reference's per-worker
// optimizer is (train.py:45-49): sqrt(v) / sqrt(1 - beta2^t) + eps, step size lr / (1 - beta1^t).
__global__ void k_adam_scalars(double *state, float *scal, double lr, double beta1, double beta2, int torch_eps)
{
    const double t = state[0] + 1.0, b1 = state[1] * beta1, b2 = state[2] * beta2;
    state[0] = t; state[1] = b1; state[2] = b2;
    if (torch_eps) {
        scal[0] = (float)(lr / (1.0 - b1));
        scal[1] = (float)(1.0 / sqrt(1.0 - b2));
    } else {
        scal[0] = (float)(lr * sqrt(1.0 - b2) / (1.0 - b1));
        scal[1] = 1.0f;                                          // x * 1 is exact: SharedAdam's denominator is untouched
    }
}

__global__ __launch_bounds__(256) void k_adam_update(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                     float *__restrict__ v, float *__restrict__ vmax,
                                                     const float *__restrict__ scal, float beta1, float beta2,
                                                     float omb1, float omb2, float eps, float wd, long long n)
{
    const float ss = scal[0], ds = scal[1];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float gi = g[i];
        const float pi = p[i];
        if (wd != 0.f) gi = fmaf(wd, pi, gi);
        const float mi = fmaf(omb1, gi, m[i] * beta1);            // exp_avg.mul_(beta1).add_(grad, alpha=1-beta1)
        const float vi = fmaf(omb2 * gi, gi, v[i] * beta2);       // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1-beta2)
        m[i] = mi; v[i] = vi;
        float d = vi;
        if (vmax != nullptr) { d = fmaxf(vmax[i], vi); vmax[i] = d; }   // AMSGrad: max of all second-moment estimates so far
        p[i] = pi - mi / (sqrtf(d) * ds + eps) * ss;
    }
}

// SharedRMSprop.step (shared_optim.py:43-87 of the reference; momentum = 0, not centered — its defaults, the only form
// main.py:89-90 constructs) and torch.optim.RMSprop with the same settings: v = alpha v + (1 - alpha) g^2,
// p -= lr g / (sqrt(v) + eps).
__global__ __launch_bounds__(256) void k_rmsprop_update(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ v,
                                                        float alpha, float oma, float eps, float lr, float wd, long long n)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float gi = g[i];
        const float pi = p[i];
        if (wd != 0.f) gi = fmaf(wd, pi, gi);
        const float vi = fmaf(oma * gi, gi, v[i] * alpha);
        v[i] = vi;
        p[i] = fmaf(-lr, gi / (sqrtf(vi) + eps), pi);
    }
}


// fc_action_tracker(one_hot(a_tracker)) added to the target's features (TAT.forward, model.py:193-194 of the reference) over
// all stored steps of a rollout: out[r][c] = f[r][c] + W[c][a[r]] + b[c] with W = fc_action_tracker.weight [C, A]. As tensor
// ops that is one_hot (scatter) + a [rows, A] x [A, C] GEMM + an add forward, and a GEMM + a column-sum reduction backward —
// 7 launches of up to 37 us on [81 920, 256] for what is a row gather: here one launch forward, two backward.
// Row r of a [T*N]-row operand takes its action from act[(r / act_n) * act_tstride + (r % act_n) * act_stride]: the rollout
// stores actions as [T, players, N], so one player's actions are N contiguous entries per step, steps players * N apart
// (act_n = N, act_tstride = players * N, act_stride = 1) — read in place instead of through a gathered copy. A flat vector
// is act_n >= rows.
__device__ __forceinline__ long long act_index(long long r, long long act_n, long long act_tstride, long long act_stride)
{
    // (rows and act_n fit 32 bits — checked by the host entry points: a 64-bit division is ~150 instructions on this ISA, and
    // these kernels do one per row and lane)
    const unsigned t = (unsigned)r / (unsigned)(act_n > 0x7fffffffLL ? 0x7fffffffLL : act_n);
    return (long long)t * act_tstride + (long long)((unsigned)r - t * (unsigned)(act_n > 0x7fffffffLL ? 0x7fffffffLL : act_n)) * act_stride;
}

constexpr int kEmbTabMax = 8 * 512;          // LDS table: A <= 8 rows of C <= 512 floats
__global__ __launch_bounds__(256) void k_embed_add(const float *__restrict__ f, const float *__restrict__ w,
                                                   const float *__restrict__ b, const long long *__restrict__ act,
                                                   long long act_stride, long long act_n, long long act_tstride,
                                                   float *__restrict__ out, long long rows, int C, int A, long long ldf)
{
    // the A embedding rows E[a][c] = w[c][a] + b[c] (the value the sum in TAT.forward adds) built once per workgroup in LDS:
    // per element one 16-byte LDS read instead of four strided 4-byte gathers + the bias (same values, same order of adds)
    __shared__ __attribute__((aligned(16))) float tab[kEmbTabMax];
    const bool use_tab = A * C <= kEmbTabMax;
    if (use_tab)
        for (int i = (int)threadIdx.x; i < A * C; i += (int)blockDim.x) {
            const int a_ = i / C, c_ = i - a_ * C;
            tab[i] = w[c_ * A + a_] + b[c_];
        }
    __syncthreads();
    const unsigned c4 = (unsigned)C / 4u;
    const unsigned total = (unsigned)rows * c4;                       // (< 2^31: checked by the host entry point)
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const long long r = (long long)(i / c4);
        const int c = (int)(i - (unsigned)r * c4) * 4;
        const int a = (int)act[act_index(r, act_n, act_tstride, act_stride)];
        float4 v = *reinterpret_cast<const float4 *>(f + r * ldf + c);
        if (use_tab) {
            const float4 e = *reinterpret_cast<const float4 *>(tab + a * C + c);
            v.x += e.x; v.y += e.y; v.z += e.z; v.w += e.w;
        } else {
            const float4 bb = *reinterpret_cast<const float4 *>(b + c);
            v.x += w[(c + 0) * A + a] + bb.x; v.y += w[(c + 1) * A + a] + bb.y;
            v.z += w[(c + 2) * A + a] + bb.z; v.w += w[(c + 3) * A + a] + bb.w;
        }
        *reinterpret_cast<float4 *>(out + r * C + c) = v;
    }
}

// backward, pass 1: per workgroup the sums of dout's rows by action (thread = column, fixed row order, eight rows' loads in
// flight), written as partial[a][c][wg] so that pass 2 reads them coalesced
constexpr int kEmbMaxA = 8;
__global__ __launch_bounds__(256) void k_embed_grad_partial(const float *__restrict__ dout, const long long *__restrict__ act,
                                                            long long act_stride, long long act_n, long long act_tstride,
                                                            float *__restrict__ partial, long long rows,
                                                            int C, int A, int rows_per_wg, int nwg)
{
    const long long r0 = (long long)blockIdx.x * rows_per_wg;
    const long long r1 = r0 + rows_per_wg < rows ? r0 + rows_per_wg : rows;
    for (int c = (int)threadIdx.x; c < C; c += (int)blockDim.x) {
        float acc[kEmbMaxA];
#pragma unroll
        for (int a = 0; a < kEmbMaxA; a++) acc[a] = 0.f;
        long long r = r0;
        for (; r + 8 <= r1; r += 8) {
            int av[8];
            float vv[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { av[u] = (int)act[act_index(r + u, act_n, act_tstride, act_stride)]; vv[u] = dout[(r + u) * C + c]; }
#pragma unroll
            for (int u = 0; u < 8; u++)
#pragma unroll
                for (int q = 0; q < kEmbMaxA; q++) acc[q] += q == av[u] ? vv[u] : 0.f;
        }
        for (; r < r1; r++) {
            const int a = (int)act[act_index(r, act_n, act_tstride, act_stride)];
            const float v = dout[r * C + c];
#pragma unroll
            for (int q = 0; q < kEmbMaxA; q++) acc[q] += q == a ? v : 0.f;
        }
#pragma unroll
        for (int a = 0; a < kEmbMaxA; a++)
            if (a < A) partial[((size_t)a * C + c) * nwg + blockIdx.x] = acc[a];
    }
}
// pass 2: one wave per column: dW[c][a] = sum over the workgroups' partials (lanes stride the records, butterfly sum: a
// fixed order), db[c] = sum_a dW[c][a]
__global__ __launch_bounds__(256) void k_embed_grad_reduce(const float *__restrict__ partial, float *__restrict__ dw,
                                                           float *__restrict__ db, int nwg, int C, int A)
{
    const int c = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = (int)(threadIdx.x & 63u);
    if (c >= C) return;
    float tot = 0.f;
    for (int a = 0; a < A; a++) {
        const float *p = partial + ((size_t)a * C + c) * nwg;
        float acc = 0.f;
        for (int g = lane; g < nwg; g += 64) acc += p[g];
        for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
        if (lane == 0) dw[c * A + a] = acc;
        tot += acc;
    }
    if (lane == 0) db[c] = tot;
}

} // namespace atr

using namespace atr;

extern "C" int atr_rollout_begin(const float *hxs, const float *cxs, float *h0, float *c0, long long pstride,
                                 const void *obs_src, void *obs_dst, long long obs_bytes, int N, int A, int R, void *stream)
{
    if (!hxs || !cxs || !h0 || !c0 || N <= 0 || A <= 0 || R <= 0 || (R & 3) || (obs_bytes & 3) || (pstride & 3)) return 1;
    if (((uintptr_t)hxs | (uintptr_t)cxs | (uintptr_t)h0 | (uintptr_t)c0) & 15u) return 1;
    if (obs_src && (((uintptr_t)obs_src | (uintptr_t)obs_dst) & 3u)) return 1;
    return atr_rollout_begin2(hxs, cxs, h0, c0, pstride, obs_src, obs_dst, obs_bytes, N, A, R, nullptr, stream);
}

extern "C" int atr_rollout_begin2(const float *hxs, const float *cxs, float *h0, float *c0, long long pstride,
                                  const void *obs_src, void *obs_dst, long long obs_bytes, int N, int A, int R,
                                  const atr_rollout_consts *consts, void *stream)
{
    if (!hxs || !cxs || !h0 || !c0 || N <= 0 || A <= 0 || R <= 0 || (R & 3) || (obs_bytes & 3) || (pstride & 3)) return 1;
    if (((uintptr_t)hxs | (uintptr_t)cxs | (uintptr_t)h0 | (uintptr_t)c0) & 15u) return 1;
    if (obs_src && (((uintptr_t)obs_src | (uintptr_t)obs_dst) & 3u)) return 1;
    RolloutBegin a{hxs, cxs, h0, c0, pstride, (const uint32_t *)obs_src, (uint32_t *)obs_dst, obs_bytes / 4, N, A, R};
    RolloutConsts k;
    memset(&k, 0, sizeof(k));
    const long long work = (long long)N * A * (R / 4);
    long long blocks = (work + 255) / 256;
    if (obs_src && blocks < 1024) blocks = 1024;
    if (consts) {
        const atr_rollout_consts &c = *consts;
        if (A != 2 || c.F <= 0 || (c.F & 3)) return 1;
        const bool need_w = c.bsum || c.w_cat || c.emb_ih;
        for (int p = 0; p < 2 && need_w; p++)
            if (!c.w_ih[p] || !c.w_hh[p] || !c.b_ih[p] || !c.b_hh[p] ||
                (((uintptr_t)c.w_ih[p] | (uintptr_t)c.w_hh[p] | (uintptr_t)c.b_ih[p] | (uintptr_t)c.b_hh[p]) & 15u)) return 1;
        if (c.emb_ih && (!c.fa_w || !c.fa_b || c.A_act < 1 || c.A_act > 8 || c.F > 256)) return 1;
        if (((uintptr_t)c.bsum | (uintptr_t)c.w_cat | (uintptr_t)c.fh0) & 15u) return 1;
        if (c.fh0 && (c.fh_ld < R || (c.fh_ld & 3) || (c.fh_pstride & 3))) return 1;
        for (int p = 0; p < 2; p++) { k.w_ih[p] = c.w_ih[p]; k.w_hh[p] = c.w_hh[p]; k.b_ih[p] = c.b_ih[p]; k.b_hh[p] = c.b_hh[p]; }
        k.bsum = c.bsum; k.w_cat = c.w_cat; k.fa_w = c.fa_w; k.fa_b = c.fa_b; k.emb_ih = c.emb_ih;
        k.counter = c.counter; k.fh0 = c.fh0; k.fh_pstride = c.fh_pstride; k.fh_ld = c.fh_ld; k.F = c.F; k.A_act = c.A_act;
        if (c.emb_ih && blocks < R) blocks = R;           // (G / 4 = R workgroups make emb_ih: one wave per gate row)
        if (c.w_cat && blocks < 384) blocks = 384;
    }
    hipLaunchKernelGGL(k_rollout_begin, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, k);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

extern "C" int atr_rollout_end(const float *hT, const float *cT, long long pstride, const uint8_t *dones, float *hxs,
                               float *cxs, int *eps_len, float *keep, int T, int N, int A, int R, void *stream)
{
    return atr_rollout_end2(hT, cT, pstride, dones, hxs, cxs, eps_len, keep, T, N, A, R, nullptr, nullptr, 0, nullptr, stream);
}

extern "C" int atr_rollout_end2(const float *hT, const float *cT, long long pstride, const uint8_t *dones, float *hxs,
                                float *cxs, int *eps_len, float *keep, int T, int N, int A, int R, const void *obs_src,
                                void *obs_dst, long long obs_bytes, uint8_t *done_dst, void *stream)
{
    if (!hT || !cT || !dones || !hxs || !cxs || !eps_len || !keep || T <= 0 || N <= 0 || A <= 0 || R <= 0 || (R & 3) || (pstride & 3))
        return 1;
    if (((uintptr_t)hxs | (uintptr_t)cxs | (uintptr_t)hT | (uintptr_t)cT) & 15u) return 1;
    if ((obs_src != nullptr) != (obs_dst != nullptr) || obs_bytes < 0 || (obs_bytes & 3) ||
        (((uintptr_t)obs_src | (uintptr_t)obs_dst) & 3u))
        return 1;
    RolloutEnd a{hT, cT, pstride, dones, hxs, cxs, eps_len, keep, T, N, A, R, (const uint32_t *)obs_src, (uint32_t *)obs_dst,
                 obs_src ? obs_bytes / 4 : 0, done_dst};
    long long work = (long long)N * A * (R / 4);
    if (work < (long long)T * N) work = (long long)T * N;
    hipLaunchKernelGGL(k_rollout_end, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

extern "C" int atr_embed_add_ld(const float *f, long long ldf, const float *w, const float *b, const long long *actions,
                                long long act_stride, long long act_n, long long act_tstride, float *out, long long rows, int C,
                                int A, void *stream)
{
    if (!f || !w || !b || !actions || !out || rows <= 0 || C <= 0 || (C & 3) || A < 1 || A > kEmbMaxA || act_n < 1) return 1;
    if (ldf < C || (ldf & 3)) return 1;
    if (rows * (C / 4) >= (1LL << 31)) return 1;
    if (((uintptr_t)f | (uintptr_t)out | (uintptr_t)b) & 15u) return 1;
    long long blocks = (rows * (C / 4) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_embed_add, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, f, w, b, actions, act_stride, act_n,
                       act_tstride, out, rows, C, A, ldf);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

extern "C" int atr_embed_add(const float *f, const float *w, const float *b, const long long *actions, long long act_stride,
                             long long act_n, long long act_tstride, float *out, long long rows, int C, int A, void *stream)
{
    return atr_embed_add_ld(f, C, w, b, actions, act_stride, act_n, act_tstride, out, rows, C, A, stream);
}

__global__ __launch_bounds__(256) void k_relu_backward_ld(const float *__restrict__ df, const float *__restrict__ f, long long ldf,
                                                          float *__restrict__ out, long long rows, int C)
{
    const unsigned c4 = (unsigned)C / 4u;
    const unsigned total = (unsigned)rows * c4;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const long long r = (long long)(i / c4);
        const int c = (int)(i - (unsigned)r * c4) * 4;
        const float4 a = *reinterpret_cast<const float4 *>(f + r * ldf + c);
        float4 d = *reinterpret_cast<const float4 *>(df + r * C + c);
        d.x = a.x > 0.f ? d.x : 0.f; d.y = a.y > 0.f ? d.y : 0.f; d.z = a.z > 0.f ? d.z : 0.f; d.w = a.w > 0.f ? d.w : 0.f;
        *reinterpret_cast<float4 *>(out + r * C + c) = d;
    }
}

extern "C" int atr_relu_backward_ld(const float *df, const float *f, long long ldf, float *out, long long rows, int C, void *stream)
{
    if (!df || !f || !out || rows <= 0 || C <= 0 || C % 4 || ldf < C || ldf % 4 || rows * (long long)(C / 4) >= (1LL << 31)) return 1;
    const long long total = rows * (C / 4);
    const long long blocks = total < 4096LL * 256 ? (total + 255) / 256 : 4096;
    hipLaunchKernelGGL(k_relu_backward_ld, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, df, f, ldf, out, rows, C);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

extern "C" long long atr_embed_grad_workspace_floats(long long rows, int C, int A)
{
    if (rows <= 0 || C <= 0 || A < 1 || A > kEmbMaxA) return -1;
    const long long nwg = rows < 1024 * 64 ? (rows + 63) / 64 : 1024;
    return nwg * A * C;
}

extern "C" int atr_embed_grad(const float *dout, const long long *actions, long long act_stride, long long act_n,
                              long long act_tstride, float *dw, float *db, float *workspace, long long rows, int C, int A,
                              void *stream)
{
    if (!dout || !actions || !dw || !db || !workspace || rows <= 0 || C <= 0 || A < 1 || A > kEmbMaxA || act_n < 1) return 1;
    if (rows >= (1LL << 31)) return 1;
    const long long nwg = rows < 1024 * 64 ? (rows + 63) / 64 : 1024;
    const int rpw = (int)((rows + nwg - 1) / nwg);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_embed_grad_partial, dim3((unsigned)nwg), dim3(256), 0, st, dout, actions, act_stride, act_n, act_tstride,
                       workspace, rows, C, A, rpw, (int)nwg);
    hipLaunchKernelGGL(k_embed_grad_reduce, dim3((unsigned)((C * 64 + 255) / 256)), dim3(256), 0, st, workspace, dw, db, (int)nwg, C, A);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

// ---- the tracker-action embedding folded out of the learner's big tensors (round 6) -------------------------------------------
// TAT.forward adds fc_action_tracker(one_hot(a_tracker)) = E[a] to the target's fc features before the LSTMCell (model.py:193-194
// of the reference). With S[a][j] = the sum of dG[r][j] over the rows r whose tracker action is a (atr_lstm_bptt_pre2 makes it per
// row tile, beside the recurrence), everything the embedding contributes to the backward pass is [4 x 512]-sized algebra:
//     dW_ih  = dG^T (f + E[a])  = dG^T f  +  S^T E          (the grouped weight-gradient launch contracts the raw features)
//     dE[a]  = sum over rows with action a of (dG W_ih)[r]  =  (S W_ih)[a];   d weight[c][a] = dE[a][c],  d bias[c] = sum_a dE[a][c]
// k_sact_reduce: S from the tiles' partial sums (fixed order); k_embed_fold: blocks [0, J) add S^T E to dW_ih's rows, the blocks
// after them make dE 64 columns at a time.
// (both kernels are a handful of workgroups with short dependent chains: what they cost is load LATENCY, so every thread keeps
// 8-16 independent loads in flight and the partial sums meet in LDS in a fixed order)
__global__ __launch_bounds__(1024) void k_sact_reduce(const float *__restrict__ part, float *__restrict__ S, int tiles, int n)
{
    __shared__ float red[16][64];
    const int lc = (int)(threadIdx.x & 63u), g = (int)(threadIdx.x >> 6);       // 64 columns x 16 tile groups
    const int c = (int)blockIdx.x * 64 + lc;
    float acc = 0.f;
    if (c < n) {
        int t = g;
        for (; t + 7 * 16 < tiles; t += 8 * 16) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = part[(size_t)(t + 16 * k) * n + c];
#pragma unroll
            for (int k = 0; k < 8; k++) acc += v[k];
        }
        for (; t < tiles; t += 16) acc += part[(size_t)t * n + c];
    }
    red[g][lc] = acc;
    __syncthreads();
    if (g == 0 && c < n) {
        float r = red[0][lc];
#pragma unroll
        for (int k = 1; k < 16; k++) r += red[k][lc];
        S[c] = r;
    }
}

__global__ __launch_bounds__(1024) void k_embed_fold(const float *__restrict__ S, const float *__restrict__ fa_w,
                                                     const float *__restrict__ fa_b, const float *__restrict__ wih,
                                                     float *__restrict__ dwih, float *__restrict__ dfa_w, float *__restrict__ dfa_b,
                                                     int J, int C, int rows_per_block)
{
    const int b = (int)blockIdx.x, tid = (int)threadIdx.x;
    const int nb_w = (J + rows_per_block - 1) / rows_per_block;
    if (b < nb_w) {                                // dW_ih[j][:] += sum_a S[a][j] E[a][:],  E[a][c] = fa_w[c][a] + fa_b[c]
        for (int idx = tid; idx < rows_per_block * C; idx += 1024) {
            const int j = b * rows_per_block + idx / C, c = idx % C;
            if (j >= J) break;
            const float4 w = *reinterpret_cast<const float4 *>(fa_w + 4 * c);
            const float bb = fa_b[c];
            float *d = dwih + (size_t)j * C + c;
            *d = *d + (((S[j] * (w.x + bb) + S[J + j] * (w.y + bb)) + S[2 * J + j] * (w.z + bb)) + S[3 * J + j] * (w.w + bb));
        }
        return;
    }
    // dE[:, c] for 64 columns: j in 16 interleaved groups (8 loads of W in flight per thread), summed in a fixed order
    __shared__ float red[16][4][64];
    const int lc = tid & 63, g = tid >> 6;
    const int c = (b - nb_w) * 64 + lc;
    float e0 = 0.f, e1 = 0.f, e2 = 0.f, e3 = 0.f;
    if (c < C) {
        int j = g;
        for (; j + 7 * 16 < J; j += 8 * 16) {
            float w[8];
#pragma unroll
            for (int k = 0; k < 8; k++) w[k] = wih[(size_t)(j + 16 * k) * C + c];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int jj = j + 16 * k;
                e0 = fmaf(S[jj], w[k], e0); e1 = fmaf(S[J + jj], w[k], e1); e2 = fmaf(S[2 * J + jj], w[k], e2); e3 = fmaf(S[3 * J + jj], w[k], e3);
            }
        }
        for (; j < J; j += 16) {
            const float w = wih[(size_t)j * C + c];
            e0 = fmaf(S[j], w, e0); e1 = fmaf(S[J + j], w, e1); e2 = fmaf(S[2 * J + j], w, e2); e3 = fmaf(S[3 * J + j], w, e3);
        }
    }
    red[g][0][lc] = e0; red[g][1][lc] = e1; red[g][2][lc] = e2; red[g][3][lc] = e3;
    __syncthreads();
    if (g == 0 && c < C) {
        float e[4];
#pragma unroll
        for (int a = 0; a < 4; a++) {
            float r = red[0][a][lc];
#pragma unroll
            for (int k = 1; k < 16; k++) r += red[k][a][lc];
            e[a] = r;
        }
        *reinterpret_cast<float4 *>(dfa_w + 4 * c) = make_float4(e[0], e[1], e[2], e[3]);
        dfa_b[c] = ((e[0] + e[1]) + e[2]) + e[3];
    }
}

extern "C" int atr_embed_fold(const float *act_sums, int tiles, const float *fa_w, const float *fa_b, const float *wih, float *dwih,
                              float *dfa_w, float *dfa_b, float *S, int J, int C, void *stream)
{
    if (!act_sums || !fa_w || !fa_b || !wih || !dwih || !dfa_w || !dfa_b || !S || tiles < 1 || J < 4 || (J & 3) || C < 1) return 1;
    if (((uintptr_t)fa_w | (uintptr_t)dfa_w) & 15u) return 1;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_sact_reduce, dim3((unsigned)((4 * J + 63) / 64)), dim3(1024), 0, st, act_sums, S, tiles, 4 * J);
    const int rpb = 4;                                    // dW_ih rows per workgroup of the first block range
    hipLaunchKernelGGL(k_embed_fold, dim3((unsigned)((J + rpb - 1) / rpb + (C + 63) / 64)), dim3(1024), 0, st, S, fa_w, fa_b, wih,
                       dwih, dfa_w, dfa_b, J, C, rpb);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

extern "C" int atr_adam_step(float *params, const float *grad, float *exp_avg, float *exp_avg_sq, float *max_exp_avg_sq,
                             double *state, float *scalars, double lr, double beta1, double beta2, double eps,
                             double weight_decay, int torch_eps, long long n, void *stream)
{
    if (!params || !grad || !exp_avg || !exp_avg_sq || !state || !scalars || n <= 0) return 1;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_adam_scalars, dim3(1), dim3(1), 0, st, state, scalars, lr, beta1, beta2, torch_eps);
    long long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_adam_update, dim3((unsigned)blocks), dim3(256), 0, st, params, grad, exp_avg, exp_avg_sq,
                       max_exp_avg_sq, scalars, (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2),
                       (float)eps, (float)weight_decay, n);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

// Up to 64 (source, offset, length) segments copied into one flat buffer by ONE launch: the gradients autograd hands back as
// ~40 separate small tensors go into the flat bucket this way (source null = the segment is zero-filled: a parameter the
// loss did not reach). Sources may start at any float (slices of a kernel's packed output), so the copy is per element.
struct ScatterSegs {
    const float *src[atr_scatter_max_segments];
    long long dst_off[atr_scatter_max_segments];
    int n[atr_scatter_max_segments];
    int blk_begin[atr_scatter_max_segments];
    int count;
};

__global__ __launch_bounds__(256) void k_scatter_segments(const ScatterSegs a, float *__restrict__ dst)
{
    const int b = (int)blockIdx.x;
    int q = 0;
    for (int j = 1; j < a.count; j++)
        if (b >= a.blk_begin[j]) q = j;
    const float *__restrict__ src = a.src[q];
    float *__restrict__ d = dst + a.dst_off[q];
    const int n = a.n[q], i0 = (b - a.blk_begin[q]) * 1024 + (int)threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int i = i0 + k * 256;
        if (i < n) d[i] = src ? src[i] : 0.0f;
    }
}

extern "C" int atr_scatter_segments(const float *const *src, const long long *dst_off, const int *n, int count, float *dst,
                                    void *stream)
{
    if (!src || !dst_off || !n || !dst || count < 1 || count > atr_scatter_max_segments) return 1;
    ScatterSegs a;
    int blocks = 0;
    for (int q = 0; q < count; q++) {
        if (n[q] < 0 || dst_off[q] < 0) return 1;
        a.src[q] = src[q]; a.dst_off[q] = dst_off[q]; a.n[q] = n[q]; a.blk_begin[q] = blocks;
        blocks += (n[q] + 1023) / 1024;
    }
    a.count = count;
    if (blocks == 0) return 0;
    hipLaunchKernelGGL(k_scatter_segments, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, dst);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

extern "C" int atr_rmsprop_step(float *params, const float *grad, float *square_avg, double lr, double alpha, double eps,
                                double weight_decay, long long n, void *stream)
{
    if (!params || !grad || !square_avg || n <= 0) return 1;
    long long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_rmsprop_update, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, params, grad, square_avg,
                       (float)alpha, (float)(1.0 - alpha), (float)eps, (float)lr, (float)weight_decay, n);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
