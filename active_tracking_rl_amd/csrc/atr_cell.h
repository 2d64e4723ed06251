// atr_cell.h — the LSTMCell step of ONE lane (four hidden units) and the actor head's partial logits, shared by the
// element-wise cell kernels (lstm_hip.hip) and the fused actor + env step kernel (track2d_hip.hip: k_act_step), so that
// both evaluate the same expressions in the same order (bit-identical states, logits and draws).
// torch.nn.LSTMCell semantics, gate order (i, f, g, o): model.py:110,172 of the reference.
#pragma once
#include <hip/hip_runtime.h>

#include "atr_sample.h"

namespace atr {

// sigmoid / tanh on the hardware exp and reciprocal (v_exp_f32, v_rcp_f32; ~1e-7 absolute, the forms atr_actor_step
// uses; __builtin_amdgcn_rcpf because __frcp_rn(x) is 1.0f / x, an IEEE division sequence): the cells of a small batch are ONE wave per SIMD executing a serial instruction stream, where libm's expf / tanhf
// and the IEEE division cost ~30 instructions per value against 4-5 here (k_act_step: 40 values per lane)
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }
__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, const float4 &v) { *reinterpret_cast<float4 *>(p) = v; }
__device__ __forceinline__ float4 fma4(float k, const float4 &a, const float4 &b)
{
    return make_float4(fmaf(k, a.x, b.x), fmaf(k, a.y, b.y), fmaf(k, a.z, b.z), fmaf(k, a.w, b.w));
}

// pre-activations (pi, pf, pg, po) of four hidden units, previous cell state cp, episode mask k of the previous step ->
// activated gates, c' = f (k c) + i g, h' = o tanh(c')
struct CellOut { float4 gi, gf, gg, go, c, h; };
__device__ __forceinline__ CellOut cell4(const float4 &pi, const float4 &pf, const float4 &pg, const float4 &po,
                                         const float4 &cp, float k)
{
    CellOut r;
    r.gi = make_float4(sigmoidf_(pi.x), sigmoidf_(pi.y), sigmoidf_(pi.z), sigmoidf_(pi.w));
    r.gf = make_float4(sigmoidf_(pf.x), sigmoidf_(pf.y), sigmoidf_(pf.z), sigmoidf_(pf.w));
    r.gg = make_float4(tanhf_(pg.x), tanhf_(pg.y), tanhf_(pg.z), tanhf_(pg.w));
    r.go = make_float4(sigmoidf_(po.x), sigmoidf_(po.y), sigmoidf_(po.z), sigmoidf_(po.w));
    r.c = make_float4(r.gf.x * (k * cp.x) + r.gi.x * r.gg.x, r.gf.y * (k * cp.y) + r.gi.y * r.gg.y,
                      r.gf.z * (k * cp.z) + r.gi.z * r.gg.z, r.gf.w * (k * cp.w) + r.gi.w * r.gg.w);
    r.h = make_float4(r.go.x * tanhf_(r.c.x), r.go.y * tanhf_(r.c.y), r.go.z * tanhf_(r.c.z), r.go.w * tanhf_(r.c.w));
    return r;
}

// actor head on a fresh hidden row held four units per lane by `rq` consecutive lanes (rq = R / 4, a power of two <= 64
// that divides the wave): this lane's partial logits over its four units, then the butterfly sum over the row's lanes
__device__ __forceinline__ void head_logits(const float4 &h, const float4 (&aw)[kMaxActions], int A, int rq,
                                            float (&logit)[kMaxActions])
{
#pragma unroll
    for (int q = 0; q < kMaxActions; q++) {
        logit[q] = 0.f;
        if (q < A) {
            const float4 w = aw[q];
            logit[q] = fmaf(h.x, w.x, fmaf(h.y, w.y, fmaf(h.z, w.z, h.w * w.w)));
        }
    }
    for (int msk = 1; msk < rq; msk <<= 1)
#pragma unroll
        for (int q = 0; q < kMaxActions; q++)
            if (q < A) logit[q] += __shfl_xor(logit[q], msk, 64);
}

} // namespace atr
