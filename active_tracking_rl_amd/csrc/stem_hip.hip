// stem_hip.hip — fused conv stem of the maze policies (perception.py:68-92 of the reference:
// conv(1->16,k3,s2,p1) 13->7, ReLU, conv(16->32,k3,s2,p1) 7->4, ReLU) as hand-written HIP for gfx950.
//
// Why: per frame the stem is 0.16 MFLOP on 676 B of input. As library GEMMs (Toeplitz-expanded weights) it costs
// 6.6x the FLOPs and round-trips a 784-wide activation through HBM (0.5 GB per 163 840-frame batch, forward and
// backward); MIOpen launches one Im2Col kernel per sample. Here a 256-thread workgroup owns one frame at a time:
// conv1's activation lives in LDS (double-buffered), each thread keeps a 72-value slice of the conv2 weights (or
// of their gradient) in registers, so 4 workgroups (16 waves) fit per CU and hide each other's LDS/global latency.
//
//   forward   x[M,169] -> y[M,512] (c,h,w order, post-ReLU)                         atr_stem_forward
//   backward  (x, y, dy) -> dW1[16,9], db1[16], dW2[32,144], db2[32]                atr_stem_backward
//             (conv1 is recomputed per frame; no gradient w.r.t. the observation is needed)
//
// Thread roles on the conv2 side (forward, dW2): tid = cp*16 + oh*4 + ciq — a pair of output channels {2cp, 2cp+1},
// one output row oh (4 positions) and a quarter of the input channels ci in [4ciq, 4ciq+4). One 16-byte LDS read of
// an a1 row feeds 3 taps x 4 positions x 2 channels = 24 FMAs; the four ci-quarters of a quad are summed with DPP.
// On the da1 side (conv2 transposed): tid = ci*16 + oh*4 + coq — input channel ci, output row oh, a quarter of the
// output channels; the 3x9 window of a1-gradients is reduced over the quad and added into LDS.
// fp32 FMA throughout (the reference computes in fp32); no MFMA: fp32 MFMA runs at the vector rate on gfx950.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/atr_policy.h"

namespace atr {

constexpr int kIn = 13, kInPad = 15;         // input side, padded by 1
constexpr int kC1 = 16, kH1 = 7;             // conv1 channels / output side
constexpr int kC2 = 32;                      // conv2 channels (4x4 outputs)
constexpr int kA1Rows = 9, kA1Stride = 12;   // a1 padded to rows -1..7, cols -1..7 (+3 so a row is 3 x 16 B)
constexpr int kA1Ch = kA1Rows * kA1Stride;   // 108 floats per channel
constexpr int kA1Size = kC1 * kA1Ch;         // 1728 floats
constexpr int kXSize = 228;                  // 15*15 = 225, rounded to a multiple of 4
constexpr int kW2 = kC2 * kC1 * 9;           // 4608
constexpr int kPartial = kW2 + kC2 + kC1 * 9 + kC1;  // per-workgroup partial gradient record: 4800 floats
constexpr int kThreads = 256;

// quad reduction on the DPP crossbar; wider ones through the permute network
__device__ __forceinline__ float quad_sum(float v)
{
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E /* quad_perm [2,3,0,1] */, 0xf, 0xf, true));
    return v;
}
__device__ __forceinline__ float xor_sum(float v, int mask) { return v + __shfl_xor(v, mask, 64); }

// Cooperative conv1 + ReLU of one frame: thread = (channel c = tid>>4, part = tid&15), outputs q = part + 16 i.
// xpad: 15x15 zero-bordered input; a1pad[16][9][12] zero-bordered output. w[9], bias: this thread's conv1 filter.
__device__ __forceinline__ void conv1_frame(const float *xpad, float *a1pad, const float (&w)[9], float bias, int tid)
{
    const int c = tid >> 4, part = tid & 15;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int q = part + 16 * i;
        if (q < 49) {
            const int oh = q / kH1, ow = q - oh * kH1;
            const float *xr = xpad + (2 * oh) * kInPad + 2 * ow;
            float acc = bias;
#pragma unroll
            for (int kh = 0; kh < 3; kh++)
#pragma unroll
                for (int kw = 0; kw < 3; kw++) acc = fmaf(xr[kh * kInPad + kw], w[kh * 3 + kw], acc);
            a1pad[c * kA1Ch + (oh + 1) * kA1Stride + ow + 1] = fmaxf(acc, 0.0f);
        }
    }
}

__device__ __forceinline__ void store_x(float *xpad, float v, int tid)
{
    if (tid < kIn * kIn) {
        const int r = tid / kIn, c = tid - r * kIn;
        xpad[(r + 1) * kInPad + c + 1] = v;
    }
}

struct LdsF {
    float x[2][kXSize];
    float a1[2][kA1Size];
};

__device__ __forceinline__ void zero_lds(float *p, int n, int tid)
{
    for (int i = tid; i < n; i += kThreads) p[i] = 0.0f;
}

// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 4) void k_stem_fwd(const float *__restrict__ x, const float *__restrict__ w1,
                                                     const float *__restrict__ b1, const float *__restrict__ w2,
                                                     const float *__restrict__ b2, float *__restrict__ y, long long M)
{
    __shared__ __attribute__((aligned(16))) LdsF s;
    const int tid = (int)threadIdx.x;
    zero_lds(&s.x[0][0], 2 * kXSize, tid);
    zero_lds(&s.a1[0][0], 2 * kA1Size, tid);
    const int ciq = tid & 3, oh = (tid >> 2) & 3, cp = tid >> 4;
    float W[2][36];
#pragma unroll
    for (int i = 0; i < 36; i++) {
        W[0][i] = w2[(2 * cp) * 144 + ciq * 36 + i];
        W[1][i] = w2[(2 * cp + 1) * 144 + ciq * 36 + i];
    }
    const float bias0 = b2[2 * cp], bias1 = b2[2 * cp + 1];
    float w1r[9];
#pragma unroll
    for (int k = 0; k < 9; k++) w1r[k] = w1[(tid >> 4) * 9 + k];
    const float b1r = b1[tid >> 4];
    __syncthreads();
    long long m = blockIdx.x;
    float xv = (m < M && tid < 169) ? x[m * 169 + tid] : 0.f;
    int buf = 0;
    for (; m < M; m += gridDim.x, buf ^= 1) {
        store_x(s.x[buf], xv, tid);
        __syncthreads();
        conv1_frame(s.x[buf], s.a1[buf], w1r, b1r, tid);
        const long long mn = m + gridDim.x;
        xv = (mn < M && tid < 169) ? x[mn * 169 + tid] : 0.f;     // prefetch the next frame under conv2
        __syncthreads();
        float acc[2][4];
#pragma unroll
        for (int j = 0; j < 4; j++) { acc[0][j] = 0.f; acc[1][j] = 0.f; }
        const float *a1 = s.a1[buf] + (ciq * 4) * kA1Ch + (2 * oh) * kA1Stride;
#pragma unroll
        for (int ci = 0; ci < 4; ci++) {
#pragma unroll
            for (int kh = 0; kh < 3; kh++) {
                const float4 *row = reinterpret_cast<const float4 *>(a1 + ci * kA1Ch + kh * kA1Stride);
                const float4 r0 = row[0], r1 = row[1], r2 = row[2];
                const float r[12] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w};
#pragma unroll
                for (int kw = 0; kw < 3; kw++) {
                    const float wa = W[0][ci * 9 + kh * 3 + kw], wb = W[1][ci * 9 + kh * 3 + kw];
#pragma unroll
                    for (int ow = 0; ow < 4; ow++) {
                        acc[0][ow] = fmaf(r[2 * ow + kw], wa, acc[0][ow]);
                        acc[1][ow] = fmaf(r[2 * ow + kw], wb, acc[1][ow]);
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) { acc[0][j] = quad_sum(acc[0][j]); acc[1][j] = quad_sum(acc[1][j]); }
        if (ciq < 2) {   // lane 0 of the quad stores channel 2cp, lane 1 channel 2cp+1
            const float b = ciq ? bias1 : bias0;
            const float4 v = ciq ? make_float4(acc[1][0], acc[1][1], acc[1][2], acc[1][3])
                                 : make_float4(acc[0][0], acc[0][1], acc[0][2], acc[0][3]);
            reinterpret_cast<float4 *>(y + m * 512)[(2 * cp + ciq) * 4 + oh] =
                make_float4(fmaxf(v.x + b, 0.f), fmaxf(v.y + b, 0.f), fmaxf(v.z + b, 0.f), fmaxf(v.w + b, 0.f));
        }
    }
}

// dW2 / db2: same thread roles as the forward; the 72 registers hold gradient accumulators instead of weights.
__global__ __launch_bounds__(256, 4) void k_stem_bwd_w2(const float *__restrict__ x, const float *__restrict__ y,
                                                        const float *__restrict__ dy, const float *__restrict__ w1,
                                                        const float *__restrict__ b1, float *__restrict__ partial, long long M)
{
    __shared__ __attribute__((aligned(16))) LdsF s;
    const int tid = (int)threadIdx.x;
    zero_lds(&s.x[0][0], 2 * kXSize, tid);
    zero_lds(&s.a1[0][0], 2 * kA1Size, tid);
    const int ciq = tid & 3, oh = (tid >> 2) & 3, cp = tid >> 4;
    float G[2][36];
#pragma unroll
    for (int i = 0; i < 36; i++) { G[0][i] = 0.f; G[1][i] = 0.f; }
    float gb0 = 0.f, gb1 = 0.f;
    float w1r[9];
#pragma unroll
    for (int k = 0; k < 9; k++) w1r[k] = w1[(tid >> 4) * 9 + k];
    const float b1r = b1[tid >> 4];
    __syncthreads();
    long long m = blockIdx.x;
    float xv = (m < M && tid < 169) ? x[m * 169 + tid] : 0.f;
    int buf = 0;
    for (; m < M; m += gridDim.x, buf ^= 1) {
        store_x(s.x[buf], xv, tid);
        const float4 *yy = reinterpret_cast<const float4 *>(y + m * 512), *dd = reinterpret_cast<const float4 *>(dy + m * 512);
        const float4 ya = yy[(2 * cp) * 4 + oh], yb = yy[(2 * cp + 1) * 4 + oh];
        const float4 da = dd[(2 * cp) * 4 + oh], db = dd[(2 * cp + 1) * 4 + oh];
        __syncthreads();
        conv1_frame(s.x[buf], s.a1[buf], w1r, b1r, tid);
        const long long mn = m + gridDim.x;
        xv = (mn < M && tid < 169) ? x[mn * 169 + tid] : 0.f;
        __syncthreads();
        const float dz[2][4] = {{ya.x > 0.f ? da.x : 0.f, ya.y > 0.f ? da.y : 0.f, ya.z > 0.f ? da.z : 0.f, ya.w > 0.f ? da.w : 0.f},
                                {yb.x > 0.f ? db.x : 0.f, yb.y > 0.f ? db.y : 0.f, yb.z > 0.f ? db.z : 0.f, yb.w > 0.f ? db.w : 0.f}};
        if (ciq == 0) {
            gb0 += dz[0][0] + dz[0][1] + dz[0][2] + dz[0][3];
            gb1 += dz[1][0] + dz[1][1] + dz[1][2] + dz[1][3];
        }
        const float *a1 = s.a1[buf] + (ciq * 4) * kA1Ch + (2 * oh) * kA1Stride;
#pragma unroll
        for (int ci = 0; ci < 4; ci++) {
#pragma unroll
            for (int kh = 0; kh < 3; kh++) {
                const float4 *row = reinterpret_cast<const float4 *>(a1 + ci * kA1Ch + kh * kA1Stride);
                const float4 r0 = row[0], r1 = row[1], r2 = row[2];
                const float r[12] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w};
#pragma unroll
                for (int kw = 0; kw < 3; kw++) {
                    float ga = G[0][ci * 9 + kh * 3 + kw], gb = G[1][ci * 9 + kh * 3 + kw];
#pragma unroll
                    for (int ow = 0; ow < 4; ow++) {
                        ga = fmaf(r[2 * ow + kw], dz[0][ow], ga);
                        gb = fmaf(r[2 * ow + kw], dz[1][ow], gb);
                    }
                    G[0][ci * 9 + kh * 3 + kw] = ga; G[1][ci * 9 + kh * 3 + kw] = gb;
                }
            }
        }
    }
    // sum over the 4 output rows (tid bits 2..3), then one record per workgroup
    float *rec = partial + (size_t)blockIdx.x * kPartial;
#pragma unroll
    for (int i = 0; i < 36; i++) {
        const float a = xor_sum(xor_sum(G[0][i], 4), 8), b = xor_sum(xor_sum(G[1][i], 4), 8);
        if (oh == 0) { rec[(2 * cp) * 144 + ciq * 36 + i] = a; rec[(2 * cp + 1) * 144 + ciq * 36 + i] = b; }
    }
    gb0 = xor_sum(xor_sum(gb0, 4), 8); gb1 = xor_sum(xor_sum(gb1, 4), 8);
    if (oh == 0 && ciq == 0) { rec[kW2 + 2 * cp] = gb0; rec[kW2 + 2 * cp + 1] = gb1; }
}

constexpr int kPwSize = kC1 * 4 * 3 * kA1Stride;   // per (ci, oh): its 3x9 window of a1-gradients, rows padded to 12
struct LdsB {
    float x[2][kXSize];
    float a1[kA1Size];
    float pw[2][kPwSize];    // double-buffered with dz: both are (re)written before the frame's first barrier,
    float dz[2][512];        // while slower threads may still be reading the previous frame's copies
};

// da1 = conv2^T(dz2) -> dz1 = da1 * (a1 > 0) -> dW1 / db1.
__global__ __launch_bounds__(256, 3) void k_stem_bwd_w1(const float *__restrict__ x, const float *__restrict__ y,
                                                        const float *__restrict__ dy, const float *__restrict__ w1,
                                                        const float *__restrict__ b1, const float *__restrict__ w2,
                                                        float *__restrict__ partial, long long M)
{
    __shared__ __attribute__((aligned(16))) LdsB s;
    const int tid = (int)threadIdx.x;
    zero_lds(&s.x[0][0], 2 * kXSize, tid);
    zero_lds(s.a1, kA1Size, tid);
    const int coq = tid & 3, oh = (tid >> 2) & 3, ci = tid >> 4;
    float Wt[8][9];   // w2[co][ci][k] for co in [8coq, 8coq+8)
#pragma unroll
    for (int c = 0; c < 8; c++)
#pragma unroll
        for (int k = 0; k < 9; k++) Wt[c][k] = w2[(8 * coq + c) * 144 + ci * 9 + k];
    float w1r[9];
#pragma unroll
    for (int k = 0; k < 9; k++) w1r[k] = w1[(tid >> 4) * 9 + k];
    const float b1r = b1[tid >> 4];
    float g1[9];
#pragma unroll
    for (int k = 0; k < 9; k++) g1[k] = 0.f;
    float gb = 0.f;
    __syncthreads();
    long long m = blockIdx.x;
    float xv = (m < M && tid < 169) ? x[m * 169 + tid] : 0.f;
    int buf = 0;
    for (; m < M; m += gridDim.x, buf ^= 1) {
        store_x(s.x[buf], xv, tid);
        if (tid < 128) {   // dz2 = dy * (y > 0) -> LDS [co][p]
            const float4 yv = reinterpret_cast<const float4 *>(y + m * 512)[tid], dv = reinterpret_cast<const float4 *>(dy + m * 512)[tid];
            reinterpret_cast<float4 *>(s.dz[buf])[tid] = make_float4(yv.x > 0.f ? dv.x : 0.f, yv.y > 0.f ? dv.y : 0.f,
                                                                  yv.z > 0.f ? dv.z : 0.f, yv.w > 0.f ? dv.w : 0.f);
        }
        __syncthreads();                       // every thread has left the previous frame: a1 may be rewritten
        conv1_frame(s.x[buf], s.a1, w1r, b1r, tid);
        const long long mn = m + gridDim.x;
        xv = (mn < M && tid < 169) ? x[mn * 169 + tid] : 0.f;
        // window of a1-gradients touched by conv2 output row oh: padded rows 2oh..2oh+2, padded cols 0..8
        float P[3][9];
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = 0; b < 9; b++) P[a][b] = 0.f;
        const float4 *dzr = reinterpret_cast<const float4 *>(s.dz[buf]);
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const float4 d4 = dzr[(8 * coq + c) * 4 + oh];
            const float d[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
            for (int kh = 0; kh < 3; kh++)
#pragma unroll
                for (int kw = 0; kw < 3; kw++) {
                    const float w = Wt[c][kh * 3 + kw];
#pragma unroll
                    for (int ow = 0; ow < 4; ow++) P[kh][2 * ow + kw] = fmaf(d[ow], w, P[kh][2 * ow + kw]);
                }
        }
        // each (ci, oh) owns its window in LDS (no atomics, no clearing); overlapping rows are summed by the reader
        float *pw = s.pw[buf] + (ci * 4 + oh) * 3 * kA1Stride;
#pragma unroll
        for (int kh = 0; kh < 3; kh++)
#pragma unroll
            for (int b = 0; b < 9; b++) {
                const float v = quad_sum(P[kh][b]);
                if (((kh * 9 + b) & 3) == coq) pw[kh * kA1Stride + b] = v;   // the quad shares the 27 stores
            }
        __syncthreads();
        {   // dz1 and conv1 weight gradients: thread = (channel c, part), positions q = part + 16 i
            const int c = tid >> 4, part = tid & 15;
            const float *a1c = s.a1 + c * kA1Ch, *pwc = s.pw[buf] + c * 4 * 3 * kA1Stride;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int q = part + 16 * i;
                if (q < 49) {
                    const int r = q / 7, cc = q - r * 7;
                    const float a = a1c[(r + 1) * kA1Stride + cc + 1];
                    // padded row R = r+1 = 2*oh + kh: odd R <- (oh=(R-1)/2, kh=1); even R <- (R/2-1, kh=2) and (R/2, kh=0)
                    const int R = r + 1;
                    float da;
                    if (R & 1) da = pwc[(((R - 1) >> 1) * 3 + 1) * kA1Stride + cc + 1];
                    else {
                        da = pwc[(((R >> 1) - 1) * 3 + 2) * kA1Stride + cc + 1];
                        if ((R >> 1) < 4) da += pwc[((R >> 1) * 3 + 0) * kA1Stride + cc + 1];
                    }
                    const float dzv = a > 0.f ? da : 0.f;
                    gb += dzv;
                    const float *xr = s.x[buf] + (2 * r) * kInPad + 2 * cc;
#pragma unroll
                    for (int kh = 0; kh < 3; kh++)
#pragma unroll
                        for (int kw = 0; kw < 3; kw++) g1[kh * 3 + kw] = fmaf(dzv, xr[kh * kInPad + kw], g1[kh * 3 + kw]);
                }
            }
        }
    }
    float *rec = partial + (size_t)blockIdx.x * kPartial + kW2 + kC2;
    const int c = tid >> 4, part = tid & 15;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const float v = xor_sum(xor_sum(xor_sum(xor_sum(g1[k], 1), 2), 4), 8);
        if (part == 0) rec[c * 9 + k] = v;
    }
    gb = xor_sum(xor_sum(xor_sum(xor_sum(gb, 1), 2), 4), 8);
    if (part == 0) rec[144 + c] = gb;
}

// Fixed-order reduction of the per-workgroup records: 16 record elements x 64 record slices per block.
__global__ __launch_bounds__(1024) void k_stem_reduce(const float *__restrict__ partial, int nrec, float *__restrict__ dw1,
                                                      float *__restrict__ db1, float *__restrict__ dw2, float *__restrict__ db2)
{
    __shared__ float red[64][17];
    const int jl = (int)(threadIdx.x & 15u), slice = (int)(threadIdx.x >> 4);
    const int j = (int)blockIdx.x * 16 + jl;
    float acc = 0.f;
    if (j < kPartial)
        for (int r = slice; r < nrec; r += 64) acc += partial[(size_t)r * kPartial + j];
    red[slice][jl] = acc;
    __syncthreads();
    if (slice == 0 && j < kPartial) {
        acc = 0.f;
        for (int q = 0; q < 64; q++) acc += red[q][jl];
        if (j < kW2) dw2[j] = acc;
        else if (j < kW2 + kC2) db2[j - kW2] = acc;
        else if (j < kW2 + kC2 + 144) dw1[j - kW2 - kC2] = acc;
        else db1[j - kW2 - kC2 - 144] = acc;
    }
}

} // namespace atr

using namespace atr;

static int stem_grid(long long M)
{
    static int cached_cus = 0;   // queried once: hipGetDeviceProperties is far too slow for a per-launch call
    if (cached_cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached_cus = n;
    }
    const int cus = cached_cus;
    const long long cap = (long long)cus * 4;   // 4 resident workgroups per CU
    return (int)(M < cap ? (M < 1 ? 1 : M) : cap);
}

extern "C" long long atr_stem_workspace_floats(long long M) { return (long long)stem_grid(M) * kPartial; }

extern "C" int atr_stem_forward(const float *x, const float *w1, const float *b1, const float *w2, const float *b2,
                                float *y, long long M, void *stream)
{
    if (!x || !w1 || !b1 || !w2 || !b2 || !y || M < 0) return -1;
    if (M == 0) return 0;
    hipLaunchKernelGGL(k_stem_fwd, dim3((unsigned)stem_grid(M)), dim3(kThreads), 0, (hipStream_t)stream, x, w1, b1, w2, b2, y, M);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int atr_stem_backward(const float *x, const float *y, const float *dy, const float *w1, const float *b1,
                                 const float *w2, float *dw1, float *db1, float *dw2, float *db2, float *workspace,
                                 long long M, void *stream)
{
    if (!x || !y || !dy || !w1 || !b1 || !w2 || !dw1 || !db1 || !dw2 || !db2 || !workspace || M <= 0) return -1;
    const int grid = stem_grid(M);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_stem_bwd_w2, dim3((unsigned)grid), dim3(kThreads), 0, st, x, y, dy, w1, b1, workspace, M);
    hipLaunchKernelGGL(k_stem_bwd_w1, dim3((unsigned)grid), dim3(kThreads), 0, st, x, y, dy, w1, b1, w2, workspace, M);
    hipLaunchKernelGGL(k_stem_reduce, dim3((kPartial + 15) / 16), dim3(1024), 0, st, workspace, grid, dw1, db1, dw2, db2);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
