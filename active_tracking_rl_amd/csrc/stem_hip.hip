// stem_hip.hip — fused conv stem of the maze policies (perception.py:68-92 of the reference:
// conv(1->16,k3,s2,p1) 13->7, ReLU, conv(16->32,k3,s2,p1) 7->4, ReLU) as hand-written HIP for gfx950.
//
// Why: per frame the stem is 0.16 MFLOP on 676 B of input. As library GEMMs (Toeplitz-expanded weights) it costs
// 6.6x the FLOPs and round-trips a 784-wide activation through HBM (0.5 GB per 163 840-frame batch, forward and
// backward); MIOpen launches one Im2Col kernel per sample. Here one wavefront owns one frame at a time, the
// intermediate activation lives in LDS and the conv2 weights (or their gradient accumulators) live in registers.
//
//   forward   x[M,169] -> y[M,512] (c,h,w order, post-ReLU)                         atr_stem_forward
//   backward  (x, y, dy) -> dW1[16,9], db1[16], dW2[32,144], db2[32]                atr_stem_backward
//             (conv1 is recomputed per frame; no gradient w.r.t. the observation is needed)
//
// Lane roles, conv2 side (forward and dW2): lane = (cp, oh) with cp = lane>>2 a pair of output channels
// {2cp, 2cp+1} and oh = lane&3 one output row (4 positions). Each 16-byte LDS read of an a1 row feeds
// 3 taps x 4 positions x 2 channels = 24 FMAs, so the loop is FMA-bound, not LDS-bound.
// Lane roles, da1 side: lane = (ci, oh): input channel ci = lane>>2 accumulates the 3x9 window of a1-gradients that
// conv2 output row oh touches, from all 32 output channels (weights w2[:, ci, :] in registers).
// fp32 FMA throughout (the reference computes in fp32); no MFMA: fp32 MFMA runs at the vector rate on gfx950.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/atr_policy.h"

namespace atr {

constexpr int kIn = 13, kInPad = 15;         // input side, padded by 1
constexpr int kC1 = 16, kH1 = 7;             // conv1 channels / output side
constexpr int kC2 = 32, kH2 = 4;             // conv2 channels / output side
constexpr int kA1Rows = 9, kA1Stride = 12;   // a1 padded to rows -1..7, cols -1..7 (+3 so a row is 3 x 16 B)
constexpr int kA1Ch = kA1Rows * kA1Stride;   // 108 floats per channel
constexpr int kA1Size = kC1 * kA1Ch;         // 1728 floats
constexpr int kXSize = 228;                  // 15*15 = 225, rounded to a multiple of 4
constexpr int kWaves = 4;
constexpr int kW2 = kC2 * kC1 * 9;           // 4608
constexpr int kPartial = kW2 + kC2 + kC1 * 9 + kC1;  // per-wave partial gradient record: 4800 floats

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// x[169] -> xpad (15x15, zero border) ; conv1 + ReLU -> a1pad[16][9][12] (zero border). w1s = w1[144] ++ b1[16] in LDS.
__device__ __forceinline__ void load_frame_and_conv1(const float *__restrict__ x, float *xpad, float *a1pad,
                                                     const float *w1s, int lane)
{
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const int j = lane + 64 * i;
        if (j < kIn * kIn) {
            const int r = j / kIn, c = j - r * kIn;
            xpad[(r + 1) * kInPad + c + 1] = x[j];
        }
    }
    wave_sync();
#pragma unroll
    for (int i = 0; i < 13; i++) {
        const int o = lane + 64 * i;
        if (o < kC1 * kH1 * kH1) {
            const int c = o / 49, q = o - c * 49;
            const int oh = q / kH1, ow = q - oh * kH1;
            float acc = w1s[144 + c];
            const float *xr = xpad + (2 * oh) * kInPad + 2 * ow;
            const float *w = w1s + c * 9;
#pragma unroll
            for (int kh = 0; kh < 3; kh++)
#pragma unroll
                for (int kw = 0; kw < 3; kw++) acc = fmaf(xr[kh * kInPad + kw], w[kh * 3 + kw], acc);
            a1pad[c * kA1Ch + (oh + 1) * kA1Stride + ow + 1] = fmaxf(acc, 0.0f);
        }
    }
    wave_sync();
}

__device__ __forceinline__ void zero_lds(float *p, int n, int lane)
{
    for (int i = lane; i < n; i += 64) p[i] = 0.0f;
}

struct Lds {
    float x[kWaves][kXSize];
    float a1[kWaves][kA1Size];
    float w1s[160];
};

// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 1) void k_stem_fwd(const float *__restrict__ x, const float *__restrict__ w1,
                                                     const float *__restrict__ b1, const float *__restrict__ w2,
                                                     const float *__restrict__ b2, float *__restrict__ y, long long M)
{
    __shared__ __attribute__((aligned(16))) Lds s;
    const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
    for (int i = (int)threadIdx.x; i < 160; i += (int)blockDim.x) s.w1s[i] = i < 144 ? w1[i] : b1[i - 144];
    zero_lds(s.x[wave], kXSize, lane);
    zero_lds(s.a1[wave], kA1Size, lane);
    __syncthreads();
    const int cp = lane >> 2, oh = lane & 3;
    float W[2][144];
#pragma unroll
    for (int i = 0; i < 144; i++) { W[0][i] = w2[(2 * cp) * 144 + i]; W[1][i] = w2[(2 * cp + 1) * 144 + i]; }
    const float bias0 = b2[2 * cp], bias1 = b2[2 * cp + 1];
    const long long wid = (long long)blockIdx.x * kWaves + wave, nw = (long long)gridDim.x * kWaves;
    for (long long m = wid; m < M; m += nw) {
        load_frame_and_conv1(x + m * 169, s.x[wave], s.a1[wave], s.w1s, lane);
        float acc[2][4];
#pragma unroll
        for (int j = 0; j < 4; j++) { acc[0][j] = bias0; acc[1][j] = bias1; }
        const float *a1 = s.a1[wave];
#pragma unroll
        for (int ci = 0; ci < kC1; ci++) {
#pragma unroll
            for (int kh = 0; kh < 3; kh++) {
                const float4 *row = reinterpret_cast<const float4 *>(a1 + ci * kA1Ch + (2 * oh + kh) * kA1Stride);
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
        float4 *yo = reinterpret_cast<float4 *>(y + m * 512);
        yo[(2 * cp) * 4 + oh] = make_float4(fmaxf(acc[0][0], 0.f), fmaxf(acc[0][1], 0.f), fmaxf(acc[0][2], 0.f), fmaxf(acc[0][3], 0.f));
        yo[(2 * cp + 1) * 4 + oh] = make_float4(fmaxf(acc[1][0], 0.f), fmaxf(acc[1][1], 0.f), fmaxf(acc[1][2], 0.f), fmaxf(acc[1][3], 0.f));
        wave_sync();   // a1/x of this wave are rewritten by the next frame
    }
}

// dW2 / db2: same lane roles as the forward; the 288 registers hold gradient accumulators instead of weights.
__global__ __launch_bounds__(256, 1) void k_stem_bwd_w2(const float *__restrict__ x, const float *__restrict__ y,
                                                        const float *__restrict__ dy, const float *__restrict__ w1,
                                                        const float *__restrict__ b1, float *__restrict__ partial, long long M)
{
    __shared__ __attribute__((aligned(16))) Lds s;
    const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
    for (int i = (int)threadIdx.x; i < 160; i += (int)blockDim.x) s.w1s[i] = i < 144 ? w1[i] : b1[i - 144];
    zero_lds(s.x[wave], kXSize, lane);
    zero_lds(s.a1[wave], kA1Size, lane);
    __syncthreads();
    const int cp = lane >> 2, oh = lane & 3;
    float G[2][144];
#pragma unroll
    for (int i = 0; i < 144; i++) { G[0][i] = 0.f; G[1][i] = 0.f; }
    float gb0 = 0.f, gb1 = 0.f;
    const long long wid = (long long)blockIdx.x * kWaves + wave, nw = (long long)gridDim.x * kWaves;
    for (long long m = wid; m < M; m += nw) {
        load_frame_and_conv1(x + m * 169, s.x[wave], s.a1[wave], s.w1s, lane);
        const float4 *yy = reinterpret_cast<const float4 *>(y + m * 512), *dd = reinterpret_cast<const float4 *>(dy + m * 512);
        const float4 ya = yy[(2 * cp) * 4 + oh], yb = yy[(2 * cp + 1) * 4 + oh];
        const float4 da = dd[(2 * cp) * 4 + oh], db = dd[(2 * cp + 1) * 4 + oh];
        const float dz[2][4] = {{ya.x > 0.f ? da.x : 0.f, ya.y > 0.f ? da.y : 0.f, ya.z > 0.f ? da.z : 0.f, ya.w > 0.f ? da.w : 0.f},
                                {yb.x > 0.f ? db.x : 0.f, yb.y > 0.f ? db.y : 0.f, yb.z > 0.f ? db.z : 0.f, yb.w > 0.f ? db.w : 0.f}};
        gb0 += dz[0][0] + dz[0][1] + dz[0][2] + dz[0][3];
        gb1 += dz[1][0] + dz[1][1] + dz[1][2] + dz[1][3];
        const float *a1 = s.a1[wave];
#pragma unroll
        for (int ci = 0; ci < kC1; ci++) {
#pragma unroll
            for (int kh = 0; kh < 3; kh++) {
                const float4 *row = reinterpret_cast<const float4 *>(a1 + ci * kA1Ch + (2 * oh + kh) * kA1Stride);
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
        wave_sync();
    }
    // sum the 4 output rows (lanes of a quad), then one record per wave
    float *rec = partial + wid * kPartial;
#pragma unroll
    for (int i = 0; i < 144; i++) {
        float a = G[0][i], b = G[1][i];
        a += __shfl_xor(a, 1, 64); a += __shfl_xor(a, 2, 64);
        b += __shfl_xor(b, 1, 64); b += __shfl_xor(b, 2, 64);
        if (oh == 0) { rec[(2 * cp) * 144 + i] = a; rec[(2 * cp + 1) * 144 + i] = b; }
    }
    gb0 += __shfl_xor(gb0, 1, 64); gb0 += __shfl_xor(gb0, 2, 64);
    gb1 += __shfl_xor(gb1, 1, 64); gb1 += __shfl_xor(gb1, 2, 64);
    if (oh == 0) { rec[kW2 + 2 * cp] = gb0; rec[kW2 + 2 * cp + 1] = gb1; }
}

constexpr int kWavesB = 2;   // 2 waves per block here: x + a1 + da1 + dz is 16.8 KB of LDS per wave
struct LdsB {
    float x[kWavesB][kXSize];
    float a1[kWavesB][kA1Size];
    float da1[kWavesB][kA1Size];
    float dz[kWavesB][512];
    float w1s[160];
};

// da1 = conv2^T(dz2) -> dz1 = da1 * (a1 > 0) -> dW1 / db1.
__global__ __launch_bounds__(128, 1) void k_stem_bwd_w1(const float *__restrict__ x, const float *__restrict__ y,
                                                        const float *__restrict__ dy, const float *__restrict__ w1,
                                                        const float *__restrict__ b1, const float *__restrict__ w2,
                                                        float *__restrict__ partial, long long M)
{
    __shared__ __attribute__((aligned(16))) LdsB s;
    const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
    for (int i = (int)threadIdx.x; i < 160; i += (int)blockDim.x) s.w1s[i] = i < 144 ? w1[i] : b1[i - 144];
    zero_lds(s.x[wave], kXSize, lane);
    zero_lds(s.a1[wave], kA1Size, lane);
    __syncthreads();
    const int ci = lane >> 2, oh = lane & 3;
    float Wt[kC2][9];   // w2[co][ci][k] for this lane's input channel
#pragma unroll
    for (int co = 0; co < kC2; co++)
#pragma unroll
        for (int k = 0; k < 9; k++) Wt[co][k] = w2[co * 144 + ci * 9 + k];
    float g1[9];
#pragma unroll
    for (int k = 0; k < 9; k++) g1[k] = 0.f;
    float gb = 0.f;
    const long long wid = (long long)blockIdx.x * kWavesB + wave, nw = (long long)gridDim.x * kWavesB;
    for (long long m = wid; m < M; m += nw) {
        load_frame_and_conv1(x + m * 169, s.x[wave], s.a1[wave], s.w1s, lane);
        // dz2 = dy * (y > 0) -> LDS [co][p]
        {
            const float4 *yy = reinterpret_cast<const float4 *>(y + m * 512), *dd = reinterpret_cast<const float4 *>(dy + m * 512);
            float4 *dst = reinterpret_cast<float4 *>(s.dz[wave]);
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const float4 yv = yy[lane + 64 * i], dv = dd[lane + 64 * i];
                dst[lane + 64 * i] = make_float4(yv.x > 0.f ? dv.x : 0.f, yv.y > 0.f ? dv.y : 0.f, yv.z > 0.f ? dv.z : 0.f, yv.w > 0.f ? dv.w : 0.f);
            }
        }
        zero_lds(s.da1[wave], kA1Size, lane);
        wave_sync();
        // window of a1-gradients touched by conv2 output row oh: padded rows 2oh..2oh+2, padded cols 0..8
        float P[3][9];
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = 0; b < 9; b++) P[a][b] = 0.f;
        const float4 *dzr = reinterpret_cast<const float4 *>(s.dz[wave]);
#pragma unroll
        for (int co = 0; co < kC2; co++) {
            const float4 d4 = dzr[co * 4 + oh];
            const float d[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
            for (int kh = 0; kh < 3; kh++)
#pragma unroll
                for (int kw = 0; kw < 3; kw++) {
                    const float w = Wt[co][kh * 3 + kw];
#pragma unroll
                    for (int ow = 0; ow < 4; ow++) P[kh][2 * ow + kw] = fmaf(d[ow], w, P[kh][2 * ow + kw]);
                }
        }
        float *da1 = s.da1[wave] + ci * kA1Ch;
#pragma unroll
        for (int kh = 0; kh < 3; kh++)
#pragma unroll
            for (int b = 0; b < 9; b++) atomicAdd(&da1[(2 * oh + kh) * kA1Stride + b], P[kh][b]);
        wave_sync();
        // dz1 and conv1 weight gradients: lane = (channel c, part): positions q = part, part+4, ...
        {
            const int c = lane >> 2, part = lane & 3;
            const float *a1c = s.a1[wave] + c * kA1Ch, *dac = s.da1[wave] + c * kA1Ch;
            for (int q = part; q < 49; q += 4) {
                const int r = q / 7, cc = q - r * 7;
                const float a = a1c[(r + 1) * kA1Stride + cc + 1];
                const float dzv = a > 0.f ? dac[(r + 1) * kA1Stride + cc + 1] : 0.f;
                gb += dzv;
                const float *xr = s.x[wave] + (2 * r) * kInPad + 2 * cc;
#pragma unroll
                for (int kh = 0; kh < 3; kh++)
#pragma unroll
                    for (int kw = 0; kw < 3; kw++) g1[kh * 3 + kw] = fmaf(dzv, xr[kh * kInPad + kw], g1[kh * 3 + kw]);
            }
        }
        wave_sync();
    }
    float *rec = partial + wid * kPartial + kW2 + kC2;
    const int c = lane >> 2, part = lane & 3;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        float v = g1[k];
        v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64);
        if (part == 0) rec[c * 9 + k] = v;
    }
    gb += __shfl_xor(gb, 1, 64); gb += __shfl_xor(gb, 2, 64);
    if (part == 0) rec[144 + c] = gb;
}

// Deterministic tree-less reduction of the per-wave records: element j of the 4800-float record summed over waves.
__global__ void k_stem_reduce(const float *__restrict__ partial, int nrec, float *__restrict__ dw1, float *__restrict__ db1,
                              float *__restrict__ dw2, float *__restrict__ db2)
{
    const int j = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (j >= kPartial) return;
    float acc = 0.f;
    for (int r = 0; r < nrec; r++) acc += partial[(size_t)r * kPartial + j];
    if (j < kW2) dw2[j] = acc;
    else if (j < kW2 + kC2) db2[j - kW2] = acc;
    else if (j < kW2 + kC2 + 144) dw1[j - kW2 - kC2] = acc;
    else db1[j - kW2 - kC2 - 144] = acc;
}

} // namespace atr

using namespace atr;

static int stem_grid(long long M)
{
    int dev = 0, cus = 256;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount;
    long long need = (M + kWaves - 1) / kWaves;
    return (int)(need < cus ? (need < 1 ? 1 : need) : cus);
}

extern "C" long long atr_stem_workspace_floats(long long M) { return (long long)stem_grid(M) * kWaves * kPartial; }

extern "C" int atr_stem_forward(const float *x, const float *w1, const float *b1, const float *w2, const float *b2,
                                float *y, long long M, void *stream)
{
    if (!x || !w1 || !b1 || !w2 || !b2 || !y || M < 0) return -1;
    if (M == 0) return 0;
    hipLaunchKernelGGL(k_stem_fwd, dim3((unsigned)stem_grid(M)), dim3(256), 0, (hipStream_t)stream, x, w1, b1, w2, b2, y, M);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int atr_stem_backward(const float *x, const float *y, const float *dy, const float *w1, const float *b1,
                                 const float *w2, float *dw1, float *db1, float *dw2, float *db2, float *workspace,
                                 long long M, void *stream)
{
    if (!x || !y || !dy || !w1 || !b1 || !w2 || !dw1 || !db1 || !dw2 || !db2 || !workspace || M <= 0) return -1;
    const int grid = stem_grid(M);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_stem_bwd_w2, dim3((unsigned)grid), dim3(256), 0, st, x, y, dy, w1, b1, workspace, M);
    hipLaunchKernelGGL(k_stem_bwd_w1, dim3((unsigned)(grid * kWaves / kWavesB)), dim3(64 * kWavesB), 0, st, x, y, dy, w1, b1, w2, workspace, M);
    hipLaunchKernelGGL(k_stem_reduce, dim3((kPartial + 255) / 256), dim3(256), 0, st, workspace, grid * kWaves, dw1, db1, dw2, db2);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
