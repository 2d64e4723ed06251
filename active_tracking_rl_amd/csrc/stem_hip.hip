// stem_hip.hip — fused conv stem of the maze policies (perception.py:68-92 of the reference:
// conv(1->16,k3,s2,p1) 13->7, ReLU, conv(16->32,k3,s2,p1) 7->4, ReLU) as hand-written HIP for gfx950.
//
// Why: per frame the stem is 0.16 MFLOP on 676 B of input. As library GEMMs (Toeplitz-expanded weights) it costs
// 6.6x the FLOPs and round-trips a 784-wide activation through HBM; MIOpen launches one Im2Col kernel per sample.
// Here ONE WAVEFRONT owns one frame at a time and never leaves the CU with it:
//   conv1 (14 kFLOP)  packed-f32 VALU; lane = (channel pair, output row); the zero-bordered activation a1 stays in
//                     the wave's own 7 KB of LDS.
//   conv2 (147 kFLOP) the f32 matrix cores: v_mfma_f32_16x16x4_f32 (exact f32, an fmaf chain) on the implicit
//                     im2col of a1 — per frame a [16 positions] x [144 taps] x [32 channels] product = 72 MFMAs whose
//                     A operand is ONE ds_read_b32 per lane (lane base + immediate offset: the tap order is chosen
//                     so that the 4 k-values of a step are 4 input channels) and whose B operand (the conv2
//                     weights) sits in 72 VGPRs for the whole kernel.
// The backward runs the same way in ONE pass over (x, y, dy): dW2 = dz2^T x im2col(a1) (72 MFMAs per frame into 72
// accumulator VGPRs) and da1 = dz2 x W2 (72 MFMAs, col2im done in registers with one cross-lane shift), conv1
// recomputed per frame; no gradient w.r.t. the observation is needed. Waves are independent (wave-local LDS, no workgroup barriers in the
// frame loop), so on each SIMD the MFMA pipe of one wave overlaps the VALU/LDS phases of the others.
//
//   forward   x[M,169] -> y[M,512] (c,h,w order, post-ReLU)                         atr_stem_forward
//   backward  (x, y, dy) -> dW1[16,9], db1[16], dW2[32,144], db2[32]                atr_stem_backward
//
// MFMA operand maps (16x16x4 f32): lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15]; it receives
// D[i = 4*(l>>4) + r][j = l&15] in accumulator register r.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/atr_policy.h"

namespace atr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kXS = 16, kXRows = 17;          // zero-bordered 15x15 input, rows padded to 16 floats (+2 zero rows)
constexpr int kXSize = kXS * kXRows;          // 272
constexpr int kC1 = 16, kC2 = 32;
constexpr int kA1Row = 12, kA1Ch = 109;       // a1 zero-bordered to 9x9 (rows of 12 floats); odd channel stride
constexpr int kA1Size = kC1 * kA1Ch;          // 1744
constexpr int kW2 = kC2 * kC1 * 9;            // 4608
constexpr int kPartial = kW2 + kC2 + kC1 * 9 + kC1;  // per-workgroup partial gradient record: 4800 floats
constexpr int kWaves = 4, kThreads = 64 * kWaves;

struct LdsF { float x[kXSize]; float a1[kA1Size]; };                   // 8064 B per wave
struct LdsB { float x[kXSize]; float a1[kA1Size]; float dz[512]; };    // 10112 B per wave

__device__ __forceinline__ void wave_lds_sync()
{
    // DS operations of one wave execute in order; this only stops the compiler from moving LDS accesses across
    // the point, so that values written by other lanes of the wave are re-read.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Workgroup barrier that orders LDS traffic only. __syncthreads() is a workgroup-scope fence over ALL memory: it drains vmcnt
// too, i.e. every wave would sit at the barrier until the output stores it has just issued are acknowledged by the L2 and the
// next pass's prefetched frames have arrived (measured in k_stem_fwd16: 40 % of the kernel). The tiles the waves hand each
// other live in LDS; global memory is not shared inside the kernel.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ float xor_sum(float v, int mask) { return v + __shfl_xor(v, mask, 64); }
__device__ __forceinline__ f32x4 mfma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// A frame's 169 floats: lane l holds elements l, l+64, l+128.
// XT = float, or uint8_t: the env's u8 observations (t2d_step_u8, values 0/1/2/4) decoded here, in the first layer
// (the np.float32(obs) cast of frame_stack, environment.py:138,146, fused into conv1: SURVEY 8f rank 1).
// A frame is fetched one loop iteration ahead of its use, so XRegs carries the RAW loaded words across the loop
// back-edge and the decode happens in store_x: converting at load time would make the prefetch a blocking load.
// Bytes are fetched as the aligned dword that holds them (never leaves the byte's own 4-byte granule): lanes l, l+64
// and l+128 of a frame share the same byte offset within their dwords.
// The loads are branch-free GLOBAL loads (the frame index clamped to the last frame, the third element's lane clamped to 40:
// what a lane beyond the data fetches is never decoded): a load under a branch, or a flat one (what a pointer that travelled
// through a kernel-argument struct becomes), makes the compiler drain vmcnt to 0 at the loop head — and with it the
// PREVIOUS frame's output stores — instead of waiting for just these three words.
struct XRegs { uint32_t w[3]; uint32_t sh; };
typedef const uint32_t __attribute__((address_space(1))) *stem_gwords;
template <typename XT>
__device__ __forceinline__ XRegs load_x(const XT *__restrict__ x, long long m, long long M, long long xs, int l)
{
    XRegs r;
    const long long mm = m < M ? m : M - 1;
    const int l2 = l < 41 ? l : 40;
    if (sizeof(XT) == 4) {
        stem_gwords p = (stem_gwords) reinterpret_cast<const uint32_t *>(x + mm * xs);
        r.sh = 0u;
        r.w[0] = p[l]; r.w[1] = p[l + 64]; r.w[2] = p[l2 + 128];
    } else {
        const uintptr_t a = reinterpret_cast<uintptr_t>(x + mm * xs) + (uintptr_t)l;
        stem_gwords p = (stem_gwords) reinterpret_cast<const uint32_t *>(a & ~(uintptr_t)3);
        const uintptr_t a2 = reinterpret_cast<uintptr_t>(x + mm * xs) + (uintptr_t)(l2 + 128);
        stem_gwords p2 = (stem_gwords) reinterpret_cast<const uint32_t *>(a2 & ~(uintptr_t)3);
        r.sh = 8u * (uint32_t)(a & 3u);
        r.w[0] = p[0]; r.w[1] = p[16]; r.w[2] = p2[0];
    }
    return r;
}
template <typename XT> __device__ __forceinline__ float decode_x(uint32_t w, uint32_t sh)
{
    return sizeof(XT) == 4 ? __uint_as_float(w) : (float)((w >> sh) & 0xffu);
}
__device__ __forceinline__ int xpad_addr(int i) { const int r = i / 13; return (r + 1) * kXS + (i - r * 13) + 1; }
template <typename XT>
__device__ __forceinline__ void store_x(float *xpad, const XRegs &r, int l)
{
    xpad[xpad_addr(l)] = decode_x<XT>(r.w[0], r.sh);
    xpad[xpad_addr(l + 64)] = decode_x<XT>(r.w[1], r.sh);
    if (l < 41) xpad[xpad_addr(l + 128)] = decode_x<XT>(r.w[2], r.sh);
}

// conv1 + ReLU of the wave's frame on packed f32 FMAs (v_pk_fma_f32: two channels per lane). Lane = (channel pair
// cp = l&7 -> channels 2cp, 2cp+1; output row oh = l>>3). Branch-free: the non-existent row oh = 7 reads the zero rows
// below the frame and writes zeros into a1's zero border. MFMA and VALU issue share the SIMD's f32 lanes (measured:
// their times add), so conv1's instruction count matters even next to 72 MFMAs.
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct Conv1W { f32x2 w[9]; f32x2 b; };
__device__ __forceinline__ Conv1W load_conv1_w(const float *__restrict__ w1, const float *__restrict__ b1, int l)
{
    Conv1W r;
    const int cp = l & 7;
#pragma unroll
    for (int k = 0; k < 9; k++) r.w[k] = f32x2{w1[(2 * cp) * 9 + k], w1[(2 * cp + 1) * 9 + k]};
    r.b = f32x2{b1[2 * cp], b1[2 * cp + 1]};
    return r;
}
__device__ __forceinline__ void conv1_wave(const float *xpad, float *a1, const Conv1W &cw, int l)
{
    const int cp = l & 7, oh = l >> 3;
    f32x2 acc[7];
#pragma unroll
    for (int ow = 0; ow < 7; ow++) acc[ow] = cw.b;
#pragma unroll
    for (int kh = 0; kh < 3; kh++) {
        const float4 *row = reinterpret_cast<const float4 *>(xpad + (2 * oh + kh) * kXS);
        const float4 r0 = row[0], r1 = row[1], r2 = row[2], r3 = row[3];
        const float r[16] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w, r3.x, r3.y, r3.z, r3.w};
#pragma unroll
        for (int kw = 0; kw < 3; kw++)
#pragma unroll
            for (int ow = 0; ow < 7; ow++)
                acc[ow] = __builtin_elementwise_fma(f32x2{r[2 * ow + kw], r[2 * ow + kw]}, cw.w[kh * 3 + kw], acc[ow]);
    }
    float *o0 = a1 + (2 * cp) * kA1Ch + (oh + 1) * kA1Row + 1, *o1 = o0 + kA1Ch;
    const bool live = oh < 7;
#pragma unroll
    for (int ow = 0; ow < 7; ow++) {
        o0[ow] = live ? fmaxf(acc[ow].x, 0.0f) : 0.0f;
        o1[ow] = live ? fmaxf(acc[ow].y, 0.0f) : 0.0f;
    }
}

__device__ __forceinline__ void zero_wave(float *p, int n, int l)
{
    for (int i = l; i < n; i += 64) p[i] = 0.0f;
}

// The workgroup's copy of the conv2 weights into LDS rows of 145 floats: 18 loads per thread, all in flight before the first
// is parked (one memory latency, not one per trip of a loop).
__device__ __forceinline__ void stage_w2(float *wst, const float *__restrict__ w2, int tid)
{
    static_assert(kW2 == 18 * kThreads, "18 elements per thread");
    const float __attribute__((address_space(1))) *g = (const float __attribute__((address_space(1))) *)w2;
    float v[18];
#pragma unroll
    for (int k = 0; k < 18; k++) v[k] = g[tid + kThreads * k];
#pragma unroll
    for (int k = 0; k < 18; k++) {
        const int i = tid + kThreads * k;
        wst[(i / 144) * 145 + (i % 144)] = v[k];
    }
}

// ------------------------------------------------------------------------------------------------------
// forward: D[pos][co] = sum_k im2col(a1)[pos][k] * W2[co][k]; step s = t*4 + cq covers tap t = kh*3+kw of the
// input channels 4cq..4cq+3 (k within the step = ci & 3).
struct StemProblem {
    const void *x;
    const float *w1, *b1, *w2, *b2;
    float *y;
    long long M, xs;
};
// Up to two independent problems per launch (the rollout's tracker and target encoders: different weights, same
// step): workgroups [0, split) work on p[0], the rest on p[1], so the chip is filled by one launch.
struct StemPair { StemProblem p[2]; int split, cus; };

template <typename XT>
__global__ __launch_bounds__(kThreads, 3) void k_stem_fwd(StemPair pr)
{
    __shared__ __attribute__((aligned(16))) LdsF lds[kWaves];
    const int l = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    LdsF &s = lds[wave];
    const bool second = (int)blockIdx.x >= pr.split;
    const StemProblem &pb = pr.p[second ? 1 : 0];
    const XT *__restrict__ x = reinterpret_cast<const XT *>(pb.x);
    const float *__restrict__ w1 = pb.w1, *__restrict__ b1 = pb.b1, *__restrict__ w2 = pb.w2, *__restrict__ b2 = pb.b2;
    float *__restrict__ y = pb.y;
    const long long M = pb.M, xs = pb.xs;
    const int blk = second ? (int)blockIdx.x - pr.split : (int)blockIdx.x;
    const int nblk = second ? (int)gridDim.x - pr.split : pr.split;
    const long long stride = (long long)nblk * kWaves;
    long long m = (long long)blk * kWaves + wave;
    XRegs xv = load_x(x, m, M, xs, l);
    const int c = l & 15, q = l >> 4;
    // The conv2 weights in MFMA operand order: every lane wants 72 of the 4608, its output channel's row strided by taps — fetched
    // straight from memory that is 64 cache lines per wave instruction, 72 times per wave, 12 waves per CU through one L1: at the
    // rollout's launch sizes (2-3 frames per wave) that gather was most of the kernel. So the workgroup copies the 18 KB once,
    // coalesced, into the (not yet used) frame buffers — rows padded to 145 floats: (17 c + 9 q) mod 64 makes the gather at most a
    // 2-way bank conflict — and the lanes pick their operands from there.
    float W[2][36];
    {
        float *wst = reinterpret_cast<float *>(lds);            // 32 x 145 floats = 18.1 KB of the workgroup's 31.5 KB
        stage_w2(wst, w2, (int)threadIdx.x);
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 9; t++)
#pragma unroll
            for (int cq = 0; cq < 4; cq++) {
                W[0][t * 4 + cq] = wst[c * 145 + (4 * cq + q) * 9 + t];
                W[1][t * 4 + cq] = wst[(16 + c) * 145 + (4 * cq + q) * 9 + t];
            }
        __syncthreads();
    }
    zero_wave(s.x, kXSize, l);
    zero_wave(s.a1, kA1Size, l);
    const float bias0 = b2[c], bias1 = b2[16 + c];
    const Conv1W cw = load_conv1_w(w1, b1, l);
    // A operand: position p = l&15 (oh = p>>2, ow = p&3), channel-in-step q
    const float *ap = s.a1 + q * kA1Ch + (2 * (c >> 2)) * kA1Row + 2 * (c & 3);
    wave_lds_sync();
    for (; m < M; m += stride) {
        store_x<XT>(s.x, xv, l);
        xv = load_x(x, m + stride, M, xs, l);        // prefetch the next frame under this one's math
        wave_lds_sync();
        conv1_wave(s.x, s.a1, cw, l);
        wave_lds_sync();
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; t++)
#pragma unroll
            for (int cq = 0; cq < 4; cq++) {
                const float a = ap[cq * 4 * kA1Ch + (t / 3) * kA1Row + (t % 3)];
                acc0 = mfma(a, W[0][t * 4 + cq], acc0);
                acc1 = mfma(a, W[1][t * 4 + cq], acc1);
            }
        float4 *yo = reinterpret_cast<float4 *>(y + m * 512);
        yo[c * 4 + q] = make_float4(fmaxf(acc0[0] + bias0, 0.f), fmaxf(acc0[1] + bias0, 0.f),
                                    fmaxf(acc0[2] + bias0, 0.f), fmaxf(acc0[3] + bias0, 0.f));
        yo[(16 + c) * 4 + q] = make_float4(fmaxf(acc1[0] + bias1, 0.f), fmaxf(acc1[1] + bias1, 0.f),
                                           fmaxf(acc1[2] + bias1, 0.f), fmaxf(acc1[3] + bias1, 0.f));
    }
}

// ------------------------------------------------------------------------------------------------------
// forward, 16 frames per workgroup pass (k_stem_fwd16): the launch sizes from 16384 frames up.
//
// k_stem_fwd above puts the 16 output positions of ONE frame on the MFMA's 16 rows — and 44 of the 144 (position, tap)
// pairs of a 4x4 output over a zero-bordered 7x7 input are structural zeros (output row 0 never sees tap row 0, ...): 31 % of
// the 72 MFMAs per frame multiply zeros, and which rows are zero changes from tap to tap, so no MFMA can be dropped.
// Here the rows are 16 FRAMES at one fixed output position: D_p[frame][co] = sum over the taps that are REAL for p, and the
// border products are simply not issued — 100 (position, tap) pairs x 4 channel groups x 2 column halves = 800 MFMAs per 16
// frames = 50 per frame instead of 72. The workgroup's 4 waves each own two output rows x one half of the output channels (50
// pairs x 4 groups = 200 MFMAs each — equal work — and a lane ends up with 32 contiguous bytes of y per frame: the first
// version split the output into 2x2 quadrants, stored 8-byte pieces, and spent 40 % of its time on them); a1 of the 16 frames
// lies in LDS frame-innermost ([ci][ih][iw][frame]: the A operand of lane (frame = l & 15, k = l >> 4) is one conflict-free
// ds_read_b32 at an immediate offset), un-bordered.
// conv1 is one lane per (frame, channel pair, half of the positions) on packed f32 FMAs, border taps skipped at compile time too.
// Every frame's arithmetic is the one of k_stem_fwd in the same order (taps ascending, channel groups inside; products with
// border zeros left out change no sum), so the two kernels agree bit for bit — a frame's result does not depend on the launch
// size it was part of.
constexpr int kF = 16;                          // frames per pass
constexpr int kXF = 212;                        // x tile: 13 rows of 16 floats per frame + 4: frame f starts at bank 20 f mod 64
constexpr int kA1F = kC1 * 49 * kF;             // a1 tile: 12544 floats
constexpr int kYC = 20, kYF = 32 * kYC;        // output tile (aliased on a1): 16 positions + 4 per channel: conflict-free 16-byte writes
struct LdsF16 { float x[kF * kXF]; float a1[kA1F]; };      // 13568 + 50176 B = 62.25 KB: two workgroups per CU
static_assert(kF * kYF <= kA1F, "the output tile must fit over a1");

// One thread fetches elements j = (tid & 15) + 16 k (k = 0..10) of frame tid >> 4 of the pass: 16 lanes sweep a frame.
struct XTile { uint32_t v[11]; };
template <typename XT>
__device__ __forceinline__ XTile load_xtile(const XT *__restrict__ x, long long m0, long long M, long long xs, int tid)
{
    XTile r;
    long long m = m0 + (tid >> 4);
    m = m < M ? m : M - 1;
    const XT __attribute__((address_space(1))) *p = (const XT __attribute__((address_space(1))) *)(x + m * xs);
#pragma unroll
    for (int k = 0; k < 11; k++) {
        int j = (tid & 15) + 16 * k;
        j = j < 169 ? j : 168;
        if (sizeof(XT) == 4) r.v[k] = __float_as_uint((float)p[j]);
        else r.v[k] = (uint32_t)p[j];
    }
    return r;
}
template <typename XT>
__device__ __forceinline__ void store_xtile(float *xt, const XTile &r, int tid)
{
    float *dst = xt + (tid >> 4) * kXF;
#pragma unroll
    for (int k = 0; k < 11; k++) {
        const int j = (tid & 15) + 16 * k;
        const int row = (j * 79) >> 10;             // j / 13 for j < 176
        const float v = sizeof(XT) == 4 ? __uint_as_float(r.v[k]) : (float)r.v[k];
        if (j < 169) dst[row * 16 + (j - row * 13)] = v;
    }
}

// conv1 + ReLU of frame f (this lane's), channel ch -> a1[(ch * 49 + r * 7 + c) * 16 + f]. Output row r reads input rows
// 2r - 1 .. 2r + 1 (row -1 and row 13 are the zero border: their taps are not issued); the two new rows of output row r + 1 are
// requested from LDS before row r is computed.
__device__ __forceinline__ void load_row13(const float *xf, int ir, float (&t)[13])
{
    const float4 *row = reinterpret_cast<const float4 *>(xf + ir * 16);
    const float4 r0 = row[0], r1 = row[1], r2 = row[2];
    const float r3 = xf[ir * 16 + 12];
    t[0] = r0.x; t[1] = r0.y; t[2] = r0.z; t[3] = r0.w; t[4] = r1.x; t[5] = r1.y; t[6] = r1.z; t[7] = r1.w;
    t[8] = r2.x; t[9] = r2.y; t[10] = r2.z; t[11] = r2.w; t[12] = r3;
}
// On packed FMAs (v_pk_fma_f32: both halves are plain f32 FMAs): a lane takes TWO channels of its frame and half of the 49 output
// positions (RH 0: rows 0-2 and row 3's columns 0-3 = 25; RH 1: row 3's columns 4-6 and rows 4-6 = 24; RH is wave-uniform) —
// 193 / 187 packed FMAs; one channel and all 49 positions per lane were 380 plain ones. It matters because conv1 runs beside the
// CU's other workgroup's conv2, and VALU and MFMA issue on a SIMD add up (rollout launch 20.5 -> 19.8 us, 163840 frames 182 -> 175 us).
template <int RH>
__device__ __forceinline__ void conv1_lane2(const float *xf, float *o0, float *o1, const f32x2 (&w)[9], f32x2 b)
{
    constexpr int r_lo = RH ? 3 : 0, r_hi = RH ? 6 : 3;
    float R[3][13], Nx[2][13];
    if (2 * r_lo - 1 >= 0) load_row13(xf, 2 * r_lo - 1, R[0]);
    load_row13(xf, 2 * r_lo, R[1]);
    load_row13(xf, 2 * r_lo + 1, R[2]);
#pragma unroll
    for (int r = r_lo; r <= r_hi; r++) {
        if (r < r_hi) {
            load_row13(xf, 2 * r + 2, Nx[0]);
            if (2 * r + 3 <= 12) load_row13(xf, 2 * r + 3, Nx[1]);
        }
        const int c_lo = (r == 3 && RH) ? 4 : 0, c_hi = (r == 3 && !RH) ? 3 : 6;
        f32x2 acc[7];
#pragma unroll
        for (int c = 0; c < 7; c++) acc[c] = b;
#pragma unroll
        for (int kh = 0; kh < 3; kh++) {
            if (2 * r - 1 + kh < 0 || 2 * r - 1 + kh > 12) continue;
#pragma unroll
            for (int kw = 0; kw < 3; kw++)
#pragma unroll
                for (int c = 0; c < 7; c++) {
                    const int ic = 2 * c - 1 + kw;
                    if (c < c_lo || c > c_hi || ic < 0 || ic > 12) continue;
                    acc[c] = __builtin_elementwise_fma(f32x2{R[kh][ic], R[kh][ic]}, w[kh * 3 + kw], acc[c]);
                }
        }
#pragma unroll
        for (int c = 0; c < 7; c++) {
            if (c < c_lo || c > c_hi) continue;
            o0[(r * 7 + c) * kF] = fmaxf(acc[c].x, 0.0f);
            o1[(r * 7 + c) * kF] = fmaxf(acc[c].y, 0.0f);
        }
#pragma unroll
        for (int j = 0; j < 13; j++) { R[0][j] = R[2][j]; R[1][j] = Nx[0][j]; R[2][j] = Nx[1][j]; }
    }
}

// conv2 of output rows 2 QA, 2 QA + 1 (8 positions) x one half of the output channels for the pass's 16 frames: D[frame][co]
// per position. ap = a1 + q * 49 * 16 + frame (this lane's A-operand base); Wh = the half's weights in B-operand order.
template <int QA>
__device__ __forceinline__ void conv2_rows(const float *ap, const float (&Wh)[36], f32x4 (&acc)[8])
{
#pragma unroll
    for (int p = 0; p < 8; p++) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int cq = 0; cq < 4; cq++)
#pragma unroll
            for (int p = 0; p < 8; p++) {
                const int oh = 2 * QA + (p >> 2), ow = p & 3;
                const int ih = 2 * oh - 1 + t / 3, iw = 2 * ow - 1 + t % 3;
                if (ih < 0 || ih > 6 || iw < 0 || iw > 6) continue;
                acc[p] = mfma(ap[(cq * 4 * 49 + ih * 7 + iw) * kF], Wh[t * 4 + cq], acc[p]);
            }
}

#ifdef STEM_PROBE        // probe build only (tools/stem_timeline_probe.py): wall-clock stamps of wave 0 at the phase boundaries
__device__ unsigned long long g_stem_probe[2048 * 8];
#define STEM_STAMP(i) do { if (tid == 0 && blockIdx.x < 2048) g_stem_probe[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
__device__ unsigned long long g_stem_probe2[2048 * 4];     // k_stem_fwd16: [0] HW_ID | XCC_ID << 32, [1] / [2] first pass: conv1 / conv2 done
#define STEM_STAMP2(i) do { if (tid == 0 && blockIdx.x < 2048) g_stem_probe2[blockIdx.x * 4 + (i)] = wall_clock64(); } while (0)
// k_stem_bwd16: per wave, inside the MFMA phase: [0] start, [1] (1) issued, [2] dz2 written, [3] last MFMA issued (s_memtime: shader clock)
#define BWD16_WAVE_STAMP(i) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 256) g_stem_probe2[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 4 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define STEM_STAMP(i) do { } while (0)
#define BWD16_WAVE_STAMP(i) do { } while (0)
#endif

template <typename XT>
__global__ __launch_bounds__(kThreads, 2) void k_stem_fwd16(StemPair pr)
{
    __shared__ __attribute__((aligned(16))) LdsF16 s;
    const int tid = (int)threadIdx.x, l = tid & 63, wave = tid >> 6;
    STEM_STAMP(0);
#ifdef STEM_PROBE
    if (tid == 0 && blockIdx.x < 2048) {
        g_stem_probe[blockIdx.x * 8 + 6] = 0ull;
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_stem_probe2[blockIdx.x * 4] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
    }
#endif
    // Which passes a workgroup takes. When the grid is two workgroups per CU, workgroups u and u + cus share CU u (dispatch order,
    // checked with HW_ID stamps), and where both want the same SIMD the OLDER one (u) issues first (measured: its pass takes 6.5 us
    // beside the younger one's 10; s_setprio did not change that). The passes rarely divide by the workgroups, so: (1) the problems
    // split the workgroups in CU order (v = 2 u + half: a CU's two workgroups are neighbours), (2) within a problem the older halves
    // come first in the block numbering — the low block numbers are the ones that get the extra pass. The headline's 4096 + 8192
    // frames = 768 passes on 512 workgroups: every CU gets three, two of them on its older workgroup, and the CU's tail is the
    // younger one finishing its only pass, not a lone workgroup with a whole pass to go (24.4 -> 20.5 us per launch).
    bool second;
    int blk, nblk;
    {
        const int i = (int)blockIdx.x, G = (int)gridDim.x, C = pr.cus;
        if (G == 2 * C) {
            const int half = i >= C ? 1 : 0, u = i - half * C, v = 2 * u + half, sp = pr.split;
            const int ne0 = (sp + 1) >> 1, no0 = sp >> 1;           // problem 0's older / younger workgroups
            second = v >= sp;
            if (!second) { blk = half ? ne0 + u : u; nblk = sp; }
            else { blk = half ? (C - ne0) + (u - no0) : u - ne0; nblk = G - sp; }
        } else {
            second = i >= pr.split;
            blk = second ? i - pr.split : i;
            nblk = second ? G - pr.split : pr.split;
        }
    }
    const StemProblem &pb = pr.p[second ? 1 : 0];
    const XT *__restrict__ x = reinterpret_cast<const XT *>(pb.x);
    const float *__restrict__ w1 = pb.w1, *__restrict__ b1 = pb.b1, *__restrict__ w2 = pb.w2, *__restrict__ b2 = pb.b2;
    float *__restrict__ y = pb.y;
    const long long M = pb.M, xs = pb.xs;
    const long long stride = (long long)nblk * kF;
    long long m0 = (long long)blk * kF;
    XTile xv = load_xtile(x, m0, M, xs, tid);
    const int c = l & 15, q = l >> 4;
    // wave = (output row pair qa, output channel half h): 50 real (position, tap) pairs x 4 channel groups = 200 MFMAs each
    const int qa = wave >> 1, h = wave & 1;
    // Prologue: every global load of the workgroup (first frames, conv2 weights, conv1 weights, biases) is in flight before the
    // first is waited for — one memory latency — and the barrier behind the weights' LDS copy orders LDS only.
    typedef const float __attribute__((address_space(1))) *gfloats;
    float wv[18];
#pragma unroll
    for (int k = 0; k < 18; k++) wv[k] = ((gfloats)w2)[tid + kThreads * k];
    const float bias = ((gfloats)b2)[16 * h + c];
    // conv1: this lane's two channels (ch, ch + 4: q <-> ch mod 4 keeps the a1 writes conflict-free) and its half of the positions
    const int ch = 8 * (wave & 1) + q, rh = wave >> 1;
    f32x2 cw[9];
#pragma unroll
    for (int k = 0; k < 9; k++) cw[k] = f32x2{((gfloats)w1)[ch * 9 + k], ((gfloats)w1)[(ch + 4) * 9 + k]};
    const f32x2 cb = f32x2{((gfloats)b1)[ch], ((gfloats)b1)[ch + 4]};
    float Wh[36];
    {
        float *wst = s.a1;                          // (see k_stem_fwd: the conv2 weights pass through LDS, rows padded to 145)
#pragma unroll
        for (int k = 0; k < 18; k++) {
            const int i = tid + kThreads * k;
            wst[(i / 144) * 145 + (i % 144)] = wv[k];
        }
        lds_barrier();
#pragma unroll
        for (int t = 0; t < 9; t++)
#pragma unroll
            for (int cq = 0; cq < 4; cq++) Wh[t * 4 + cq] = wst[(16 * h + c) * 145 + (4 * cq + q) * 9 + t];
    }
    store_xtile<XT>(s.x, xv, tid);
    xv = load_xtile(x, m0 + stride, M, xs, tid);
    const float *xf = s.x + c * kXF;
    float *a1o = s.a1 + ch * 49 * kF + c;
    const float *ap = s.a1 + q * 49 * kF + c;
    lds_barrier();
    STEM_STAMP(1);
    for (; m0 < M; m0 += stride) {
        if (rh == 0) conv1_lane2<0>(xf, a1o, a1o + 4 * 49 * kF, cw, cb);
        else conv1_lane2<1>(xf, a1o, a1o + 4 * 49 * kF, cw, cb);
        lds_barrier();                            // a1 complete; the x tile is free
        STEM_STAMP(2);
#ifdef STEM_PROBE
        if (m0 == (long long)blk * kF) STEM_STAMP2(1);
#endif
        store_xtile<XT>(s.x, xv, tid);
        xv = load_xtile(x, m0 + 2 * stride, M, xs, tid);
        f32x4 acc[8];
        if (qa == 0) conv2_rows<0>(ap, Wh, acc);
        else conv2_rows<1>(ap, Wh, acc);
        // Output: lane (c, q) holds frames 4 q + r, channel 16 h + c, 8 positions — 32 bytes here, 32 there: stored from
        // the accumulators that is 64 separate half-sector pieces per store instruction, and the L2's request rate (not its
        // bandwidth) becomes the kernel's bound (measured: 254 us at 163840 frames, 167 us with the same bytes laid out 1 KB
        // per instruction). So the 16 frames' outputs meet in LDS (over a1, which conv2 is done with) and leave as whole rows.
        lds_barrier();                              // a1 is free (and the next x tile complete)
        STEM_STAMP(3);
#ifdef STEM_PROBE
        if (m0 == (long long)blk * kF) STEM_STAMP2(2);
#endif
#pragma unroll
        for (int r = 0; r < 4; r++) {
            float4 *yt = reinterpret_cast<float4 *>(s.a1 + (4 * q + r) * kYF + (16 * h + c) * kYC + 8 * qa);
            yt[0] = make_float4(fmaxf(acc[0][r] + bias, 0.f), fmaxf(acc[1][r] + bias, 0.f), fmaxf(acc[2][r] + bias, 0.f),
                                fmaxf(acc[3][r] + bias, 0.f));
            yt[1] = make_float4(fmaxf(acc[4][r] + bias, 0.f), fmaxf(acc[5][r] + bias, 0.f), fmaxf(acc[6][r] + bias, 0.f),
                                fmaxf(acc[7][r] + bias, 0.f));
        }
        lds_barrier();
        // (plain stores: streamed past the L2 — nontemporal — the launch itself is 1-2 us shorter, but the GEMM that reads y
        // next pays more than that: measured +25..65 us per 20-step iteration)
        // Thread tid takes 16-byte piece tid & 3 of channel (tid >> 2) & 31 of frames (tid >> 7) + 2 i: all eight pieces are read
        // from LDS before the first leaves, and the stores go out relative to ONE uniform base (the pass's first row of y) with a
        // 32-bit lane offset — written frame by frame with 64-bit addresses and a bounds test each, the eight stores were eight
        // dependent LDS round trips and cost 32 address registers.
        {
            const unsigned fo = (unsigned)(tid >> 7), co = (unsigned)((tid >> 2) & 31), k = (unsigned)(tid & 3);
            const float *yt = s.a1 + fo * kYF + co * kYC + 4 * k;
            float4 v[8];
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = *reinterpret_cast<const float4 *>(yt + 2 * i * kYF);
            float *yb = y + m0 * 512;                                // uniform
            const unsigned o = fo * 512u + co * 16u + 4u * k;
            const long long left = M - m0;
            if (left >= kF) {
#pragma unroll
                for (int i = 0; i < 8; i++) *reinterpret_cast<float4 *>(yb + (o + 1024u * i)) = v[i];
            } else {
                const unsigned nv = (unsigned)left;
#pragma unroll
                for (int i = 0; i < 8; i++)
                    if (fo + 2u * i < nv) *reinterpret_cast<float4 *>(yb + (o + 1024u * i)) = v[i];
            }
        }
        lds_barrier();                              // the output tile has left a1
        STEM_STAMP(4);
#ifdef STEM_PROBE
        if (m0 == (long long)blk * kF) STEM_STAMP(7);      // (the end of the workgroup's FIRST pass; slot 6: its pass count)
        if (tid == 0 && blockIdx.x < 2048) g_stem_probe[blockIdx.x * 8 + 6] += 1ull;
#endif
    }
#ifdef STEM_PROBE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STEM_STAMP(5);
#endif
}

// dz2 = dy * (y > 0) of one frame: lane (c, q) holds channels c and 16+c, positions 4q..4q+3.
struct DzRegs { float4 y0, y1, d0, d1; };
typedef float stem_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 stem_ldg4(const float *p)
{
    const stem_v4f v = *reinterpret_cast<const stem_v4f __attribute__((address_space(1))) *>(
        (const float __attribute__((address_space(1))) *)p);
    return make_float4(v[0], v[1], v[2], v[3]);
}
// (branch-free like load_x: a frame index past the end re-reads the last frame, and the loop never uses it)
__device__ __forceinline__ DzRegs load_dz(const float *__restrict__ y, const float *__restrict__ dy, long long m,
                                          long long M, int c, int q)
{
    DzRegs r;
    const long long mm = m < M ? m : M - 1;
    const float *yy = y + mm * 512, *dd = dy + mm * 512;
    r.y0 = stem_ldg4(yy + 4 * (c * 4 + q)); r.y1 = stem_ldg4(yy + 4 * ((16 + c) * 4 + q));
    r.d0 = stem_ldg4(dd + 4 * (c * 4 + q)); r.d1 = stem_ldg4(dd + 4 * ((16 + c) * 4 + q));
    return r;
}
__device__ __forceinline__ float4 relu_mask(const float4 &yv, const float4 &dv)
{
    return make_float4(yv.x > 0.f ? dv.x : 0.f, yv.y > 0.f ? dv.y : 0.f, yv.z > 0.f ? dv.z : 0.f, yv.w > 0.f ? dv.w : 0.f);
}

// ------------------------------------------------------------------------------------------------------
// backward, one pass over (x, y, dy) per frame:
//  (1) dW2[co][ci][t] = sum_{frames, pos} dz2[co][pos] * a1pad[ci][2oh+kh][2ow+kw]: per output row s = oh the MFMA step
//      is A = dz2^T [co x 4 positions of the row], B = im2col rows [4 positions x 16 ci] for tap t -> 72 accumulator
//      VGPRs held across the whole frame loop.
//  (2) da1 = conv2^T(dz2): P_t[pos][ci] = sum_co dz2[pos][co] * W2[co][ci][t] (72 MFMAs, W2 in 72 VGPRs), scattered
//      to a1pad[ci][2oh+kh][2ow+kw] in registers: lane (ci = l&15, oh = l>>4) ends up with the a1-gradient of the
//      real rows 2oh (tap row kh=1) and 2oh+1 (kh=2 of its own output row + kh=0 of the next one, fetched from lane
//      l+16); dz1 = da1 * (a1 > 0) -> dW1 / db1 against the frame's input rows.
template <typename XT>
__global__ __launch_bounds__(kThreads, 2) void k_stem_bwd(const XT *__restrict__ x, const float *__restrict__ y,
                                                          const float *__restrict__ dy, const float *__restrict__ w1,
                                                          const float *__restrict__ b1, const float *__restrict__ w2,
                                                          float *__restrict__ partial, long long M, long long xs)
{
    __shared__ __attribute__((aligned(16))) LdsB lds[kWaves];
    __shared__ float w2s[kW2];
    const int l = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    LdsB &s = lds[wave];
    const long long stride = (long long)gridDim.x * kWaves;
    long long m = (long long)blockIdx.x * kWaves + wave;
    const int c = l & 15, q = l >> 4;
    XRegs xv = load_x(x, m, M, xs, l);
    DzRegs zv = load_dz(y, dy, m, M, c, q);
    zero_wave(s.x, kXSize, l);
    zero_wave(s.a1, kA1Size, l);
    f32x4 acc[2][9];        // dW2 tiles: rows co = 16n + 4q + r, column ci = c, tap t
#pragma unroll
    for (int t = 0; t < 9; t++) { acc[0][t] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[1][t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    float gb0 = 0.f, gb1 = 0.f;
    // B operand of (2): W2[co = 4 st + q][ci = c][t], shared by the workgroup's waves in LDS (bank-conflict free in
    // the native [co][ci][t] order: 9c + 16q mod 64 is injective over the wave)
    for (int i = (int)threadIdx.x; i < kW2; i += kThreads) w2s[i] = w2[i];
    const float *wq = w2s + q * 144 + c * 9;
    const Conv1W cw = load_conv1_w(w1, b1, l);
    f32x2 g01[3], gAB[3];   // dW1 accumulators: taps (kh, 0..1) | {tap (2,kw) via row A, tap (0,kw) via row B}
    float g2[3];            // taps (kh, 2)
#pragma unroll
    for (int k = 0; k < 3; k++) { g01[k] = f32x2{0.f, 0.f}; gAB[k] = f32x2{0.f, 0.f}; g2[k] = 0.f; }
    float gb = 0.f;
    const float *bp = s.a1 + c * kA1Ch + 2 * q;       // B operand of (1): input channel c, output column q of row s
    __syncthreads();
    for (; m < M; m += stride) {
        store_x<XT>(s.x, xv, l);
        const float4 z0 = relu_mask(zv.y0, zv.d0), z1 = relu_mask(zv.y1, zv.d1);
        gb0 += (z0.x + z0.y) + (z0.z + z0.w);
        gb1 += (z1.x + z1.y) + (z1.z + z1.w);
        reinterpret_cast<float4 *>(s.dz)[c * 4 + q] = z0;
        reinterpret_cast<float4 *>(s.dz)[(16 + c) * 4 + q] = z1;
        xv = load_x(x, m + stride, M, xs, l);
        zv = load_dz(y, dy, m + stride, M, c, q);
        wave_lds_sync();
        conv1_wave(s.x, s.a1, cw, l);
        wave_lds_sync();
        // (1) dW2
#pragma unroll
        for (int r = 0; r < 4; r++) {               // output row oh = r: positions 4r + q
            const float a0 = s.dz[c * 16 + 4 * r + q], a1v = s.dz[(16 + c) * 16 + 4 * r + q];
#pragma unroll
            for (int t = 0; t < 9; t++) {
                // padded rows 0 and 8 of a1 are the zero border: output row 0 never sees tap row 0, output row 3 never tap
                // row 2 — 12 of the 72 products are structurally zero and are not issued
                if ((r == 0 && t / 3 == 0) || (r == 3 && t / 3 == 2)) continue;
                const float b = bp[(2 * r + t / 3) * kA1Row + (t % 3)];
                acc[0][t] = mfma(a0, b, acc[0][t]);
                acc[1][t] = mfma(a1v, b, acc[1][t]);
            }
        }
        // (2) da1 -> dz1 -> dW1, db1
        f32x4 P[9];
#pragma unroll
        for (int t = 0; t < 9; t++) P[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < 8; st++) {
            const float a = s.dz[(4 * st + q) * 16 + c];       // A[pos = l&15][co = 4 st + q]
#pragma unroll
            for (int t = 0; t < 9; t++) P[t] = mfma(a, wq[st * 4 * 144 + t], P[t]);
        }
        // lane (ci = c, oh = q): P[kh*3+kw][ow] -> window columns 2ow+kw (padded 0..8), real columns 1..7
        float pw[3][7];
#pragma unroll
        for (int kh = 0; kh < 3; kh++) {
            const f32x4 p0 = P[kh * 3], p1 = P[kh * 3 + 1], p2 = P[kh * 3 + 2];
            pw[kh][0] = p1[0];
            pw[kh][1] = p2[0] + p0[1];
            pw[kh][2] = p1[1];
            pw[kh][3] = p2[1] + p0[2];
            pw[kh][4] = p1[2];
            pw[kh][5] = p2[2] + p0[3];
            pw[kh][6] = p1[3];
        }
        const float *a1c = s.a1 + c * kA1Ch + (2 * q + 1) * kA1Row + 1;
        float dzA[7], dzB[7];
#pragma unroll
        for (int j = 0; j < 7; j++) {
            const float up = __shfl_down(pw[0][j], 16, 64);      // kh = 0 row of output row oh+1
            const float aA = a1c[j], aB = a1c[kA1Row + j];
            dzA[j] = aA > 0.f ? pw[1][j] : 0.f;
            dzB[j] = (q < 3 && aB > 0.f) ? pw[2][j] + up : 0.f;
            gb += dzA[j] + dzB[j];
        }
        // a1 row r, tap kh reads padded input row 2r + kh: row A (r = 2q) uses rows 4q..4q+2, row B rows 4q+2..4q+4.
        // Packed FMAs: taps kw = 0,1 pair up against two neighbouring input values; on the shared input row (j = 2)
        // the rows A and B pair up against one broadcast input value.
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const float4 *row = reinterpret_cast<const float4 *>(s.x + (4 * q + j) * kXS);
            const float4 r0 = row[0], r1 = row[1], r2 = row[2], r3 = row[3];
            const float r[16] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w, r3.x, r3.y, r3.z, r3.w};
            if (j == 2) {
#pragma unroll
                for (int kw = 0; kw < 3; kw++) {
                    f32x2 g = gAB[kw];                      // {tap (2,kw) of row A, tap (0,kw) of row B}
#pragma unroll
                    for (int cc = 0; cc < 7; cc++)
                        g = __builtin_elementwise_fma(f32x2{dzA[cc], dzB[cc]}, f32x2{r[2 * cc + kw], r[2 * cc + kw]}, g);
                    gAB[kw] = g;
                }
            } else {
                const int kh = j < 2 ? j : j - 2;           // rows 0,1 feed row A; rows 3,4 feed row B (taps 1,2)
                const float(&dz)[7] = j < 2 ? dzA : dzB;
                f32x2 g = g01[kh];
                float g2v = g2[kh];
#pragma unroll
                for (int cc = 0; cc < 7; cc++) {
                    g = __builtin_elementwise_fma(f32x2{dz[cc], dz[cc]}, f32x2{r[2 * cc], r[2 * cc + 1]}, g);
                    g2v = fmaf(dz[cc], r[2 * cc + 2], g2v);
                }
                g01[kh] = g;
                g2[kh] = g2v;
            }
        }
    }
    // fold the packed accumulators back into the 3x3 filter gradient
    float g1[9];
#pragma unroll
    for (int kh = 0; kh < 3; kh++) { g1[kh * 3] = g01[kh].x; g1[kh * 3 + 1] = g01[kh].y; g1[kh * 3 + 2] = g2[kh]; }
#pragma unroll
    for (int kw = 0; kw < 3; kw++) { g1[6 + kw] += gAB[kw].x; g1[kw] += gAB[kw].y; }
#pragma unroll
    for (int k = 0; k < 9; k++) g1[k] = xor_sum(xor_sum(g1[k], 16), 32);
    gb = xor_sum(xor_sum(gb, 16), 32);
    gb0 = xor_sum(xor_sum(gb0, 16), 32);
    gb1 = xor_sum(xor_sum(gb1, 16), 32);
    // one record per workgroup: the four waves add up in wave order (fixed order -> reproducible sums)
    __syncthreads();
    float *red = reinterpret_cast<float *>(lds);
    float *rec = partial + (size_t)blockIdx.x * kPartial;
    for (int wv = 0; wv < kWaves; wv++) {
        if (wave == wv) {
#pragma unroll
            for (int n = 0; n < 2; n++)
#pragma unroll
                for (int t = 0; t < 9; t++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int slot = ((n * 9 + t) * 4 + r) * 64 + l;
                        const float v = acc[n][t][r] + (wv ? red[slot] : 0.f);
                        if (wv < kWaves - 1) red[slot] = v;
                        else rec[(16 * n + 4 * q + r) * 144 + c * 9 + t] = v;
                    }
            if (q == 0) {
                const float v0 = gb0 + (wv ? red[kW2 + c] : 0.f), v1 = gb1 + (wv ? red[kW2 + 16 + c] : 0.f);
                if (wv < kWaves - 1) { red[kW2 + c] = v0; red[kW2 + 16 + c] = v1; }
                else { rec[kW2 + c] = v0; rec[kW2 + 16 + c] = v1; }
#pragma unroll
                for (int k = 0; k < 10; k++) {
                    const int slot = kW2 + kC2 + k * 16 + c;
                    const float v = (k < 9 ? g1[k < 9 ? k : 0] : gb) + (wv ? red[slot] : 0.f);
                    if (wv < kWaves - 1) red[slot] = v;
                    else if (k < 9) rec[kW2 + kC2 + c * 9 + k] = v;
                    else rec[kW2 + kC2 + 144 + c] = v;
                }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------
// backward, 16 frames per workgroup pass (k_stem_bwd16): the learner's launch sizes (from 16384 frames up).
//
// The same move as k_stem_fwd16: with 16 FRAMES on an MFMA dimension, the products with the zero border of a1 are not issued —
// and with it the register col2im, the cross-lane shifts and the packed-FMA filter gradient of k_stem_bwd go too:
//   (1) dW2[co][ci][t] += sum over frames and the positions that are REAL for tap t of dz2[f][pos][co] * a1[f][ci][loc(pos, t)]:
//       A = dz2 as [co x 4 frames], B = a1 as [4 frames x ci]; 100 (position, tap) pairs x 4 frame groups x 2 halves of co
//       = 800 MFMAs per 16 frames (k_stem_bwd: 60 per frame = 960).
//   (2) da1[f][ci] at each of the 49 a1 locations = sum over the (position, tap) pairs that touch it and over co of
//       dz2[f][pos][co] * W2[co][ci][t] — chained into ONE accumulator per location (no scatter): 100 pairs x 8 groups of 4 co
//       = 800 MFMAs (k_stem_bwd: 72 per frame = 1152).
//   (3) dz1 = da1 * (a1 > 0) comes out of (2) as [frame = 4 q + r][ci = lane & 15] — which IS the A operand layout of an MFMA over
//       frames {r, 4 + r, 8 + r, 12 + r}: dW1[ci][tap] += dz1^T x window(x)[frame][tap] is 4 MFMAs per location (196 per pass) whose
//       B operand is one ds_read_b32 of the zero-bordered x tile per lane (lane & 15 = tap; column 9 is fed ones: db1).
// 512 threads, one workgroup per CU: the pass's tiles (x 15.6 KB, a1 53 KB, dz2 35 KB) are shared by 8 waves — two per SIMD —
// that each take one eighth of (1) (a 2x2 quadrant of positions x one half of co: 100 MFMAs) and one eighth of the 49 locations
// of (2) + (3) (12 pairs each, the centre's 4 on top of one of them). Both tiles are laid out [..][frame][17]: read with the frame
// on the lane's low bits or on its high bits, the 64 lanes of a ds_read_b32 fall into different banks either way.
constexpr int kXB = 244;                        // x tile per frame: 15 rows x 16 (zero border all round) + 4: bank 52 f mod 64
constexpr int kA1L = 16 * 17;                   // a1 tile: [49 locations][16 frames][17]
constexpr int kDzP = 2 * 16 * 17 + 4;           // dz2 tile: [16 positions][2 halves of co][16 frames][17] (+4: the writer's banks)
constexpr int kThreadsB16 = 512;
struct LdsB16 { float x[kF * kXB]; float a1[49 * kA1L]; float dz[2][16 * kDzP]; float w2[kW2]; };   // 15.3 + 52.1 + 2 x 34.3 + 18 KB = 153.9 KB

// which of the 8 waves owns a1 location (ih, iw) in steps (2) + (3): four rotationally symmetric 24-pair regions cut in two,
// the centre (4 pairs) with role 4
__host__ __device__ constexpr int bwd16_role_of(int ih, int iw)
{
    // (three locations change hands so that the four SIMDs — waves w and w + 4 share one — carry 252 / 244 / 252 / 248 of the 996
    // MFMAs of (2) + (3) instead of 260 / 256 / 240 / 240)
    if (ih == 5 && iw == 6) return 7;
    if (ih == 6 && iw == 0) return 4;
    if (ih == 0 && iw == 0) return 6;
    return (ih <= 2 && iw <= 3) ? (ih == 1 ? 0 : 1)
         : (ih <= 3 && iw >= 4) ? (ih <= 1 ? 2 : 3)
         : (ih >= 4 && iw >= 3) ? (ih == 5 ? 4 : 5)
         : (ih >= 3 && iw <= 2) ? (ih <= 4 ? 6 : 7)
         : 4;
}

template <int ROLE, typename Mid>
__device__ __forceinline__ void bwd16_mfma_phase(const LdsB16 &s, const float *dzt, int c, int q, f32x4 (&dw2)[9], f32x4 &dw1, Mid &&mid)
{
    // ---- (1): quadrant (QA, QB) of the output positions x half H of the output channels
    constexpr int QA = ROLE >> 2, QB = (ROLE >> 1) & 1, H = ROLE & 1;
    BWD16_WAVE_STAMP(0);
    {
        const float *dzl = dzt + H * 272 + q * 17 + c;           // A[i = co][k = frame 4 kk + q]
        const float *a1l = s.a1 + q * 17 + c;                    // B[k = frame 4 kk + q][j = ci]
        float A[4][4];
#pragma unroll
        for (int p = 0; p < 4; p++)
#pragma unroll
            for (int kk = 0; kk < 4; kk++) A[p][kk] = dzl[((2 * QA + (p >> 1)) * 4 + 2 * QB + (p & 1)) * kDzP + kk * 4 * 17];
#pragma unroll
        for (int t = 0; t < 9; t++)
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const int oh = 2 * QA + (p >> 1), ow = 2 * QB + (p & 1);
                const int ih = 2 * oh - 1 + t / 3, iw = 2 * ow - 1 + t % 3;
                if (ih < 0 || ih > 6 || iw < 0 || iw > 6) continue;
#pragma unroll
                for (int kk = 0; kk < 4; kk++) dw2[t] = mfma(A[p][kk], a1l[(ih * 7 + iw) * kA1L + kk * 4 * 17], dw2[t]);
            }
    }
    // (the next pass's dz2 tile is written HERE, between the wave's two MFMA streams: at the head of the phase all eight waves would
    // be writing at once with the matrix pipe idle)
    BWD16_WAVE_STAMP(1);
    mid();
    BWD16_WAVE_STAMP(2);
    // ---- (2) + (3): this role's a1 locations
    const float *dzf = dzt + c * 17 + q;                         // A[i = frame][k = co 4 kc + q]
    const float *w2l = s.w2 + q * 144 + c * 9;                   // B[k = co 4 kc + q][j = ci] of tap t: native [co][ci][t] order,
                                                                 // bank (16 q + 9 c) mod 64: conflict-free
    const float *a1m = s.a1 + (4 * q) * 17 + c;                  // the mask: a1[loc][frame 4 q + r][ci = c]
    const int tap = c < 9 ? c : 0;
    const float *xw = s.x + (4 * q) * kXB + (tap / 3) * 16 + (tap % 3);     // B of (3): lane & 15 = tap, frame 4 q + r
    // A location's tail (mask, then the 4 MFMAs of (3)) depends on its last MFMA of (2): it is issued one location late, behind
    // the next location's MFMAs, so that the matrix pipe has work while the result comes back.
    auto tail = [&](const f32x4 &d, int ih, int iw) {
        const int loc = ih * 7 + iw;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float dz1 = a1m[loc * kA1L + r * 17] > 0.f ? d[r] : 0.f;
            const float xv = xw[r * kXB + (2 * ih) * 16 + 2 * iw];
            dw1 = mfma(dz1, c < 9 ? xv : (c == 9 ? 1.0f : 0.0f), dw1);
        }
    };
    f32x4 dp = {0.f, 0.f, 0.f, 0.f};
    int pih = -1, piw = -1;
#pragma unroll
    for (int ih = 0; ih < 7; ih++)
#pragma unroll
        for (int iw = 0; iw < 7; iw++) {
            if (bwd16_role_of(ih, iw) != ROLE) continue;
            f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kh = 0; kh < 3; kh++)
#pragma unroll
                for (int kw = 0; kw < 3; kw++) {
                    const int oh2 = ih + 1 - kh, ow2 = iw + 1 - kw;          // 2 oh, 2 ow
                    if ((oh2 & 1) || (ow2 & 1) || oh2 < 0 || oh2 > 6 || ow2 < 0 || ow2 > 6) continue;
                    const int pos = (oh2 >> 1) * 4 + (ow2 >> 1);
#pragma unroll
                    for (int kc = 0; kc < 8; kc++)
                        d = mfma(dzf[pos * kDzP + (kc >> 2) * 272 + (kc & 3) * 4], w2l[kc * 4 * 144 + kh * 3 + kw], d);
                }
            if (pih >= 0) tail(dp, pih, piw);
            dp = d; pih = ih; piw = iw;
        }
    tail(dp, pih, piw);
    BWD16_WAVE_STAMP(3);
}

template <typename XT>
__global__ __launch_bounds__(kThreadsB16) void k_stem_bwd16(const XT *__restrict__ x, const float *__restrict__ y,
                                                            const float *__restrict__ dy, const float *__restrict__ w1,
                                                            const float *__restrict__ b1, const float *__restrict__ w2,
                                                            float *__restrict__ partial, long long M, long long xs)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_b16[];
    LdsB16 &s = *reinterpret_cast<LdsB16 *>(lds_b16);
    const int tid = (int)threadIdx.x, l = tid & 63, wave = tid >> 6, c = l & 15, q = l >> 4;
    const long long stride = (long long)gridDim.x * kF;
    long long m0 = (long long)blockIdx.x * kF;
    // the prefetch registers: 4 float4 of y and of dy of a pass (element idx = tid + 512 i: frame idx >> 7, 4 positions of one output
    // channel), 6 elements of x (frame tid >> 5)
    float4 yv[4], dv[4];
    uint32_t xr[6];
    auto load_yd = [&](long long mb) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int idx = tid + kThreadsB16 * i;
            long long m = mb + (idx >> 7);
            m = m < M ? m : M - 1;
            yv[i] = stem_ldg4(y + m * 512 + (idx & 127) * 4);
            dv[i] = stem_ldg4(dy + m * 512 + (idx & 127) * 4);
        }
    };
    auto load_x16 = [&](long long mb) {
        long long m = mb + (tid >> 5);
        m = m < M ? m : M - 1;
        const XT __attribute__((address_space(1))) *px = (const XT __attribute__((address_space(1))) *)(x + m * xs);
#pragma unroll
        for (int k = 0; k < 6; k++) {
            int e = (tid & 31) + 32 * k;
            e = e < 169 ? e : 168;
            xr[k] = sizeof(XT) == 4 ? __float_as_uint((float)px[e]) : (uint32_t)px[e];
        }
    };
    float gb2 = 0.f;                                // db2 of channel (tid >> 2) & 31, this thread's elements
    auto write_x = [&]() {                          // the interior of the bordered x tile
        float *xd = s.x + (tid >> 5) * kXB;
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const int e = (tid & 31) + 32 * k, row = (e * 79) >> 10;
            if (e < 169) xd[(row + 1) * 16 + (e - row * 13) + 1] = sizeof(XT) == 4 ? __uint_as_float(xr[k]) : (float)xr[k];
        }
    };
    auto write_dz = [&](float *dzt, long long mb) {   // dz2 = dy * (y > 0) of the pass at mb (zero past the last frame)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int idx = tid + kThreadsB16 * i, f = idx >> 7, co = (idx >> 2) & 31, k4 = idx & 3;
            float4 z = relu_mask(yv[i], dv[i]);
            if (mb + f >= M) z = make_float4(0.f, 0.f, 0.f, 0.f);
            gb2 += (z.x + z.y) + (z.z + z.w);
            float *d = dzt + (4 * k4) * kDzP + (co >> 4) * 272 + f * 17 + (co & 15);
            d[0] = z.x; d[kDzP] = z.y; d[2 * kDzP] = z.z; d[3 * kDzP] = z.w;
        }
    };
    load_yd(m0);
    load_x16(m0);
    for (int i = tid; i < kF * kXB; i += kThreadsB16) s.x[i] = 0.0f;      // the border stays zero: passes rewrite the interior only
    // B operand of (2): W2 in LDS for the whole kernel (in 72 VGPRs it pushed the MFMA phase into scratch spills)
    for (int i = tid; i < kW2; i += kThreadsB16) s.w2[i] = w2[i];
    // conv1 (recomputed) on packed FMAs: lane = (frame l & 15, channels 2 wave and 2 wave + 1, output rows 2 q and 2 q + 1)
    Conv1W cw;
#pragma unroll
    for (int k = 0; k < 9; k++) cw.w[k] = f32x2{w1[(2 * wave) * 9 + k], w1[(2 * wave + 1) * 9 + k]};
    cw.b = f32x2{b1[2 * wave], b1[2 * wave + 1]};
    f32x4 dw2[9], dw1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 9; t++) dw2[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();                                // the x tile is zeroed
    write_x();
    write_dz(s.dz[0], m0);
    load_yd(m0 + stride);
    load_x16(m0 + stride);
    lds_barrier();
    // Per pass: conv1 | barrier | dz2 of the NEXT pass into the other buffer, then the MFMA phase | barrier | x of the next pass
    // | barrier. (a1 and x are single: the MFMA phase reads both; dz2 is double, so that its 16 LDS writes per thread and the next
    // pass's loads go out under the MFMAs.)
    int buf = 0;
    for (; m0 < M; m0 += stride, buf ^= 1) {
        STEM_STAMP(0);
        // ---- conv1 -> a1 tile [loc][frame][17]
        {
            const float *xf = s.x + c * kXB + (4 * q) * 16;         // padded input rows 4 q .. 4 q + 4
            float *a1o = s.a1 + (2 * q * 7) * kA1L + c * 17 + 2 * wave;
            float R[5][16];
#pragma unroll
            for (int j = 0; j < 5; j++) {
                const float4 *row = reinterpret_cast<const float4 *>(xf + j * 16);
                const float4 a = row[0], b = row[1], cc = row[2], d = row[3];
                const float t[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, cc.x, cc.y, cc.z, cc.w, d.x, d.y, d.z, d.w};
#pragma unroll
                for (int jj = 0; jj < 16; jj++) R[j][jj] = t[jj];
            }
#pragma unroll
            for (int rr = 0; rr < 2; rr++) {
                if (rr == 1 && q == 3) break;                       // output row 7 does not exist
                f32x2 acc[7];
#pragma unroll
                for (int cc = 0; cc < 7; cc++) acc[cc] = cw.b;
#pragma unroll
                for (int kh = 0; kh < 3; kh++)
#pragma unroll
                    for (int kw = 0; kw < 3; kw++)
#pragma unroll
                        for (int cc = 0; cc < 7; cc++) {
                            const float xv = R[2 * rr + kh][2 * cc + kw];
                            acc[cc] = __builtin_elementwise_fma(f32x2{xv, xv}, cw.w[kh * 3 + kw], acc[cc]);
                        }
#pragma unroll
                for (int cc = 0; cc < 7; cc++) {
                    a1o[(rr * 7 + cc) * kA1L] = fmaxf(acc[cc].x, 0.0f);
                    a1o[(rr * 7 + cc) * kA1L + 1] = fmaxf(acc[cc].y, 0.0f);
                }
            }
        }
        lds_barrier();
        STEM_STAMP(1);
        // ---- the next pass's dz2 (its y / dy arrived during conv1), the loads of the pass after it, then the MFMA phase by role
        STEM_STAMP(2);
        const float *dzt = s.dz[buf];
        auto mid = [&]() {
            write_dz(s.dz[buf ^ 1], m0 + stride);
            load_yd(m0 + 2 * stride);
        };
        switch (wave) {
        case 0: bwd16_mfma_phase<0>(s, dzt, c, q, dw2, dw1, mid); break;
        case 1: bwd16_mfma_phase<1>(s, dzt, c, q, dw2, dw1, mid); break;
        case 2: bwd16_mfma_phase<2>(s, dzt, c, q, dw2, dw1, mid); break;
        case 3: bwd16_mfma_phase<3>(s, dzt, c, q, dw2, dw1, mid); break;
        case 4: bwd16_mfma_phase<4>(s, dzt, c, q, dw2, dw1, mid); break;
        case 5: bwd16_mfma_phase<5>(s, dzt, c, q, dw2, dw1, mid); break;
        case 6: bwd16_mfma_phase<6>(s, dzt, c, q, dw2, dw1, mid); break;
        default: bwd16_mfma_phase<7>(s, dzt, c, q, dw2, dw1, mid); break;
        }
#ifdef STEM_PROBE
        if (l == 0 && blockIdx.x < 2048 / 2) g_stem_probe[(1024 + blockIdx.x) * 8 + wave] = wall_clock64();
#endif
        lds_barrier();
        STEM_STAMP(3);
        write_x();
        load_x16(m0 + 2 * stride);
        lds_barrier();
    }
    // ---- one record per workgroup; the waves add up in wave order (fixed order -> reproducible sums)
    float *red = s.a1;                              // kPartial floats, zeroed
    for (int i = tid; i < kPartial; i += kThreadsB16) red[i] = 0.0f;
    float *red2 = s.dz[0];                          // db2: [32 channels][16 contributors]
    red2[((tid >> 2) & 31) * 16 + (tid & 3) * 4 + (tid >> 7)] = gb2;
    __syncthreads();
    for (int wv = 0; wv < 8; wv++) {
        if (wave == wv) {
            const int h = wave & 1;
#pragma unroll
            for (int t = 0; t < 9; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) red[(16 * h + 4 * q + r) * 144 + c * 9 + t] += dw2[t][r];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                if (c < 9) red[kW2 + kC2 + (4 * q + r) * 9 + c] += dw1[r];
                else if (c == 9) red[kW2 + kC2 + 144 + 4 * q + r] += dw1[r];
            }
        }
        __syncthreads();
    }
    if (tid < 32) {
        float v = 0.f;
        for (int k = 0; k < 16; k++) v += red2[tid * 16 + k];
        red[kW2 + tid] = v;
    }
    __syncthreads();
    float *rec = partial + (size_t)blockIdx.x * kPartial;
    for (int i = tid; i < kPartial; i += kThreadsB16) rec[i] = red[i];
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

static int stem_grid(long long M, int blocks_per_cu)
{
    static int cached_cus = 0;   // queried once: hipGetDeviceProperties is far too slow for a per-launch call
    if (cached_cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached_cus = n;
    }
    const long long cap = (long long)cached_cus * blocks_per_cu, need = (M + kWaves - 1) / kWaves;
    return (int)(need < cap ? (need < 1 ? 1 : need) : cap);
}
constexpr int kFwdBlocksPerCu = 3, kBwdBlocksPerCu = 2;

// (workgroups per CU of the backward. Measured under the pipelined schedule at 4096 envs, where the question is how the other
// replica's rollout fares beside this kernel: 1 per CU +1 % alone but a loss once the dW GEMM runs in its co-run form,
// 4 / 8 / 16 per CU — shorter-lived workgroups, more partial records — 16.55 / 16.50 / 16.42 M env steps/s against 16.73 with 2)
static int stem_bwd_blocks()
{
    static const int bpc = getenv("ATR_STEM_BWD_BLOCKS") ? atoi(getenv("ATR_STEM_BWD_BLOCKS")) : kBwdBlocksPerCu;   // (occupancy experiments)
    return bpc >= 1 && bpc <= 16 ? bpc : kBwdBlocksPerCu;
}

extern "C" long long atr_stem_workspace_floats(long long M) { return (long long)stem_grid(M, stem_bwd_blocks()) * kPartial; }

static StemProblem make_problem(const void *x, long long x_stride, const float *w1, const float *b1, const float *w2,
                                const float *b2, float *y, long long M)
{
    StemProblem p;
    p.x = x; p.w1 = w1; p.b1 = b1; p.w2 = w2; p.b2 = b2; p.y = y; p.M = M; p.xs = x_stride;
    return p;
}

// From 16384 frames per launch up the forward runs 16 frames per workgroup pass (k_stem_fwd16: no MFMA on border zeros; 0.65 of the
// f32 MFMA peak at 163840 frames against 0.54 of k_stem_fwd); below that by use_fwd16()'s rule. Same results bit for bit either way.
constexpr int kFwd16BlocksPerCu = 2;
static long long fwd16_min_frames()
{
    static const long long v = getenv("ATR_STEM_FWD16_MIN") ? atoll(getenv("ATR_STEM_FWD16_MIN")) : 16384;   // (crossover experiments)
    return v;
}
// Which forward kernel a launch of `passes` 16-frame passes (over all its problems) and M frames gets. From fwd16_min_frames()
// up always the 16-frame kernel; below it only where there are enough frames (3072) for a pass per CU and the passes load the CUs
// evenly — at most one pass per workgroup slot, or CU rounds that are >= 85 % full (k_stem_fwd16 hands a CU's passes to its two
// workgroups itself). 768 passes (the headline's tracker-aware pair at 4096 envs: 4096 + 8192 frames) are exactly three per CU:
// 19.8 us against the wave-per-frame kernel's 24.4 (round 5, before the CU-aware assignment: a tie at 24.1); 512 passes
// (4096 + 4096) 15.6 against 15.5 (19.3 in round 5's measurement), 384 passes 14.2 / 14.2, 192 passes 8.9 / 9.1. A third workgroup
// per CU would make 768 passes one round: built (byte x tile, 52 KB of LDS) and dropped — under the 168-VGPR cap of three waves per
// SIMD the kernel spills (241 us at 163840 frames against 182). ATR_STEM_FWD16_MIN set by hand switches the rule off (the A/B tools).
static bool use_fwd16(long long M, long long passes)
{
    if (M >= fwd16_min_frames()) return true;
    if (getenv("ATR_STEM_FWD16_MIN") || M < 3072) return false;
    const long long cus = (long long)stem_grid(1LL << 40, 1), slots = cus * kFwd16BlocksPerCu;
    const long long rounds = (passes + cus - 1) / cus;
    return passes <= slots || passes * 100 >= rounds * cus * 85;
}

static int stem_grid16(long long M)
{
    static const int bpc = getenv("ATR_STEM_FWD16_BLOCKS") ? atoi(getenv("ATR_STEM_FWD16_BLOCKS")) : kFwd16BlocksPerCu;   // (co-run experiments)
    const long long cap = (long long)stem_grid(1LL << 40, bpc >= 1 && bpc <= 2 ? bpc : kFwd16BlocksPerCu), need = (M + kF - 1) / kF;
    return (int)(need < cap ? (need < 1 ? 1 : need) : cap);
}

template <typename XT>
static int stem_forward_impl(const XT *x, long long x_stride, const float *w1, const float *b1, const float *w2,
                             const float *b2, float *y, long long M, void *stream)
{
    if (!x || !w1 || !b1 || !w2 || !b2 || !y || M < 0 || x_stride < 169) return -1;
    if (M == 0) return 0;
    StemPair pr;
    pr.cus = stem_grid(1LL << 40, 1);
    pr.p[0] = pr.p[1] = make_problem(x, x_stride, w1, b1, w2, b2, y, M);
    if (use_fwd16(M, (M + kF - 1) / kF)) {
        pr.split = stem_grid16(M);
        hipLaunchKernelGGL((k_stem_fwd16<XT>), dim3((unsigned)pr.split), dim3(kThreads), 0, (hipStream_t)stream, pr);
    } else {
        pr.split = stem_grid(M, kFwdBlocksPerCu);
        hipLaunchKernelGGL((k_stem_fwd<XT>), dim3((unsigned)pr.split), dim3(kThreads), 0, (hipStream_t)stream, pr);
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

template <typename XT>
static int stem_forward2_impl(const XT *x0, long long x0_stride, const float *w1_0, const float *b1_0, const float *w2_0,
                              const float *b2_0, float *y0, long long M0, const XT *x1, long long x1_stride,
                              const float *w1_1, const float *b1_1, const float *w2_1, const float *b2_1, float *y1,
                              long long M1, void *stream)
{
    if (!x0 || !w1_0 || !b1_0 || !w2_0 || !b2_0 || !y0 || M0 <= 0 || x0_stride < 169 || !x1 || !w1_1 || !b1_1 ||
        !w2_1 || !b2_1 || !y1 || M1 <= 0 || x1_stride < 169)
        return -1;
    StemPair pr;
    pr.cus = stem_grid(1LL << 40, 1);
    pr.p[0] = make_problem(x0, x0_stride, w1_0, b1_0, w2_0, b2_0, y0, M0);
    pr.p[1] = make_problem(x1, x1_stride, w1_1, b1_1, w2_1, b2_1, y1, M1);
    // split the resident workgroups in proportion to the frame counts (every wave gets the same number of frames)
    const bool f16 = use_fwd16(M0 + M1, (M0 + kF - 1) / kF + (M1 + kF - 1) / kF);
    const int per = f16 ? kF : kWaves;
    const int total = f16 ? stem_grid16(M0 + M1) : stem_grid(M0 + M1, kFwdBlocksPerCu);
    int g0 = (int)(((long long)total * M0 + (M0 + M1) / 2) / (M0 + M1));
    const int need0 = (int)((M0 + per - 1) / per), need1 = (int)((M1 + per - 1) / per);
    if (g0 < 1) g0 = 1;
    if (g0 > need0) g0 = need0;
    int g1 = total - g0;
    if (g1 < 1) g1 = 1;
    if (g1 > need1) g1 = need1;
    pr.split = g0;
    if (f16) hipLaunchKernelGGL((k_stem_fwd16<XT>), dim3((unsigned)(g0 + g1)), dim3(kThreads), 0, (hipStream_t)stream, pr);
    else hipLaunchKernelGGL((k_stem_fwd<XT>), dim3((unsigned)(g0 + g1)), dim3(kThreads), 0, (hipStream_t)stream, pr);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

static long long bwd16_min_frames()
{
    static const long long v = getenv("ATR_STEM_BWD16_MIN") ? atoll(getenv("ATR_STEM_BWD16_MIN")) : 16384;   // (crossover experiments)
    return v;
}

template <typename XT>
static int stem_backward_impl(const XT *x, long long x_stride, const float *y, const float *dy, const float *w1,
                              const float *b1, const float *w2, float *dw1, float *db1, float *dw2, float *db2,
                              float *workspace, long long M, void *stream)
{
    if (!x || !y || !dy || !w1 || !b1 || !w2 || !dw1 || !db1 || !dw2 || !db2 || !workspace || M <= 0 || x_stride < 169)
        return -1;
    hipStream_t st = (hipStream_t)stream;
    int grid;
    if (M >= bwd16_min_frames()) {      // 16 frames per workgroup pass, one workgroup per CU (never more records than stem_grid's)
        static bool lds_set[2] = {false, false};
        if (!lds_set[sizeof(XT) == 4]) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_stem_bwd16<XT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)sizeof(LdsB16)) != hipSuccess)
                return -2;
            lds_set[sizeof(XT) == 4] = true;
        }
        const long long cap = stem_grid(1LL << 40, 1), need = (M + kF - 1) / kF;
        grid = (int)(need < cap ? need : cap);
        hipLaunchKernelGGL((k_stem_bwd16<XT>), dim3((unsigned)grid), dim3(kThreadsB16), sizeof(LdsB16), st, x, y, dy, w1, b1, w2,
                           workspace, M, x_stride);
    } else {
        grid = stem_grid(M, stem_bwd_blocks());
        hipLaunchKernelGGL((k_stem_bwd<XT>), dim3((unsigned)grid), dim3(kThreads), 0, st, x, y, dy, w1, b1, w2, workspace, M,
                           x_stride);
    }
    hipLaunchKernelGGL(k_stem_reduce, dim3((kPartial + 15) / 16), dim3(1024), 0, st, workspace, grid, dw1, db1, dw2, db2);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int atr_stem_forward(const float *x, long long x_stride, const float *w1, const float *b1, const float *w2,
                                const float *b2, float *y, long long M, void *stream)
{
    return stem_forward_impl<float>(x, x_stride, w1, b1, w2, b2, y, M, stream);
}
extern "C" int atr_stem_forward_u8(const unsigned char *x, long long x_stride, const float *w1, const float *b1,
                                   const float *w2, const float *b2, float *y, long long M, void *stream)
{
    return stem_forward_impl<uint8_t>(x, x_stride, w1, b1, w2, b2, y, M, stream);
}
extern "C" int atr_stem_forward2(const float *x0, long long x0_stride, const float *w1_0, const float *b1_0,
                                 const float *w2_0, const float *b2_0, float *y0, long long M0, const float *x1,
                                 long long x1_stride, const float *w1_1, const float *b1_1, const float *w2_1,
                                 const float *b2_1, float *y1, long long M1, void *stream)
{
    return stem_forward2_impl<float>(x0, x0_stride, w1_0, b1_0, w2_0, b2_0, y0, M0, x1, x1_stride, w1_1, b1_1, w2_1, b2_1,
                                     y1, M1, stream);
}
extern "C" int atr_stem_forward2_u8(const unsigned char *x0, long long x0_stride, const float *w1_0, const float *b1_0,
                                    const float *w2_0, const float *b2_0, float *y0, long long M0,
                                    const unsigned char *x1, long long x1_stride, const float *w1_1, const float *b1_1,
                                    const float *w2_1, const float *b2_1, float *y1, long long M1, void *stream)
{
    return stem_forward2_impl<uint8_t>(x0, x0_stride, w1_0, b1_0, w2_0, b2_0, y0, M0, x1, x1_stride, w1_1, b1_1, w2_1,
                                       b2_1, y1, M1, stream);
}
extern "C" int atr_stem_backward(const float *x, long long x_stride, const float *y, const float *dy, const float *w1,
                                 const float *b1, const float *w2, float *dw1, float *db1, float *dw2, float *db2,
                                 float *workspace, long long M, void *stream)
{
    return stem_backward_impl<float>(x, x_stride, y, dy, w1, b1, w2, dw1, db1, dw2, db2, workspace, M, stream);
}
extern "C" int atr_stem_backward_u8(const unsigned char *x, long long x_stride, const float *y, const float *dy,
                                    const float *w1, const float *b1, const float *w2, float *dw1, float *db1, float *dw2,
                                    float *db2, float *workspace, long long M, void *stream)
{
    return stem_backward_impl<uint8_t>(x, x_stride, y, dy, w1, b1, w2, dw1, db1, dw2, db2, workspace, M, stream);
}

#ifdef STEM_PROBE
extern "C" int atr_stem_probe_read(unsigned long long *host, int n)
{
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(atr::g_stem_probe), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
extern "C" int atr_stem_probe2_read(unsigned long long *host, int n)
{
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(atr::g_stem_probe2), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
