// Do two waves of one SIMD overlap one wave's operand traffic (global load -> LDS write -> LDS read) with the other's MFMAs?
// Every wave: `blocks` K blocks of 16, each = 6 x (global_load_dwordx4 + ds_write_b128) + 2 x 3 ds_read_b128 + 16 v_mfma_f32_32x32x2
// (the actor-step kernel's mix at half its K-block size, 15.4 KB of LDS per wave). Grid 256 x 256 threads = one wave per SIMD,
// grid 512 = two.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kB = 16, kLs = kB + 4, kOp = 32 * kLs;
__global__ __launch_bounds__(256, 2) void k(const float *src, float *out, int blocks)
{
    __shared__ __attribute__((aligned(16))) float lds[4][2 * 3 * kOp];
    const int l = threadIdx.x & 63, wave = threadIdx.x >> 6, jj = l & 31, kk = l >> 5;
    const int rs = l >> 2, k4 = l & 3;                    // 16 rows x 4 float4 per load instruction
    const float *g = src + (size_t)((blockIdx.x * 4 + wave) & 1023) * 32 * 512;   // this wave's 32 rows of 512 floats (L2-resident)
    const float *pa0 = g + (size_t)rs * 512 + 4 * k4, *pa1 = pa0 + 16 * 512;
    const float *pb = src + (4u << 20) + (size_t)rs * 512 + 4 * k4;
    float *my = lds[wave];
    float *stp = my + rs * kLs + 4 * k4;
    const float *rd = my + jj * kLs + 4 * kk;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; i++) for (int p = 0; p < 2; p++) for (int j = 0; j < 16; j++) acc[i][p][j] = 0;
    float4 a0, a1, b0, b1, c0, c1;
#define LOADB(b) { const int o = kB * ((b) & 31); a0 = *(const float4 *)(pa0 + o); a1 = *(const float4 *)(pa1 + o); \
    b0 = *(const float4 *)(pb + o); b1 = *(const float4 *)(pb + 16 * 512 + o); c0 = *(const float4 *)(pb + 32 * 512 + o); c1 = *(const float4 *)(pb + 48 * 512 + o); }
#define STOREB(b) { float *d = stp + ((b) & 1) * 3 * kOp; *(float4 *)d = a0; *(float4 *)(d + 16 * kLs) = a1; d += kOp; \
    *(float4 *)d = b0; *(float4 *)(d + 16 * kLs) = b1; d += kOp; *(float4 *)d = c0; *(float4 *)(d + 16 * kLs) = c1; }
    LOADB(0); STOREB(0); LOADB(1);
    for (int b = 0; b < blocks; b++) {
        STOREB(b + 1); LOADB(b + 2);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_sched_barrier(0);
        const float *s = rd + (b & 1) * 3 * kOp;
#pragma unroll
        for (int cc = 0; cc < kB / 8; cc++) {
            const float4 av = *(const float4 *)(s + 8 * cc), v0 = *(const float4 *)(s + kOp + 8 * cc), v1 = *(const float4 *)(s + 2 * kOp + 8 * cc);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, v0.x, acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, v1.x, acc[1][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, v0.y, acc[0][1], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, v1.y, acc[1][1], 0, 0, 0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, v0.z, acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, v1.z, acc[1][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, v0.w, acc[0][1], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, v1.w, acc[1][1], 0, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    float sum = 0;
    for (int i = 0; i < 2; i++) for (int p = 0; p < 2; p++) for (int j = 0; j < 16; j++) sum += acc[i][p][j];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = sum;
}
int main()
{
    float *src, *out;
    (void)hipMalloc(&src, 96 << 20); (void)hipMemset(src, 0, 96 << 20); (void)hipMalloc(&out, 1024 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int blocks : {24, 240}) for (int grid : {256, 512}) {
        auto run = [&] { hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, src, out, blocks); };
        run(); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0); for (int i = 0; i < 20; i++) run(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double mf = (double)blocks * 16 * 64;     // MFMA cycles per wave
        printf("K blocks per wave %4d, grid %3d (%d wave(s) per SIMD): %8.2f us per launch; MFMA time per SIMD %.2f us at 2.1 GHz\n", blocks, grid,
               grid / 256, ms * 50, mf * (grid / 256) / 2100.0);
    }
    return 0;
}
