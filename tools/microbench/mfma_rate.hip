#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC> __global__ __launch_bounds__(256) void k16(float *out, int iters)
{
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = f32x4{0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC> __global__ __launch_bounds__(256) void k32(float *out, int iters)
{
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; i++) for (int j = 0; j < 16; j++) acc[i][j] = 0;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; i++) for (int j = 0; j < 16; j++) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <class F> float time_ms(F f)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main()
{
    float *out; hipMalloc(&out, 4096 * 256 * 4);
    const int iters = 20000;
    for (int wg : {256, 512, 1024}) {
        float t1 = time_ms([&] { hipLaunchKernelGGL((k16<4>), dim3(wg), dim3(256), 0, 0, out, iters); });
        float t2 = time_ms([&] { hipLaunchKernelGGL((k16<8>), dim3(wg), dim3(256), 0, 0, out, iters / 2); });
        float t3 = time_ms([&] { hipLaunchKernelGGL((k32<2>), dim3(wg), dim3(256), 0, 0, out, iters); });
        float t4 = time_ms([&] { hipLaunchKernelGGL((k32<4>), dim3(wg), dim3(256), 0, 0, out, iters / 2); });
        double f16 = 2048.0 * 4 * iters * wg * 4, f32 = 4096.0 * 2 * iters * wg * 4;
        printf("wg=%4d  16x16x4 (4 acc) %.1f TF/s  (8 acc) %.1f TF/s   32x32x2 (2 acc) %.1f TF/s  (4 acc) %.1f TF/s\n", wg,
               f16 / t1 / 1e9, f16 / t2 / 1e9, f32 / t3 / 1e9, f32 / t4 / 1e9);
    }
    return 0;
}
