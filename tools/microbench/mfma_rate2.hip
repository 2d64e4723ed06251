#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
// variant 0: operands fixed; 1: operands from a float4 ring reloaded from global each chunk (like the actor-step kernel)
template <int VAR> __global__ __launch_bounds__(256, 1) void k32(const float *src, float *out, int chunks)
{
    f32x16 acc[2][2];
    for (int i = 0; i < 2; i++) for (int p = 0; p < 2; p++) for (int j = 0; j < 16; j++) acc[i][p][j] = 0;
    const int l = threadIdx.x & 63;
    const float *pa = src + (size_t)(blockIdx.x * 256 + threadIdx.x) * 4;
    float4 ra[4], rb0[4], rb1[4];
    for (int c = 0; c < 4; c++) { ra[c] = *(const float4 *)(pa + 1024 * c); rb0[c] = *(const float4 *)(pa + 1024 * c + 65536); rb1[c] = *(const float4 *)(pa + 1024 * c + 131072); }
    for (int c = 0; c < chunks; c += 4) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            float4 av = ra[u], b0 = rb0[u], b1 = rb1[u];
            if (VAR == 1) {
                const int cn = (c + u + 4) & 63;
                ra[u] = *(const float4 *)(pa + 1024 * cn); rb0[u] = *(const float4 *)(pa + 1024 * cn + 65536); rb1[u] = *(const float4 *)(pa + 1024 * cn + 131072);
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b0.x, acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b1.x, acc[1][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b0.y, acc[0][1], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b1.y, acc[1][1], 0, 0, 0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b0.z, acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b1.z, acc[1][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b0.w, acc[0][1], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b1.w, acc[1][1], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 2; i++) for (int p = 0; p < 2; p++) for (int j = 0; j < 16; j++) s += acc[i][p][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + l;
}
int main()
{
    float *src, *out;
    (void)hipMalloc(&src, 64 << 20); (void)hipMemset(src, 0, 64 << 20); (void)hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int chunks : {48, 4800}) for (int var = 0; var < 2; var++) {
        auto run = [&] { if (var) hipLaunchKernelGGL((k32<1>), dim3(256), dim3(256), 0, 0, src, out, chunks); else hipLaunchKernelGGL((k32<0>), dim3(256), dim3(256), 0, 0, src, out, chunks); };
        run(); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0); for (int i = 0; i < 10; i++) run(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        double fl = 4096.0 * 8 * chunks * 256 * 4 * 10;
        printf("chunks=%5d var=%d  %.2f us per launch  %.1f TF/s\n", chunks, var, ms * 100, fl / ms / 1e9);
    }
    return 0;
}
