// How fast does v_mfma_f32_16x16x4_f32 issue when consecutive MFMAs of a wave accumulate into the SAME registers (a dependent chain), into
// 2 / 4 / 8 alternating accumulators, and with one or two waves per SIMD?  Cycles per MFMA per SIMD (nominal: 32).
//   hipcc --offload-arch=gfx950 -O2 mfma_chain.hip -o mfma_chain && ./mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ unsigned long long g_clk[2];   // s_memtime counts, s_memrealtime (100 MHz) counts over the loop of block 0
template <int NACC>
__global__ void k(float *out, int iters, float a, float b)
{
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    float av = a + threadIdx.x, bv = b - threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 64 / NACC; r++)
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i], 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) { g_clk[0] = c1 - c0; g_clk[1] = w1 - w0; }
    float s = 0.f;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC> float run(int waves_per_simd, float *out)
{
    const int iters = 2000, threads = 256 * waves_per_simd;        // one workgroup per CU: 4 SIMDs x waves_per_simd waves
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(threads), 0, 0, out, 10, 1.f, 2.f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(threads), 0, 0, out, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)iters * 64 * waves_per_simd;
    unsigned long long clk[2];
    hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_clk), sizeof(clk));
    printf("   [%d acc, %d wave(s): s_memtime / s_memrealtime = %.1f counts per 10 ns]\n", NACC, waves_per_simd, (double)clk[0] / (double)clk[1]);
    return (float)(ms * 1e-3 * 2.4e9 / mfma_per_simd);             // cycles at 2.4 GHz per MFMA per SIMD
}
int main()
{
    float *out; hipMalloc(&out, 256 * 1024 * 4);
    for (int w = 1; w <= 2; w++)
        printf("%d wave(s) per SIMD: cycles per MFMA (at 2.4 GHz)  1 accumulator %.1f   2 accumulators %.1f   4 accumulators %.1f   8 accumulators %.1f\n", w,
               run<1>(w, out), run<2>(w, out), run<4>(w, out), run<8>(w, out));
    return 0;
}
