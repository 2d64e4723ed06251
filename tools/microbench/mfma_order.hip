// Is v_mfma_f32_16x16x4_f32's D = C + sum_k A[i][k] B[k][j] bit-identical to the fmaf chain fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0,c))))?
// (and which other orders match, if not). hipcc --offload-arch=gfx950 -O2 -ffp-contract=off mfma_order.hip -o mfma_order && ./mfma_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float *A, const float *B, const float *C, float *D)   // A [16][4], B [4][16], C/D [16][16]
{
    const int l = threadIdx.x, i = l & 15, kk = l >> 4;
    f32x4 c;
    for (int r = 0; r < 4; r++) c[r] = C[(4 * kk + r) * 16 + i];
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i * 4 + kk], B[kk * 16 + i], c, 0, 0, 0);
    for (int r = 0; r < 4; r++) D[(4 * kk + r) * 16 + i] = c[r];
}
int main()
{
    float hA[64], hB[64], hC[256], hD[256];
    float *dA, *dB, *dC, *dD;
    hipMalloc(&dA, 256); hipMalloc(&dB, 256); hipMalloc(&dC, 1024); hipMalloc(&dD, 1024);
    int n_chain = 0, n_rev = 0, n_tree = 0, n_exact = 0, total = 0;
    srand(1);
    for (int trial = 0; trial < 2000; trial++) {
        for (int i = 0; i < 64; i++) { hA[i] = (rand() / (float)RAND_MAX - 0.5f) * 4.f; hB[i] = (rand() / (float)RAND_MAX - 0.5f) * 4.f; }
        for (int i = 0; i < 256; i++) hC[i] = (rand() / (float)RAND_MAX - 0.5f) * 8.f;
        hipMemcpy(dA, hA, 256, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 256, hipMemcpyHostToDevice); hipMemcpy(dC, hC, 1024, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; i++)
            for (int j = 0; j < 16; j++) {
                const float a0 = hA[i * 4], a1 = hA[i * 4 + 1], a2 = hA[i * 4 + 2], a3 = hA[i * 4 + 3];
                const float b0 = hB[j], b1 = hB[16 + j], b2 = hB[32 + j], b3 = hB[48 + j], c = hC[i * 16 + j];
                const float chain = fmaf(a3, b3, fmaf(a2, b2, fmaf(a1, b1, fmaf(a0, b0, c))));
                const float rev = fmaf(a0, b0, fmaf(a1, b1, fmaf(a2, b2, fmaf(a3, b3, c))));
                const float tree = (float)((double)c + ((double)a0 * b0 + (double)a1 * b1 + (double)a2 * b2 + (double)a3 * b3));
                const double ex = (double)c + (double)a0 * b0 + (double)a1 * b1 + (double)a2 * b2 + (double)a3 * b3;
                const float d = hD[i * 16 + j];
                total++; n_chain += d == chain; n_rev += d == rev; n_tree += d == tree; n_exact += d == (float)ex;
            }
    }
    printf("elements %d: equal to the ascending fmaf chain %d, descending chain %d, one rounding of the exact sum %d / %d\n", total, n_chain, n_rev, n_tree, n_exact);
    return 0;
}
