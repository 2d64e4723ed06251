// grid_barrier.hip — what does a layer boundary cost INSIDE a persistent kernel on this 8-XCD part, against a kernel boundary?
//   persistent: G workgroups (one or two per CU, all co-resident) loop K times over { write my slot; grid barrier; read a
//               slot written by a workgroup on ANOTHER XCD and check it } — the barrier is an agent-scope atomic counter +
//               release / acquire fences, i.e. what exchanging activations between layers needs;
//   launches:   the same exchange as K dependent launches of a G-workgroup kernel in a hipGraph.
// Build: hipcc --offload-arch=gfx950 -O3 -o scratch_exp/grid_barrier tools/microbench/grid_barrier.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void k_persistent(unsigned *counter, float *slots, int K, int *errors)
{
    const int G = (int)gridDim.x, b = (int)blockIdx.x;
    int bad = 0;
    for (int it = 1; it <= K; it++) {
        if (threadIdx.x == 0) slots[b] = (float)(it * 1000 + b);
        __syncthreads();
        if (threadIdx.x == 0) {
            __atomic_thread_fence(__ATOMIC_RELEASE);                                   // agent scope (default for HIP device code)
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)it * (unsigned)G;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) { }
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
        }
        __syncthreads();
        const int peer = (b + 1) % G;                                                  // workgroup b + 1 runs on the next XCD
        if (threadIdx.x == 0 && slots[peer] != (float)(it * 1000 + peer)) bad++;
        __syncthreads();
        if (threadIdx.x == 0) {                                                        // second barrier: slots may be rewritten
            __hip_atomic_fetch_add(counter + 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)it * (unsigned)G;
            while (__hip_atomic_load(counter + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) { }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && bad) atomicAdd(errors, bad);
}

__global__ void k_step(float *slots, int it, int *errors)
{
    const int G = (int)gridDim.x, b = (int)blockIdx.x;
    const int peer = (b + 1) % G;
    if (threadIdx.x == 0) {
        if (it > 1 && slots[G * ((it - 1) & 1) + peer] != (float)((it - 1) * 1000 + peer)) atomicAdd(errors, 1);
        slots[G * (it & 1) + b] = (float)(it * 1000 + b);
    }
}

int main()
{
    unsigned *counter;
    float *slots;
    int *errors;
    hipMalloc(&counter, 4096);
    hipMalloc(&slots, 8192 * sizeof(float));
    hipMalloc(&errors, 4);
    hipStream_t st;
    hipStreamCreate(&st);
    const int K = 200;
    for (int G : {256, 512}) {
        hipMemsetAsync(errors, 0, 4, st);
        double best = 1e30;
        for (int rep = 0; rep < 5; rep++) {
            hipMemsetAsync(counter, 0, 4096, st);
            hipStreamSynchronize(st);
            auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(k_persistent, dim3(G), dim3(256), 0, st, counter, slots, K, errors);
            hipStreamSynchronize(st);
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (us < best) best = us;
        }
        int e = 0;
        hipMemcpy(&e, errors, 4, hipMemcpyDeviceToHost);
        printf("persistent kernel, %3d workgroups: %7.2f us per exchange (two grid barriers + a cross-XCD read), stale reads %d\n", G,
               best / K, e);
        // the same exchange as K dependent launches in a graph
        hipMemsetAsync(errors, 0, 4, st);
        hipGraph_t g;
        hipGraphExec_t ge;
        hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        for (int it = 1; it <= K; it++) hipLaunchKernelGGL(k_step, dim3(G), dim3(256), 0, st, slots, it, errors);
        hipStreamEndCapture(st, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, st);
        hipStreamSynchronize(st);
        best = 1e30;
        for (int rep = 0; rep < 5; rep++) {
            auto t0 = std::chrono::steady_clock::now();
            hipGraphLaunch(ge, st);
            hipStreamSynchronize(st);
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (us < best) best = us;
        }
        hipMemcpy(&e, errors, 4, hipMemcpyDeviceToHost);
        printf("hipGraph of %d dependent launches, %3d workgroups: %7.2f us per launch, stale reads %d\n", K, G, best / K, e);
    }
    return 0;
}
