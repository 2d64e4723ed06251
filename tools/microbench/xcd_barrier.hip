// xcd_barrier.hip — what does a layer boundary cost inside a persistent kernel when the workgroups that exchange activations
// all sit on ONE XCD (one L2)? tools/microbench/grid_barrier.hip measured the chip-wide form (agent-scope release / acquire: an L2
// write-back + invalidate per boundary, 5.7 us) against a dependent launch in a hipGraph (1.9 us). Here every XCD's workgroups
// form a group of their own: data written with plain stores (write-through to the XCD's L2), released with a WORKGROUP-scope
// fence (s_waitcnt only: no L2 write-back), an atomic counter that lives in that L2 (RMW atomics execute in the L2; the spin reads it with sc1 loads, which miss the L1 —
// a fetch_add(0) is folded into a workgroup-scope load by the compiler, and that one may hit the L1), and an
// acquire that only drops the CU's L1 (buffer_inv sc1). The check reads what the other workgroups of the XCD wrote.
//   per iteration: every workgroup writes BYTES of its slot, barrier, reads the slots of 3 peers on its XCD and checks them,
//   barrier (slots may be rewritten).
// Build: hipcc --offload-arch=gfx950 -O3 -o scratch_exp/xcd_barrier tools/microbench/xcd_barrier.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

#define XCC_ID() (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15)      // hwreg(HW_REG_XCC_ID, 0, 4)

struct Ctl {
    unsigned claim[8 * 32];      // per XCD (128-byte apart): slots handed out
    unsigned bar[8 * 32];        // per XCD: barrier counter
};

// MODE 0: acquire = buffer_inv sc1 + plain loads; 1: no invalidate, loads with sc1 (device scope: miss the L1); 2: buffer_inv sc0
// + plain loads; 3: NO invalidate and plain loads, but every iteration uses a fresh set of slots (no address is read twice in
// one launch and none is read before it is written: can a never-read line be stale in the L1? — what a rollout step's
// activations look like: each layer's rows are written once, full 128-byte lines, and read after the barrier)
template <int MODE>
__global__ __launch_bounds__(256) void k_xcd(Ctl *ctl, float *slots_base, int floats, int K, int per_xcd, int *errors, unsigned *where)
{
    __shared__ unsigned s_slot;
    const int tid = (int)threadIdx.x;
    const unsigned xcc = XCC_ID();
    if (tid == 0) {
        s_slot = __hip_atomic_fetch_add(&ctl->claim[32 * xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        where[blockIdx.x] = xcc;
    }
    __syncthreads();
    const unsigned slot = s_slot;
    if (slot >= (unsigned)per_xcd) {                 // more workgroups on this XCD than planned: not part of the exchange
        if (tid == 0) atomicAdd(errors + 1, 1);
        return;
    }
    unsigned *bar = &ctl->bar[32 * xcc];
    unsigned epoch = 0;
    int bad = 0;
    for (int it = 1; it <= K; it++) {
        float *slots = slots_base + (MODE == 3 ? (size_t)(it - 1) * 8 * per_xcd * floats : (size_t)0);
        float *mine = slots + ((size_t)xcc * per_xcd + slot) * floats;
        for (int i = tid; i < floats; i += 256) mine[i] = (float)(it * 4096 + (int)slot * 64 + (i & 63));
        // stores acknowledged by the L2 before this wave arrives (a workgroup-scope release fence emits NO s_waitcnt on
        // gfx950 in this mode — checked in the ISA — so the wait is spelled out; gfx9 counts stores in vmcnt)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        epoch += (unsigned)per_xcd;
        if (tid == 0) {
            __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            int spins = 0;        // sc1 load: misses the L1, served by the L2; capped so that a wrong assumption cannot hang the box
            while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch && ++spins < (1 << 20)) __builtin_amdgcn_s_sleep(1);
            if (spins >= (1 << 20)) atomicAdd(errors + 1, 1000);
        }
        __syncthreads();
        if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");    // buffer_inv sc1: this CU's L1
        if (MODE == 2) asm volatile("buffer_inv sc0" ::: "memory");
        for (int pp = 1; pp <= 3; pp++) {
            const unsigned peer = (slot + (unsigned)pp * 7u) % (unsigned)per_xcd;
            const float *theirs = slots + ((size_t)xcc * per_xcd + peer) * floats;
            for (int i = tid; i < floats; i += 256) {
                float v;
                if (MODE != 1) v = theirs[i];
                else v = __hip_atomic_load(theirs + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v != (float)(it * 4096 + (int)peer * 64 + (i & 63))) bad++;
            }
        }
        __syncthreads();
        epoch += (unsigned)per_xcd;
        if (tid == 0) {
            __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            int spins = 0;        // sc1 load: misses the L1, served by the L2; capped so that a wrong assumption cannot hang the box
            while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch && ++spins < (1 << 20)) __builtin_amdgcn_s_sleep(1);
            if (spins >= (1 << 20)) atomicAdd(errors + 1, 1000);
        }
        __syncthreads();
    }
    if (bad) atomicAdd(errors, bad);
}

int main()
{
    Ctl *ctl;
    float *slots;
    int *errors;
    unsigned *where;
    const int max_floats = 4096;
    hipMalloc(&ctl, sizeof(Ctl));
    const int K = 400;
    hipMalloc(&slots, (size_t)K * 8 * 64 * 256 * sizeof(float) + (size_t)8 * 64 * max_floats * sizeof(float));   // (mode 3: K slot sets of 1 KB slots)
    hipMalloc(&errors, 8);
    hipMalloc(&where, 1024 * sizeof(unsigned));
    hipStream_t st;
    hipStreamCreate(&st);
    const char *names[4] = {"buffer_inv sc1 + plain loads", "sc1 loads", "buffer_inv sc0 + plain loads", "fresh slots every iteration, plain loads, no invalidate"};
    for (int mode = 0; mode < 4; mode++)
        for (int G : {256, 128, 512})
            for (int floats : {256, 4096}) {
                if (mode == 3 && floats != 256) continue;
                const int per_xcd = G / 8;
                double best = 1e30;
                int e[2] = {0, 0};
                for (int rep = 0; rep < 4; rep++) {
                    hipMemsetAsync(ctl, 0, sizeof(Ctl), st);
                    hipMemsetAsync(errors, 0, 8, st);
                    hipStreamSynchronize(st);
                    auto t0 = std::chrono::steady_clock::now();
                    if (mode == 0) hipLaunchKernelGGL(k_xcd<0>, dim3(G), dim3(256), 0, st, ctl, slots, floats, K, per_xcd, errors, where);
                    else if (mode == 1) hipLaunchKernelGGL(k_xcd<1>, dim3(G), dim3(256), 0, st, ctl, slots, floats, K, per_xcd, errors, where);
                    else if (mode == 2) hipLaunchKernelGGL(k_xcd<2>, dim3(G), dim3(256), 0, st, ctl, slots, floats, K, per_xcd, errors, where);
                    else hipLaunchKernelGGL(k_xcd<3>, dim3(G), dim3(256), 0, st, ctl, slots, floats, K, per_xcd, errors, where);
                    hipStreamSynchronize(st);
                    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                    if (us < best) best = us;
                    int ee[2];
                    hipMemcpy(ee, errors, 8, hipMemcpyDeviceToHost);
                    e[0] += ee[0]; e[1] += ee[1];
                }
                std::vector<unsigned> w(G);
                hipMemcpy(w.data(), where, G * sizeof(unsigned), hipMemcpyDeviceToHost);
                int rr = 0, hist[16] = {0};
                for (int b = 0; b < G; b++) { rr += (w[b] == (unsigned)(b % 8)); hist[w[b] & 15]++; }
                printf("mode %d (%s)  %3d workgroups, %5d B per slot: %6.2f us per iteration (2 XCD barriers + write + 3 peer reads), stale reads %d, "
                       "overflow workgroups %d, blockIdx %% 8 == XCC_ID for %d of %d, per-XCD counts",
                       mode, names[mode], G, floats * 4, best / K, e[0], e[1], rr, G);
                for (int x = 0; x < 8; x++) printf(" %d", hist[x]);
                printf("\n");
            }
    return 0;
}
