// two_queue.hip — how much do two hipGraph replays on two streams overlap on this stack, as a function of what the kernels are?
//   chain(n, us):  a graph of n dependent launches of a 1-workgroup kernel that spins for `us` microseconds
//   big(us, wgs):  one launch of `wgs` workgroups spinning for `us`
// Cases: chain alone; two chains on two streams; chain next to a big kernel that leaves CUs free (wgs = 64) or fills the
// chip (wgs = 1024). Build: hipcc --offload-arch=gfx950 -O3 -o scratch_exp/two_queue tools/microbench/two_queue.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void k_spin(long long ticks, int *sink)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) { }
    if (sink && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) *sink = 1;
}

// a streaming kernel: every workgroup keeps writing (mode 1) or reading (mode 2) its 1 MB window of `buf` until `ticks` pass
__global__ void k_stream(long long ticks, float *buf, int mode)
{
    const long long t0 = wall_clock64();
    float4 *p = reinterpret_cast<float4 *>(buf) + (size_t)blockIdx.x * 65536;
    float acc = 0.f;
    int it = 0;
    while (wall_clock64() - t0 < ticks) {
        for (int i = threadIdx.x; i < 65536; i += blockDim.x) {
            if (mode == 1) p[i] = make_float4(it, 1.f, 2.f, 3.f);
            else if (mode == 3) {
                float *q = reinterpret_cast<float *>(p + i);
                __builtin_nontemporal_store((float)it, q); __builtin_nontemporal_store(1.f, q + 1);
                __builtin_nontemporal_store(2.f, q + 2); __builtin_nontemporal_store(3.f, q + 3);
            } else if (mode == 4) {
                float *q = reinterpret_cast<float *>(p + i);
                __hip_atomic_store(q, (float)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(q + 1, 1.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(q + 2, 2.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(q + 3, 3.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            } else acc += p[i].x;
        }
        it++;
    }
    if (acc == 12345.f) buf[0] = acc;
}

static hipGraphExec_t stream_graph(hipStream_t st, long long ticks, int wgs, float *buf, int mode)
{
    hipGraph_t g;
    hipGraphExec_t e;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    hipLaunchKernelGGL(k_stream, dim3(wgs), dim3(256), 0, st, ticks, buf, mode);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
    return e;
}

// workgroups that hold LDS (so that only `per_cu` of them fit a CU) and spin `ticks` each, `rounds` times (a persistent grid
// loops inside; a multi-wave grid comes back as rounds x more workgroups)
template <int LDS_BYTES>
__global__ __launch_bounds__(256) void k_hold(long long ticks, int rounds, int *sink)
{
    __shared__ char lds[LDS_BYTES];
    lds[threadIdx.x] = (char)threadIdx.x;
    __syncthreads();
    for (int r = 0; r < rounds; r++) {
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < ticks) { }
    }
    if (sink && lds[threadIdx.x] == 77 && blockIdx.x == 0x7fffffff) *sink = 1;
}

template <int LDS_BYTES>
static hipGraphExec_t lds_chain_graph(hipStream_t st, int n, long long ticks, int wgs)
{
    hipGraph_t g;
    hipGraphExec_t e;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < n; i++) hipLaunchKernelGGL(k_hold<LDS_BYTES>, dim3(wgs), dim3(256), 0, st, ticks, 1, nullptr);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
    return e;
}

static hipGraphExec_t hold_graph(hipStream_t st, int wgs, long long ticks, int rounds)
{
    hipGraph_t g;
    hipGraphExec_t e;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    hipLaunchKernelGGL(k_hold<65536>, dim3(wgs), dim3(256), 0, st, ticks, rounds, nullptr);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
    return e;
}

static hipGraphExec_t chain_graph(hipStream_t st, int n, long long ticks, int wgs)
{
    hipGraph_t g;
    hipGraphExec_t e;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < n; i++) hipLaunchKernelGGL(k_spin, dim3(wgs), dim3(64), 0, st, ticks, nullptr);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
    return e;
}

int main()
{
    const long long tick_per_us = 100;   // wall_clock64: 100 MHz
    hipStream_t s[6];
    for (auto &x : s) hipStreamCreate(&x);
    auto run = [&](const char *name, std::vector<std::pair<hipGraphExec_t, hipStream_t>> gs) {
        for (int rep = 0; rep < 3; rep++) for (auto &p : gs) hipGraphLaunch(p.first, p.second);
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        const int R = 20;
        for (int rep = 0; rep < R; rep++) for (auto &p : gs) hipGraphLaunch(p.first, p.second);
        hipDeviceSynchronize();
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / R;
        printf("%-70s %9.1f us\n", name, us);
    };
    {
        hipStream_t a = s[0], b = s[1];
        hipGraphExec_t d0 = chain_graph(a, 300, 5 * tick_per_us, 1);
        run("chain of 300 x 5 us (1 workgroup), alone", {{d0, a}});
        // 64 KB of LDS per workgroup: two fit a CU, 512 are resident at a time
        hipGraphExec_t multi = hold_graph(b, 512 * 10, 100 * tick_per_us, 1);     // ten waves of 512 workgroups x 100 us
        hipGraphExec_t pers = hold_graph(b, 512, 100 * tick_per_us, 10);          // one wave that loops ten times
        hipGraphExec_t pers1 = hold_graph(b, 256, 100 * tick_per_us, 10);         // one workgroup per CU, looping
        hipGraphExec_t shortw = hold_graph(b, 512 * 100, 10 * tick_per_us, 1);    // a hundred waves x 10 us
        run("compute co-runner: 5120 workgroups x 100 us (2 per CU resident), alone", {{multi, b}});
        run("   chain of 300 x 5 us next to it", {{multi, b}, {d0, a}});
        run("compute co-runner: 512 persistent workgroups x 10 rounds x 100 us, alone", {{pers, b}});
        run("   chain of 300 x 5 us next to it", {{pers, b}, {d0, a}});
        run("compute co-runner: 256 persistent workgroups x 10 rounds x 100 us, alone", {{pers1, b}});
        run("   chain of 300 x 5 us next to it", {{pers1, b}, {d0, a}});
        run("compute co-runner: 51200 workgroups x 10 us, alone", {{shortw, b}});
        run("   chain of 300 x 5 us next to it", {{shortw, b}, {d0, a}});
        // chains whose workgroups need LDS: 72 KB does not fit next to two 64 KB workgroups (160 KB per CU), 24 KB does
        hipGraphExec_t c72 = lds_chain_graph<73728>(a, 300, 5 * tick_per_us, 256), c24 = lds_chain_graph<24576>(a, 300, 5 * tick_per_us, 256);
        hipGraphExec_t c8 = lds_chain_graph<8192>(a, 300, 5 * tick_per_us, 256);
        run("chain of 300 x 5 us x 256 workgroups with 72 KB LDS, alone", {{c72, a}});
        run("   next to 5120 workgroups x 100 us (2 x 64 KB per CU)", {{multi, b}, {c72, a}});
        run("   next to 51200 workgroups x 10 us (2 x 64 KB per CU)", {{shortw, b}, {c72, a}});
        run("   next to 256 persistent workgroups (1 x 64 KB per CU)", {{pers1, b}, {c72, a}});
        run("chain of 300 x 5 us x 256 workgroups with 24 KB LDS, alone", {{c24, a}});
        run("   next to 5120 workgroups x 100 us (2 x 64 KB per CU)", {{multi, b}, {c24, a}});
        run("   next to 51200 workgroups x 10 us (2 x 64 KB per CU)", {{shortw, b}, {c24, a}});
        run("chain of 300 x 5 us x 256 workgroups with 8 KB LDS, alone", {{c8, a}});
        run("   next to 5120 workgroups x 100 us (2 x 64 KB per CU)", {{multi, b}, {c8, a}});
    }
    float *buf;
    hipMalloc(&buf, (size_t)2048 << 20);
    if (getenv("TWOQ_MEM")) {
        hipStream_t a = s[0], b = s[1];
        hipGraphExec_t d0 = chain_graph(a, 300, 5 * tick_per_us, 1);
        run("chain of 300 x 5 us (1 workgroup), alone", {{d0, a}});
        float *ubuf = nullptr;
        if (hipExtMallocWithFlags((void **)&ubuf, (size_t)1024 << 20, hipDeviceMallocUncached) != hipSuccess) { printf("no uncached memory\n"); ubuf = buf; }
        for (int mode = 1; mode <= 5; mode++)
            for (int wgs : {64, 256, 1024}) {
                hipGraphExec_t w = stream_graph(b, 1000 * tick_per_us, wgs, mode == 5 ? ubuf : buf, mode == 5 ? 1 : mode);
                char nm[128];
                const char *what[] = {"", "WRITING", "READING", "WRITING (nontemporal)", "WRITING (system-scope stores)", "WRITING uncached memory"};
                snprintf(nm, sizeof nm, "1000 us of %d workgroups %s, alone", wgs, what[mode]);
                run(nm, {{w, b}});
                run("   chain of 300 x 5 us next to it (the excess over the chain alone = its slowdown during 1000 us)", {{w, b}, {d0, a}});
            }
    }
    for (int pair = 1; pair <= (getenv("TWOQ_MEM") ? 1 : 0); pair++) {
        hipStream_t a = s[0], b = s[pair];
        printf("--- streams 0 and %d\n", pair);
        hipGraphExec_t c0 = chain_graph(a, 100, 0, 1), c1 = chain_graph(b, 100, 0, 1);
        run("chain of 100 empty launches, alone", {{c0, a}});
        run("two such chains, two streams", {{c0, a}, {c1, b}});
        hipGraphExec_t d0 = chain_graph(a, 100, 5 * tick_per_us, 1), d1 = chain_graph(b, 100, 5 * tick_per_us, 1);
        run("chain of 100 x 5 us (1 workgroup), alone", {{d0, a}});
        run("two such chains, two streams", {{d0, a}, {d1, b}});
        hipGraphExec_t big64 = chain_graph(b, 1, 1000 * tick_per_us, 64), big1k = chain_graph(b, 1, 1000 * tick_per_us, 2048);
        run("one 1000 us kernel of 64 workgroups, alone", {{big64, b}});
        run("chain of 100 x 5 us next to it", {{big64, b}, {d0, a}});
        run("one 1000 us kernel of 2048 workgroups, alone", {{big1k, b}});
        run("chain of 100 x 5 us next to it", {{big1k, b}, {d0, a}});
        hipGraphExec_t m0 = chain_graph(b, 20, 50 * tick_per_us, 256);
        run("chain of 20 x 50 us x 256 workgroups, alone", {{m0, b}});
        run("chain of 100 x 5 us next to it", {{m0, b}, {d0, a}});
    }
    return 0;
}
