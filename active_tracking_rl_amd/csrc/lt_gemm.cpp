// lt_gemm.cpp — the policy's plain Linear layers through hipBLASLt, called directly (include/atr_policy.h: atr_linear).
//
//   C[b] = act(A[b] W[b]^T + bias)      A [M, K] rows lda, W [N, K] rows ldw (nn.Linear layout), C [M, N] rows ldc
//
// Why not torch.addmm / torch.bmm: the rollout's LSTMCell (model.py:110,137,172,203 of the reference) wants its two GEMMs
// as ONE product over the concatenated row [features | k h_prev] (K = 256 + 128) so that the step's last kernel reads one gate
// tensor instead of two. That needs the encoder's fc + ReLU (perception.py:81,90) to write its 256 columns INTO rows of
// 384 floats (ldc = 384) — and PyTorch's addmm with a fused ReLU epilogue falls back to copy + GEMM + ReLU launches as soon
// as the output is not contiguous. The library underneath has no such limit; this file is that one call, plus what TunableOp
// does for torch's GEMMs: every distinct problem is timed once over the library's candidate kernels (outside any stream
// capture) and the fastest is kept for the life of the process.
//
// The library is NOT linked: PyTorch-ROCm ships and loads its own libhipblaslt.so (with its kernel database next to it); a
// second copy from /opt/rocm would drag a second HIP runtime into the process. atr_lt_init() is given the path of the copy
// PyTorch uses (fused.py passes torch/lib/libhipblaslt.so) and resolves the dozen entry points from it.
#include "../../include/atr_policy.h"

#if __has_include(<hipblaslt/hipblaslt.h>)
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <hipblaslt/hipblaslt-version.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <string>
#include <mutex>
#include <tuple>
#include <vector>


namespace {

struct LtApi {
    void *so = nullptr;
    decltype(&hipblasLtCreate) Create = nullptr;
    decltype(&hipblasLtMatrixLayoutCreate) LayoutCreate = nullptr;
    decltype(&hipblasLtMatrixLayoutDestroy) LayoutDestroy = nullptr;
    decltype(&hipblasLtMatrixLayoutSetAttribute) LayoutSet = nullptr;
    decltype(&hipblasLtMatmulDescCreate) DescCreate = nullptr;
    decltype(&hipblasLtMatmulDescDestroy) DescDestroy = nullptr;
    decltype(&hipblasLtMatmulDescSetAttribute) DescSet = nullptr;
    decltype(&hipblasLtMatmulPreferenceCreate) PrefCreate = nullptr;
    decltype(&hipblasLtMatmulPreferenceDestroy) PrefDestroy = nullptr;
    decltype(&hipblasLtMatmulPreferenceSetAttribute) PrefSet = nullptr;
    decltype(&hipblasLtMatmulAlgoGetHeuristic) Heuristic = nullptr;
    decltype(&hipblasLtMatmul) Matmul = nullptr;
    decltype(&hipblasLtGetVersion) GetVersion = nullptr;
    decltype(&hipblasLtGetGitRevision) GetGitRevision = nullptr;
    // hipblaslt_ext::getIndexFromAlgo / getKernelNameFromAlgo (C++ entry points of the same library, optional): the identity of
    // a kernel choice that survives a re-ordering of the heuristic's list
    int (*IndexFromAlgo)(hipblasLtMatmulAlgo_t &) = nullptr;
    std::string (*KernelNameFromAlgo)(void *, hipblasLtMatmulAlgo_t &) = nullptr;
};

constexpr size_t kWorkspaceBytes = 64u << 20;
constexpr int kMaxAlgos = 512;

struct Plan {
    hipblasLtMatmulDesc_t desc = nullptr;
    hipblasLtMatrixLayout_t la = nullptr, lb = nullptr, ld = nullptr;
    hipblasLtMatmulAlgo_t algo;
    size_t ws_bytes = 0;
    bool tuned = false;
    int candidates = 0, chosen = -1, preset = -1;
    int preset_solution = -1;     // the recorded choice's solution index (-1: not recorded, position alone decides)
    int solution = -1;            // solution index of the kernel in use (-1: the library does not say)
    int source = ATR_LT_SOURCE_NONE;
    float best_us = 0.f;
};

// everything that selects a kernel: shape, strides, batch, epilogue (pointers are per call)
typedef std::tuple<int, int, int, int, long long, long long, long long, long long, long long, long long, int, int> Key;

struct LtState {
    LtApi api;
    hipblasLtHandle_t handle = nullptr;
    void *workspace = nullptr;
    std::map<Key, Plan> plans;
    std::mutex mu;
    char err[256] = {0};
    int version = 0;              // hipblasLtGetVersion of the LOADED library (major * 100000 + minor * 100 + patch)
    char gitrev[256] = {0};
};

LtState g;

int fail(const char *what, int code)
{
    snprintf(g.err, sizeof(g.err), "%s (%d)", what, code);
    return -1;
}

template <class F> bool sym(F &f, const char *name)
{
    f = reinterpret_cast<F>(dlsym(g.api.so, name));
    return f != nullptr;
}

Key key_of(const atr_linear_args &a)
{
    return Key(a.M, a.N, a.K, a.batch, a.lda, a.ldw, a.ldc, a.stride_a, a.stride_w, a.stride_c, a.relu ? 1 : 0, a.bias ? 1 : 0);
}

// the workspace of a call: the caller's (two streams of one process must not share one: the pipelined schedule runs a
// learner's bootstrap step beside the other replica's rollout) or, when none is given, the library's own
void *ws_ptr(const atr_linear_args &a) { return a.workspace ? a.workspace : g.workspace; }
size_t ws_size(const atr_linear_args &a) { return a.workspace ? (size_t)a.workspace_bytes : kWorkspaceBytes; }

// Row-major C [M, N] = A [M, K] W[N, K]^T is, read column-major, D [N, M] (ld = ldc) = op_T(W' [K, N], ld = ldw) x A' [K, M]
// (ld = lda): the library's column-major convention with m = N, n = M, k = K; the bias runs along D's rows = C's columns.
int build_plan(const atr_linear_args &a, Plan &p)
{
    LtApi &L = g.api;
    int rc;
    if ((rc = L.DescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F)) != 0) return fail("hipblasLtMatmulDescCreate", rc);
    const int32_t opT = HIPBLAS_OP_T, opN = HIPBLAS_OP_N;
    L.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opT, sizeof(opT));
    L.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opN, sizeof(opN));
    uint32_t epi = HIPBLASLT_EPILOGUE_DEFAULT;
    if (a.bias && a.relu) epi = HIPBLASLT_EPILOGUE_RELU_BIAS;
    else if (a.bias) epi = HIPBLASLT_EPILOGUE_BIAS;
    else if (a.relu) epi = HIPBLASLT_EPILOGUE_RELU;
    if ((rc = L.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof(epi))) != 0) return fail("epilogue", rc);
    if (a.bias) {
        const int32_t bt = HIP_R_32F;
        L.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt));
    }
    if ((rc = L.LayoutCreate(&p.la, HIP_R_32F, (uint64_t)a.K, (uint64_t)a.N, a.ldw)) != 0) return fail("layout W", rc);
    if ((rc = L.LayoutCreate(&p.lb, HIP_R_32F, (uint64_t)a.K, (uint64_t)a.M, a.lda)) != 0) return fail("layout A", rc);
    if ((rc = L.LayoutCreate(&p.ld, HIP_R_32F, (uint64_t)a.N, (uint64_t)a.M, a.ldc)) != 0) return fail("layout C", rc);
    if (a.batch > 1) {
        const int32_t bc = a.batch;
        const int64_t sw = a.stride_w, sa = a.stride_a, sc = a.stride_c;
        L.LayoutSet(p.la, HIPBLASLT_MATRIX_LAYOUT_BATCH_COUNT, &bc, sizeof(bc));
        L.LayoutSet(p.lb, HIPBLASLT_MATRIX_LAYOUT_BATCH_COUNT, &bc, sizeof(bc));
        L.LayoutSet(p.ld, HIPBLASLT_MATRIX_LAYOUT_BATCH_COUNT, &bc, sizeof(bc));
        L.LayoutSet(p.la, HIPBLASLT_MATRIX_LAYOUT_STRIDED_BATCH_OFFSET, &sw, sizeof(sw));
        L.LayoutSet(p.lb, HIPBLASLT_MATRIX_LAYOUT_STRIDED_BATCH_OFFSET, &sa, sizeof(sa));
        L.LayoutSet(p.ld, HIPBLASLT_MATRIX_LAYOUT_STRIDED_BATCH_OFFSET, &sc, sizeof(sc));
    }
    return 0;
}

int candidates(const atr_linear_args &a, Plan &p, std::vector<hipblasLtMatmulHeuristicResult_t> &res)
{
    LtApi &L = g.api;
    if (a.bias) L.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &a.bias, sizeof(a.bias));
    hipblasLtMatmulPreference_t pref = nullptr;
    int rc = L.PrefCreate(&pref);
    if (rc != 0) return fail("hipblasLtMatmulPreferenceCreate", rc);
    const uint64_t wsz = ws_size(a);
    L.PrefSet(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsz, sizeof(wsz));
    res.resize(kMaxAlgos);
    int n = 0;
    rc = L.Heuristic(g.handle, p.desc, p.la, p.lb, p.ld, p.ld, pref, kMaxAlgos, res.data(), &n);
    L.PrefDestroy(pref);
    if (rc != 0 || n <= 0) return fail("hipblasLtMatmulAlgoGetHeuristic: no kernel for this problem", rc);
    res.resize(n);
    return 0;
}

int run(const atr_linear_args &a, Plan &p, const hipblasLtMatmulAlgo_t &algo, size_t ws_bytes, hipStream_t st)
{
    LtApi &L = g.api;
    if (a.bias) L.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &a.bias, sizeof(a.bias));
    const float one = 1.0f, zero = 0.0f;
    return L.Matmul(g.handle, p.desc, &one, a.w, p.la, a.a, p.lb, &zero, a.c, p.ld, a.c, p.ld, &algo, ws_ptr(a),
                    ws_bytes ? ws_size(a) : 0, st);
}

bool capturing(hipStream_t st)
{
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return false; }
    return cs != hipStreamCaptureStatusNone;
}

int check_args(const atr_linear_args *a)
{
    if (!a || !a->a || !a->w || !a->c) return fail("atr_linear: null argument", 0);
    if (a->M <= 0 || a->N <= 0 || a->K <= 0 || a->batch < 1) return fail("atr_linear: bad shape", 0);
    if (a->lda < a->K || a->ldw < a->K || a->ldc < a->N) return fail("atr_linear: row stride shorter than the row", 0);
    if (a->bias && a->batch > 1) return fail("atr_linear: a bias needs batch == 1 (the library has no per-batch bias stride)", 0);
    if (!g.handle) return fail("atr_linear: call atr_lt_init first", 0);
    if (a->workspace && (((uintptr_t)a->workspace & 15u) || a->workspace_bytes < 0)) return fail("atr_linear: workspace must be 16-byte aligned", 0);
    return 0;
}

int solution_of(hipblasLtMatmulAlgo_t &algo) { return g.api.IndexFromAlgo ? g.api.IndexFromAlgo(algo) : -1; }

// Does the library we loaded agree with the header this file was compiled against? PyTorch ships its own libhipblaslt.so
// (another release than /opt/rocm's hipblaslt.h); comparing version numbers would either refuse a library that works or accept
// one that does not, so the check is functional: one 32 x 32 x 32 product with the RELU_BIAS epilogue, a row stride on the
// output and small integer operands (exact in any summation order) against the host's loops. It exercises every enum value and
// struct layout this file relies on (descriptor attributes, epilogue codes, the heuristic result record, the algo record).
int self_test()
{
    constexpr int n = 32, ldc = 40;
    std::vector<float> ha(n * n), hw(n * n), hb(n), hc(n * ldc, -7.f), want(n * ldc, -7.f);
    for (int i = 0; i < n * n; i++) { ha[i] = (float)((i * 7 + 3) % 5 - 2); hw[i] = (float)((i * 11 + 1) % 7 - 3); }
    for (int i = 0; i < n; i++) hb[i] = (float)(i % 9 - 4);
    for (int m = 0; m < n; m++)
        for (int j = 0; j < n; j++) {
            float acc = hb[j];
            for (int k = 0; k < n; k++) acc += ha[m * n + k] * hw[j * n + k];
            want[m * ldc + j] = acc > 0.f ? acc : 0.f;
        }
    float *da = nullptr, *dw = nullptr, *db = nullptr, *dc = nullptr;
    hipStream_t st = nullptr;
    int rc = -1;
    do {
        if (hipMalloc(&da, ha.size() * 4) != hipSuccess || hipMalloc(&dw, hw.size() * 4) != hipSuccess
            || hipMalloc(&db, hb.size() * 4) != hipSuccess || hipMalloc(&dc, hc.size() * 4) != hipSuccess) break;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) break;
        (void)hipMemcpy(da, ha.data(), ha.size() * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(dc, hc.data(), hc.size() * 4, hipMemcpyHostToDevice);
        atr_linear_args a;
        memset(&a, 0, sizeof(a));
        a.a = da; a.w = dw; a.bias = db; a.c = dc;
        a.lda = n; a.ldw = n; a.ldc = ldc;
        a.M = n; a.N = n; a.K = n; a.batch = 1; a.relu = 1;
        Plan p;
        if (build_plan(a, p) != 0) break;
        std::vector<hipblasLtMatmulHeuristicResult_t> res;
        bool ran = false;
        if (candidates(a, p, res) == 0)
            for (size_t i = 0; i < res.size() && !ran; i++)
                if (res[i].state == HIPBLAS_STATUS_SUCCESS && res[i].workspaceSize <= ws_size(a))
                    ran = run(a, p, res[i].algo, res[i].workspaceSize, st) == 0;
        g.api.LayoutDestroy(p.la); g.api.LayoutDestroy(p.lb); g.api.LayoutDestroy(p.ld); g.api.DescDestroy(p.desc);
        if (!ran || hipStreamSynchronize(st) != hipSuccess) break;
        (void)hipMemcpy(hc.data(), dc, hc.size() * 4, hipMemcpyDeviceToHost);
        rc = memcmp(hc.data(), want.data(), hc.size() * 4) == 0 ? 0 : -2;
    } while (0);
    if (st) (void)hipStreamDestroy(st);
    (void)hipFree(da); (void)hipFree(dw); (void)hipFree(db); (void)hipFree(dc);
    (void)hipGetLastError();
    return rc;
}

// the plan of this problem, built and (outside a capture) tuned on first use
int plan_for(const atr_linear_args &a, hipStream_t st, Plan **out)
{
    const Key k = key_of(a);
    auto it = g.plans.find(k);
    if (it == g.plans.end()) {
        Plan p;
        if (build_plan(a, p) != 0) return -1;
        it = g.plans.emplace(k, p).first;
    }
    Plan &p = it->second;
    if (p.preset >= 0 && p.chosen < 0) {       // a recorded choice (atr_linear_set_choice): candidate number `preset`, no timing
        std::vector<hipblasLtMatmulHeuristicResult_t> res;
        if (candidates(a, p, res) != 0) return -1;
        p.candidates = (int)res.size();
        auto usable = [&](int i) {
            return i >= 0 && i < (int)res.size() && res[i].state == HIPBLAS_STATUS_SUCCESS && res[i].workspaceSize <= ws_size(a);
        };
        // The record is (position in the heuristic's list, solution index). The position alone means nothing on another build
        // of the library (or another workspace limit): it is only taken when the kernel AT that position is the recorded one;
        // otherwise the list is searched for the recorded solution; failing that the record is refused and the list is timed
        // as if nothing had been recorded (below) — slower start, never an untimed kernel marked as tuned.
        int take = -1;
        if (p.preset_solution < 0 || !g.api.IndexFromAlgo) {
            if (usable(p.preset)) take = p.preset;
        } else {
            if (usable(p.preset) && solution_of(res[p.preset].algo) == p.preset_solution) take = p.preset;
            for (int i = 0; i < (int)res.size() && take < 0; i++)
                if (usable(i) && solution_of(res[i].algo) == p.preset_solution) take = i;
        }
        if (take >= 0) {
            p.algo = res[take].algo;
            p.ws_bytes = res[take].workspaceSize;
            p.chosen = take;
            p.solution = solution_of(p.algo);
            p.tuned = true;                    // (as good as tuned: the choice was timed when it was recorded)
            p.source = ATR_LT_SOURCE_RECORDED;
        } else {
            p.source = ATR_LT_SOURCE_REFUSED;  // (replaced by TIMED / FIRST_USABLE right below)
        }
        p.preset = -1;
        p.preset_solution = -1;
    }
    if (p.chosen < 0 || (!p.tuned && !capturing(st))) {
        std::vector<hipblasLtMatmulHeuristicResult_t> res;
        if (candidates(a, p, res) != 0) return -1;
        p.candidates = (int)res.size();
        int best = -1;
        float best_ms = 1e30f;
        if (capturing(st)) {
            for (int i = 0; i < (int)res.size() && best < 0; i++)
                if (res[i].state == HIPBLAS_STATUS_SUCCESS && res[i].workspaceSize <= ws_size(a)) best = i;
        } else {
            // time every candidate on the caller's stream, on the caller's buffers (C is overwritten with the right
            // result every time): a first pass of 2 untimed + 6 timed launches each, then the five fastest again with 40
            // timed launches (a 6-launch sample is too noisy to tell kernels within 10 % of each other apart)
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            auto time_algo = [&](int i, int reps) -> float {
                bool ok = true;
                for (int w = 0; w < 2 && ok; w++) ok = run(a, p, res[i].algo, res[i].workspaceSize, st) == 0;
                if (!ok) return -1.f;
                (void)hipEventRecord(e0, st);
                for (int w = 0; w < reps && ok; w++) ok = run(a, p, res[i].algo, res[i].workspaceSize, st) == 0;
                (void)hipEventRecord(e1, st);
                if (hipEventSynchronize(e1) != hipSuccess || !ok) { (void)hipGetLastError(); return -1.f; }
                float ms = 0.f;
                (void)hipEventElapsedTime(&ms, e0, e1);
                return ms / (float)reps;
            };
            std::vector<std::pair<float, int>> first;
            for (int i = 0; i < (int)res.size(); i++) {
                if (res[i].state != HIPBLAS_STATUS_SUCCESS || res[i].workspaceSize > ws_size(a)) continue;
                const float ms = time_algo(i, 6);
                if (ms >= 0.f) first.push_back(std::make_pair(ms, i));
            }
            std::sort(first.begin(), first.end());
            for (int j = 0; j < (int)first.size() && j < 5; j++) {
                const float ms = time_algo(first[j].second, 40);
                if (ms >= 0.f && ms < best_ms) { best_ms = ms; best = first[j].second; }
            }
            (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
            p.tuned = best >= 0;
            p.best_us = best >= 0 ? best_ms * 1e3f : 0.f;
        }
        if (best < 0) return fail("atr_linear: no usable hipBLASLt kernel", 0);
        p.algo = res[best].algo;
        p.ws_bytes = res[best].workspaceSize;
        p.chosen = best;
        p.solution = solution_of(p.algo);
        const bool refused = p.source == ATR_LT_SOURCE_REFUSED || p.source == ATR_LT_SOURCE_REFUSED_TIMED;
        p.source = p.tuned ? (refused ? ATR_LT_SOURCE_REFUSED_TIMED : ATR_LT_SOURCE_TIMED) : ATR_LT_SOURCE_FIRST_USABLE;
    }
    *out = &p;
    return 0;
}

} // namespace

extern "C" const char *atr_lt_last_error(void) { return g.err; }

extern "C" int atr_lt_init(const char *libhipblaslt_path)
{
    std::lock_guard<std::mutex> lock(g.mu);
    if (g.handle) return 0;
    if (!libhipblaslt_path || !*libhipblaslt_path) return fail("atr_lt_init: path of PyTorch's libhipblaslt.so required", 0);
    g.api.so = dlopen(libhipblaslt_path, RTLD_NOW | RTLD_LOCAL);
    if (!g.api.so) { snprintf(g.err, sizeof(g.err), "dlopen(%s): %s", libhipblaslt_path, dlerror()); return -1; }
    LtApi &L = g.api;
    const bool ok = sym(L.Create, "hipblasLtCreate") && sym(L.LayoutCreate, "hipblasLtMatrixLayoutCreate")
        && sym(L.LayoutDestroy, "hipblasLtMatrixLayoutDestroy") && sym(L.LayoutSet, "hipblasLtMatrixLayoutSetAttribute")
        && sym(L.DescCreate, "hipblasLtMatmulDescCreate") && sym(L.DescDestroy, "hipblasLtMatmulDescDestroy")
        && sym(L.DescSet, "hipblasLtMatmulDescSetAttribute") && sym(L.PrefCreate, "hipblasLtMatmulPreferenceCreate")
        && sym(L.PrefDestroy, "hipblasLtMatmulPreferenceDestroy") && sym(L.PrefSet, "hipblasLtMatmulPreferenceSetAttribute")
        && sym(L.Heuristic, "hipblasLtMatmulAlgoGetHeuristic") && sym(L.Matmul, "hipblasLtMatmul");
    if (!ok) return fail("atr_lt_init: libhipblaslt.so lacks an entry point", 0);
    (void)sym(L.GetVersion, "hipblasLtGetVersion");
    (void)sym(L.GetGitRevision, "hipblasLtGetGitRevision");
    (void)sym(L.IndexFromAlgo, "_ZN13hipblaslt_ext16getIndexFromAlgoER22_hipblasLtMatmulAlgo_t");
    (void)sym(L.KernelNameFromAlgo, "_ZN13hipblaslt_ext21getKernelNameFromAlgoB5cxx11EPvR22_hipblasLtMatmulAlgo_t");
    int rc = L.Create(&g.handle);
    if (rc != 0) { g.handle = nullptr; return fail("hipblasLtCreate", rc); }
    if (hipMalloc(&g.workspace, kWorkspaceBytes) != hipSuccess) { g.handle = nullptr; return fail("hipMalloc(workspace)", 0); }
    if (L.GetVersion) (void)L.GetVersion(g.handle, &g.version);
    if (L.GetGitRevision) (void)L.GetGitRevision(g.handle, g.gitrev);
    g.gitrev[sizeof(g.gitrev) - 1] = 0;
    rc = self_test();
    if (rc != 0) {
        // the loaded library does not behave as the header this file was compiled against says: leave the direct path off
        // (fused.lt_available() is then False and the model keeps torch's GEMMs)
        (void)hipFree(g.workspace);
        g.workspace = nullptr;
        g.handle = nullptr;
        snprintf(g.err, sizeof(g.err), "atr_lt_init: self-test %s with libhipblaslt %d (header %d.%d.%d)",
                 rc == -2 ? "computed a wrong product" : "could not run", g.version, HIPBLASLT_VERSION_MAJOR,
                 HIPBLASLT_VERSION_MINOR, HIPBLASLT_VERSION_PATCH);
        return -1;
    }
    return 0;
}

// version: hipblasLtGetVersion of the loaded library; header_version: the same encoding for the header this file was built
// against; gitrev: the loaded library's revision string (up to len - 1 characters). Returns 0, or -1 before atr_lt_init.
extern "C" int atr_lt_library_info(int *version, int *header_version, char *gitrev, int len)
{
    std::lock_guard<std::mutex> lock(g.mu);
    if (!g.handle) return -1;
    if (version) *version = g.version;
    if (header_version) *header_version = HIPBLASLT_VERSION_MAJOR * 100000 + HIPBLASLT_VERSION_MINOR * 100 + HIPBLASLT_VERSION_PATCH;
    if (gitrev && len > 0) { strncpy(gitrev, g.gitrev, (size_t)len - 1); gitrev[len - 1] = 0; }
    return 0;
}

// the kernel name of this problem's current choice (hipblaslt_ext::getKernelNameFromAlgo; "" when the library does not say)
extern "C" int atr_linear_kernel_name(const atr_linear_args *a, char *out, int len)
{
    std::lock_guard<std::mutex> lock(g.mu);
    if (!a || !out || len <= 0) return -1;
    out[0] = 0;
    auto it = g.plans.find(key_of(*a));
    if (it == g.plans.end() || it->second.chosen < 0) return -1;
    if (g.api.KernelNameFromAlgo) {
        const std::string s = g.api.KernelNameFromAlgo((void *)g.handle, it->second.algo);
        strncpy(out, s.c_str(), (size_t)len - 1);
        out[len - 1] = 0;
    }
    return 0;
}

extern "C" int atr_linear(const atr_linear_args *a, void *stream)
{
    std::lock_guard<std::mutex> lock(g.mu);
    if (check_args(a) != 0) return -1;
    hipStream_t st = (hipStream_t)stream;
    Plan *p = nullptr;
    if (plan_for(*a, st, &p) != 0) return -1;
    if (p->ws_bytes > ws_size(*a)) return fail("atr_linear: workspace smaller than the one this problem was tuned with", 0);
    const int rc = run(*a, *p, p->algo, p->ws_bytes, st);
    return rc == 0 ? 0 : fail("hipblasLtMatmul", rc);
}

// Use candidate number `index` of the library's list for this problem instead of timing the list (a choice recorded by an
// earlier run: lt_tuning_gfx950.json; or another one for launches that will run next to other work). solution >= 0: the
// recorded kernel's solution index — the position is only believed when the kernel there IS that solution, else the list is
// searched for it. If the list turns out shorter, the candidate unusable or the solution absent, the record is refused and
// the next call times the list as if nothing had been selected (plan_info's source says so).
extern "C" int atr_linear_set_choice(const atr_linear_args *a, int index, int solution)
{
    std::lock_guard<std::mutex> lock(g.mu);
    if (check_args(a) != 0) return -1;
    const Key k = key_of(*a);
    auto it = g.plans.find(k);
    if (it == g.plans.end()) {
        Plan p;
        if (build_plan(*a, p) != 0) return -1;
        it = g.plans.emplace(k, p).first;
    }
    if (index < 0) return fail("atr_linear_set_choice: negative index", 0);
    it->second.preset = index;                 // (also over an earlier choice: the next call resolves the candidate list again)
    it->second.preset_solution = solution;
    it->second.chosen = -1;
    it->second.tuned = false;
    return 0;
}

extern "C" int atr_linear_plan_info(const atr_linear_args *a, int *candidates_out, int *chosen_out, int *tuned_out, float *best_us_out,
                                    int *solution_out, int *source_out)
{
    std::lock_guard<std::mutex> lock(g.mu);
    if (!a) return -1;
    auto it = g.plans.find(key_of(*a));
    if (it == g.plans.end()) return -1;
    if (candidates_out) *candidates_out = it->second.candidates;
    if (chosen_out) *chosen_out = it->second.chosen;
    if (tuned_out) *tuned_out = it->second.tuned ? 1 : 0;
    if (best_us_out) *best_us_out = it->second.best_us;
    if (solution_out) *solution_out = it->second.solution;
    if (source_out) *source_out = it->second.source;
    return 0;
}

#else   // no hipblaslt.h on the build box: the rest of the library still builds; fused.lt_available() is False at run time and
        // the model keeps torch's GEMMs for these layers

static const char *const kNoLt = "built without <hipblaslt/hipblaslt.h>: the direct hipBLASLt path is not available";
extern "C" const char *atr_lt_last_error(void) { return kNoLt; }
extern "C" int atr_lt_init(const char *) { return -1; }
extern "C" int atr_linear(const atr_linear_args *, void *) { return -1; }
extern "C" int atr_linear_set_choice(const atr_linear_args *, int, int) { return -1; }
extern "C" int atr_linear_plan_info(const atr_linear_args *, int *, int *, int *, float *, int *, int *) { return -1; }
extern "C" int atr_lt_library_info(int *, int *, char *, int) { return -1; }
extern "C" int atr_linear_kernel_name(const atr_linear_args *, char *, int) { return -1; }

#endif
