// coop_gemm.h — the GEMM engine of the small-shard rollout step (k_coop_step, csrc/track2d_hip.hip): one workgroup of 8 waves
// computes a handful of 16-row x 32-column output tiles ("units", C = act(A W^T + bias), nn.Linear layout, exact f32 on
// v_mfma_f32_16x16x4_f32) whose contractions are laid end to end and cut into 8 equal pieces, one per wave.
//
// Why this shape. At the strong-scaling shard sizes (512 / 1024 envs per GPU) a layer of the policy (perception.py:81,90 and
// model.py:110,137,172,203 of the reference: the encoder's fc + ReLU, nn.LSTMCell's two GEMMs) is 0.4 GFLOP = 2.6 us of the
// chip's f32 matrix pipes — IF all 256 CUs work on it, which a tile must be small for: 16 rows x 32 columns gives a 512-env
// layer 64 + 64 units of different K (512 and 1024 for the two encoders) that pair up into equal work for the 32 workgroups
// of an XCD. Inside the workgroup the pieces must be equal again — a wave is a serial chain of MFMAs — so the K ranges of
// the workgroup's units are concatenated and every wave takes one eighth of the total, whichever units that crosses
// (at most kCoopMaxSeg of them); the partial tiles meet in LDS and are summed in wave order = ascending K (a fixed order:
// results do not depend on timing).
//
// Operands: a lane that fetched its own MFMA operands (16 bytes of row r at k-slot q) would touch 16 different rows per
// quarter-wave — 64 cache-line lookups for 1 KB, which makes the CU's one L1 the bottleneck (measured: the first version of
// this file did, and a layer took 3x its matrix-pipe time). So rows are fetched the way they lie in memory — 8 lanes x 16 B
// cover one 128-byte line of a row, a wave instruction fetches 8 rows x 32 k — parked in a WAVE-PRIVATE LDS slice (36-float
// row stride: conflict-free 16-byte reads in MFMA layout; no workgroup barrier: DS operations of one wave execute in order)
// and read back as lane (r = l & 15, q = l >> 4) -> 16 bytes of row r at k0 + 4 q: the four K slots of one 16x16x4 MFMA are
// {k0 + j, k0 + 4 + j, k0 + 8 + j, k0 + 12 + j} for component j (A and B agree on it, which is all a contraction needs).
// Loads run kCoopSlots - 1 K blocks ahead of the MFMAs in registers: with one workgroup per CU nothing else hides latency.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace atr {

typedef float coop_f32x4 __attribute__((ext_vector_type(4)));

struct CoopUnit {
    const float *a;      // first of the tile's 16 rows (row stride lda floats; K contiguous)
    const float *w;      // first of the tile's 32 weight rows ([out, in] layout, row stride ldw)
    const float *bias;   // nullable: 32 values at the tile's first column
    float *c;            // first output element (row stride ldc)
    int lda, ldw, ldc;
    int K;               // multiple of kCoopBlk (32)
    int rows_valid;      // 1..16: rows beyond shadow the last valid one on the way in and are not stored
    int relu;
};

constexpr int kCoopWaves = 8;
constexpr int kCoopThreads = 64 * kCoopWaves;
constexpr int kCoopMaxUnits = 16;
constexpr int kCoopMaxSeg = 2;         // units one wave's share of the concatenated K range may touch
constexpr int kCoopBlk = 32;           // K block staged per trip (two 16-wide MFMA iterations)
constexpr int kCoopLs = kCoopBlk + 4;  // LDS row stride of a staged block (floats)
constexpr int kCoopStage = 48 * kCoopLs;   // one wave's staged block: 16 A rows + 32 W rows
constexpr int kCoopSlots = 6;          // register slots of the pipeline: kCoopSlots - 1 K blocks in flight ahead of the MFMAs
constexpr int kCoopTile = 16 * 32;     // floats of one partial tile

struct CoopLds {                        // the engine's LDS (the caller places it; 16-byte aligned)
    CoopUnit units[kCoopMaxUnits];
    int first_unit[kCoopWaves];         // per wave: the first unit its share touches, and how many
    int n_seg[kCoopWaves];
    float part[kCoopWaves * kCoopMaxSeg * kCoopTile];
    float stage[kCoopWaves * kCoopStage];   // per wave: the K block being multiplied
    float bias[kCoopMaxUnits * 32];         // the units' bias rows (zeros where a unit has none)
    int contrib[kCoopMaxUnits * kCoopWaves];   // per unit: the partial-tile slots that hold a piece of it, in K order
    int n_contrib[kCoopMaxUnits];
    unsigned long long *probe;              // nullable: wave 0's clock at {enter, first block parked, loop end, all waves done, stored}
};

__device__ __forceinline__ float4 coop_ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
// The unit descriptors travel through LDS, where a pointer loses its address space: loaded through it, the operands would be
// FLAT loads, which count in lgkmcnt as well as vmcnt and return in no fixed order relative to other memory types — every
// LDS wait of the pipeline would drain the global loads too. Said explicitly: these are global-memory addresses.
typedef const float __attribute__((address_space(1))) *coop_gptr;
typedef float coop_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 coop_ldg4(const float *p)
{
    const coop_v4f v = *reinterpret_cast<const coop_v4f __attribute__((address_space(1))) *>((coop_gptr)p);
    return make_float4(v[0], v[1], v[2], v[3]);
}

// The engine runs in stages so that a caller can put a barrier between "the weights are requested" and "the activations are
// requested" (k_coop_step: the layer's input is what the OTHER workgroups of the XCD are still writing, its weights are not):
//   coop_plan     (after the units are in L.units and a __syncthreads()): cuts the concatenated K range, per wave
//   coop_fetch_w / coop_fetch_a   request the W / A part of the first kCoopSlots K blocks
//   coop_run      the pipeline, the partial tiles, the fixed-order sum, bias + activation, the stores
// coop_gemm = all of them back to back.
//
// A wave's share is a run of 32-wide K blocks of the concatenated range, possibly crossing from one unit into the next. It is
// ONE software pipeline across that crossing: kCoopSlots register slots, every trip of the (branch-free as far as memory
// operations go) loop parks the oldest slot in LDS, re-requests it kCoopSlots blocks ahead, and multiplies — so the
// compiler's wait counts are static (vmcnt = everything but the oldest slot) and 5 blocks = 30 KB per wave stay in flight.
// Blocks past the end of the share are fetched again from its last block and not multiplied.
struct CoopPipe {
    int wave, l, lo, hi, n_seg, first;
    int seg_end[kCoopMaxSeg];
    // per touched unit: wave-uniform base addresses of its A rows / W rows, moved back by the unit's first block so that global
    // block g sits at base + g * 128 bytes; and this lane's BYTE offsets from them — loop-invariant, 32-bit: an operand load is
    // global_load_dwordx4 v, v_offset, s[base] with the block's displacement added on the scalar side
    const char *pa[kCoopMaxSeg], *pw[kCoopMaxSeg];
    unsigned oa0[kCoopMaxSeg], oa1[kCoopMaxSeg], ow[kCoopMaxSeg][4];
    float4 ra[kCoopSlots][2], rw[kCoopSlots][4];
    float bias;          // this thread's element of the bias table (unit tid >> 5, column tid & 31), requested at plan time
    bool ok;
};

__device__ __forceinline__ const char *coop_uni_ptr(const void *p)
{
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const char *)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ float4 coop_ldg4_at(const char *base, unsigned off)
{
    typedef const char __attribute__((address_space(1))) *gbytes;
    const coop_v4f v = *reinterpret_cast<const coop_v4f __attribute__((address_space(1))) *>((gbytes)base + off);
    return make_float4(v[0], v[1], v[2], v[3]);
}

__device__ __forceinline__ void coop_plan(CoopLds &L, int n_units, int tid, CoopPipe &P)
{
    P.wave = __builtin_amdgcn_readfirstlane(tid >> 6); P.l = tid & 63;
    int total = 0;
    for (int u = 0; u < n_units; u++) total += L.units[u].K / kCoopBlk;      // (the cut points are K blocks)
    total = __builtin_amdgcn_readfirstlane(total);
    P.lo = (int)(((long long)total * P.wave) / kCoopWaves); P.hi = (int)(((long long)total * (P.wave + 1)) / kCoopWaves);
    // the (at most kCoopMaxSeg) units this share touches — staging roles: lane (rs = l >> 3, k4 = l & 7) fetches
    // float4 X[rs + 8 i][32 blk + 4 k4]
    const int rs = P.l >> 3, k4 = P.l & 7;
    P.n_seg = 0; P.first = 0; P.ok = true;
    int base = 0;
#pragma unroll
    for (int sg = 0; sg < kCoopMaxSeg; sg++) {
        P.seg_end[sg] = 0x7fffffff; P.pa[sg] = coop_uni_ptr(L.units[0].a); P.pw[sg] = coop_uni_ptr(L.units[0].w);
        P.oa0[sg] = P.oa1[sg] = 0;
        P.ow[sg][0] = P.ow[sg][1] = P.ow[sg][2] = P.ow[sg][3] = 0;
    }
#pragma unroll 1
    for (int u = 0; u < n_units; u++) {
        const int its = __builtin_amdgcn_readfirstlane(L.units[u].K / kCoopBlk);
        const int b0 = max(P.lo, base), b1 = min(P.hi, base + its);
        if (b0 < b1) {
            if (P.n_seg >= kCoopMaxSeg) { P.ok = false; break; }
            const CoopUnit U = L.units[u];
            if (P.n_seg == 0) P.first = u;
            const long long back = -(long long)base * kCoopBlk * 4;           // global block g of this unit starts at (g - base) * 128 B
#pragma unroll
            for (int sg = 0; sg < kCoopMaxSeg; sg++)
                if (sg == P.n_seg) {
                    P.seg_end[sg] = b1;
                    P.pa[sg] = coop_uni_ptr(U.a) + back; P.pw[sg] = coop_uni_ptr(U.w) + back;
                    const int lda = __builtin_amdgcn_readfirstlane(U.lda), ldw = __builtin_amdgcn_readfirstlane(U.ldw);
                    const int rv = __builtin_amdgcn_readfirstlane(U.rows_valid);
                    P.oa0[sg] = 4u * (unsigned)(min(rs, rv - 1) * lda + 4 * k4);
                    P.oa1[sg] = 4u * (unsigned)(min(rs + 8, rv - 1) * lda + 4 * k4);
#pragma unroll
                    for (int i = 0; i < 4; i++) P.ow[sg][i] = 4u * (unsigned)((rs + 8 * i) * ldw + 4 * k4);
                }
            P.n_seg++;
        }
        base += its;
    }
    // the bias element this thread will add at the end (a cold global load: requested now, used after the pipeline)
    {
        const int bu = min(tid >> 5, max(n_units - 1, 0));
        const float *b = L.units[bu].bias;
        const float bv = *(coop_gptr)((b ? b : L.units[0].w) + (tid & 31));      // (always a valid address: one load, no branch)
        P.bias = b ? bv : 0.f;
    }
}

static_assert(kCoopMaxSeg == 2, "the segment select below is written for two");
template <int SLOT> __device__ __forceinline__ void coop_fetch_w(CoopPipe &P, int blk)
{
    const int g = max(min(blk, P.hi - 1), 0);               // past the end: the last block again (not used)
    const bool s1 = g >= P.seg_end[0];                      // (wave-uniform)
    const char *w = (s1 ? P.pw[1] : P.pw[0]) + (long long)g * (kCoopBlk * 4);
#pragma unroll
    for (int i = 0; i < 4; i++) P.rw[SLOT][i] = coop_ldg4_at(w, s1 ? P.ow[1][i] : P.ow[0][i]);
}
template <int SLOT> __device__ __forceinline__ void coop_fetch_a(CoopPipe &P, int blk)
{
    const int g = max(min(blk, P.hi - 1), 0);
    const bool s1 = g >= P.seg_end[0];
    const char *a = (s1 ? P.pa[1] : P.pa[0]) + (long long)g * (kCoopBlk * 4);
    P.ra[SLOT][0] = coop_ldg4_at(a, s1 ? P.oa0[1] : P.oa0[0]);
    P.ra[SLOT][1] = coop_ldg4_at(a, s1 ? P.oa1[1] : P.oa1[0]);
}
__device__ __forceinline__ void coop_prefetch_w(CoopPipe &P)
{
    // (a wave without a share fetches the first block of the first unit: valid addresses, never multiplied)
    coop_fetch_w<0>(P, P.lo); coop_fetch_w<1>(P, P.lo + 1); coop_fetch_w<2>(P, P.lo + 2);
    coop_fetch_w<3>(P, P.lo + 3); coop_fetch_w<4>(P, P.lo + 4); coop_fetch_w<5>(P, P.lo + 5);
}
__device__ __forceinline__ void coop_prefetch_a(CoopPipe &P)
{
    coop_fetch_a<0>(P, P.lo); coop_fetch_a<1>(P, P.lo + 1); coop_fetch_a<2>(P, P.lo + 2);
    coop_fetch_a<3>(P, P.lo + 3); coop_fetch_a<4>(P, P.lo + 4); coop_fetch_a<5>(P, P.lo + 5);
}

// Returns false if a wave's share touched more than kCoopMaxSeg units (nothing is stored for the overflow: the caller flags a
// fault). Ends with the tiles stored (plain stores: the caller makes them visible — a kernel boundary or an XCD barrier).
__device__ __forceinline__ bool coop_run(CoopLds &L, int n_units, int tid, CoopPipe &P)
{
    const int wave = P.wave, l = P.l, r = l & 15, q = l >> 4, rs = l >> 3, k4 = l & 7;
    const int lo = P.lo, hi = P.hi;
    float *st = L.stage + (size_t)wave * kCoopStage;
    float *stp = st + rs * kCoopLs + 4 * k4;                          // where this lane parks what it fetched
    const float *rda = st + r * kCoopLs + 4 * q;                      // MFMA-layout reads: A rows 0..15, W rows 16..47
    const float *rdb0 = st + (16 + r) * kCoopLs + 4 * q, *rdb1 = st + (32 + r) * kCoopLs + 4 * q;
    coop_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#define COOP_ST4(dst, v) (*reinterpret_cast<float4 *>(dst) = (v))
#define COOP_PARK(SLOT)                                                                                                \
    do {                                                                                                               \
        COOP_ST4(stp, P.ra[SLOT][0]); COOP_ST4(stp + 8 * kCoopLs, P.ra[SLOT][1]);                                      \
        COOP_ST4(stp + 16 * kCoopLs, P.rw[SLOT][0]); COOP_ST4(stp + 24 * kCoopLs, P.rw[SLOT][1]);                      \
        COOP_ST4(stp + 32 * kCoopLs, P.rw[SLOT][2]); COOP_ST4(stp + 40 * kCoopLs, P.rw[SLOT][3]);                      \
    } while (0)
#define COOP_WAVE_SYNC()                                                                                               \
    do {                                                                                                               \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();                        \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                                                         \
    } while (0)
    // D[4 q + i][r] of the two 16x16 blocks -> this wave's partial tile [row][col] (32 floats per row) of segment SEG
#define COOP_DUMP(SEG)                                                                                                 \
    do {                                                                                                               \
        float *pt_ = L.part + (size_t)(wave * kCoopMaxSeg + (SEG)) * kCoopTile + (4 * q) * 32 + r;                     \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; i_++) { pt_[i_ * 32] = acc0[i_]; pt_[i_ * 32 + 16] = acc1[i_]; }    \
    } while (0)
#define COOP_TRIP(SLOT, BLK)                                                                                           \
    do {                                                                                                               \
        COOP_PARK(SLOT);                                                                                               \
        coop_fetch_a<SLOT>(P, (BLK) + kCoopSlots); coop_fetch_w<SLOT>(P, (BLK) + kCoopSlots);                          \
        if ((BLK) < hi) {                                   /* (wave-uniform; no global-memory operation inside) */    \
            if ((BLK) == P.seg_end[0]) {                    /* the share crosses into its second unit here */          \
                COOP_DUMP(0);                                                                                          \
                acc0 = coop_f32x4{0.f, 0.f, 0.f, 0.f}; acc1 = coop_f32x4{0.f, 0.f, 0.f, 0.f};                          \
            }                                                                                                          \
            COOP_WAVE_SYNC();                                                                                          \
            _Pragma("unroll") for (int h_ = 0; h_ < 2; h_++) {                                                         \
                const float4 a_ = coop_ld4(rda + 16 * h_), x_ = coop_ld4(rdb0 + 16 * h_), y_ = coop_ld4(rdb1 + 16 * h_); \
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_.x, x_.x, acc0, 0, 0, 0);                                \
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_.x, y_.x, acc1, 0, 0, 0);                                \
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_.y, x_.y, acc0, 0, 0, 0);                                \
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_.y, y_.y, acc1, 0, 0, 0);                                \
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_.z, x_.z, acc0, 0, 0, 0);                                \
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_.z, y_.z, acc1, 0, 0, 0);                                \
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_.w, x_.w, acc0, 0, 0, 0);                                \
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_.w, y_.w, acc1, 0, 0, 0);                                \
            }                                                                                                          \
            COOP_WAVE_SYNC();                                                                                          \
        }                                                                                                              \
    } while (0)
    static_assert(kCoopSlots == 6, "the rotation below is written out for six register slots");
    unsigned long long *probe = L.probe;
    if (probe && tid == 0) probe[0] = wall_clock64();
#pragma unroll 1
    for (int b = lo; b < hi; b += kCoopSlots) {
        COOP_TRIP(0, b);
        if (probe && tid == 0 && b == lo) probe[1] = wall_clock64();
        COOP_TRIP(1, b + 1); COOP_TRIP(2, b + 2); COOP_TRIP(3, b + 3); COOP_TRIP(4, b + 4); COOP_TRIP(5, b + 5);
    }
    if (lo < hi) COOP_DUMP(P.n_seg - 1);
#undef COOP_ST4
#undef COOP_PARK
#undef COOP_WAVE_SYNC
#undef COOP_DUMP
#undef COOP_TRIP
    // wave w's segment sg holds a piece of unit first[w] + sg. A quarter of the workgroup per unit, four columns per thread:
    // the pieces are summed in wave order (= ascending K) with unconditional 16-byte LDS reads (a wave that holds no piece of
    // the unit reads its slot 0 and the value is dropped), then bias, activation and one 16-byte store per thread
    if (l == 0) { L.first_unit[wave] = P.first; L.n_seg[wave] = (lo < hi && P.ok) ? P.n_seg : 0; }
    L.bias[tid] = P.bias;
    if (probe && tid == 0) probe[2] = wall_clock64();
    __syncthreads();
    if (probe && tid == 0) probe[3] = wall_clock64();
    int fu[kCoopWaves], ns[kCoopWaves];
#pragma unroll
    for (int w = 0; w < kCoopWaves; w++) { fu[w] = L.first_unit[w]; ns[w] = L.n_seg[w]; }
    const int e = tid & 127, orow = e >> 3, oc4 = (e & 7) * 4;
#pragma unroll 1
    for (int u = tid >> 7; u < n_units; u += 4) {
        float4 pv[kCoopWaves];
        bool has[kCoopWaves];
#pragma unroll
        for (int w = 0; w < kCoopWaves; w++) {
            const int sg = u - fu[w];
            has[w] = sg >= 0 && sg < ns[w];
            pv[w] = coop_ld4(L.part + (size_t)(w * kCoopMaxSeg + (has[w] ? sg : 0)) * kCoopTile + orow * 32 + oc4);
        }
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        bool any = false;
#pragma unroll
        for (int w = 0; w < kCoopWaves; w++) {
            v.x += has[w] ? pv[w].x : 0.f; v.y += has[w] ? pv[w].y : 0.f; v.z += has[w] ? pv[w].z : 0.f; v.w += has[w] ? pv[w].w : 0.f;
            any = any || has[w];
        }
        const float4 b = coop_ld4(L.bias + u * 32 + oc4);
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        const CoopUnit &U = L.units[u];
        if (U.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (orow < U.rows_valid && any) {
            typedef float __attribute__((address_space(1))) *gout;
            coop_v4f o = {v.x, v.y, v.z, v.w};
            *reinterpret_cast<coop_v4f __attribute__((address_space(1))) *>((gout)(U.c + (size_t)orow * U.ldc + oc4)) = o;
        }
    }
    if (probe && tid == 0) probe[4] = wall_clock64();
    return P.ok;
}

__device__ __forceinline__ bool coop_gemm(CoopLds &L, int n_units, int tid)
{
    CoopPipe P;
    coop_plan(L, n_units, tid, P);
    coop_prefetch_w(P);
    coop_prefetch_a(P);
    return coop_run(L, n_units, tid, P);
}

} // namespace atr
