// t2d_device.h — device-side building blocks of the batched Track2D environment (gfx950 / CDNA4 only).
//
// Execution model: ONE WAVEFRONT (64 lanes) PER ENV. An env's 82x82 0/1 map is a 1 KiB bit-packed tile
// (row r = words 3r..3r+2, bit c&31 of word c>>5): 64 lanes x 16 B = the whole tile in one coalesced
// global_load_dwordx4, staged in LDS for the wall tests, the 13x13 crops and (on reset) in-place
// generation. All scalar per-env state is wave-uniform; lane 0 commits it.
//
// Random numbers: Philox4x32-10 counter streams keyed (seed) / (block, episode, global env id, stream) —
// the spec is the PHILOX mode of oracle/track2d_oracle.c, which these functions match bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace t2d {

constexpr int kTileWords = 256;   // 1 KiB per env (246 used)
constexpr int kRowWords = 3;
constexpr int kWavesPerBlock = 4; // 256-thread workgroups: 4 envs per block
constexpr int kObsPerEnv = 338;   // 2 agents x 13 x 13

enum : int { MAP_BLOCK = 0, MAP_MAZE = 1, MAP_EMPTY = 2 };
enum : int { TGT_ADV = 0, TGT_PZR = 1, TGT_FAR = 2, TGT_NAV = 3, TGT_RAM = 4, TGT_RPF = 5 };
enum : uint32_t { STREAM_MAP = 0, STREAM_SPAWN = 1, STREAM_TARGET = 2, STREAM_ACTION = 7 };

// ---- wave-level helpers -----------------------------------------------------------------------------
__device__ __forceinline__ void wave_lds_sync()
{
    // DS operations of one wave execute in order; this only stops the compiler from moving LDS
    // accesses across the point and lets other lanes' writes be re-read.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int uni(int v) { return (int)__builtin_amdgcn_readfirstlane((uint32_t)v); }

// ---- Philox4x32-10 ----------------------------------------------------------------------------------
struct u32x4 { uint32_t x, y, z, w; };

__device__ __forceinline__ u32x4 philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1,
                                               uint32_t c2, uint32_t c3)
{
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return u32x4{c0, c1, c2, c3};
}

// Sequential word reader over one stream (wave-uniform use).
struct Stream {
    uint32_t k0, k1, episode, env, stream;
    uint32_t ctr;      // next word index
    uint32_t blk;      // cached block index (0xffffffff = none)
    u32x4 w;
    __device__ __forceinline__ void init(uint32_t k0_, uint32_t k1_, uint32_t ep, uint32_t env_, uint32_t s,
                                         uint32_t ctr_)
    {
        k0 = k0_; k1 = k1_; episode = ep; env = env_; stream = s; ctr = ctr_; blk = 0xffffffffu;
        w = u32x4{0, 0, 0, 0};
    }
    __device__ __forceinline__ uint32_t next()
    {
        uint32_t i = ctr++;
        uint32_t b = i >> 2;
        if (b != blk) { w = philox4x32_10(k0, k1, b, episode, env, stream); blk = b; }
        uint32_t j = i & 3u;
        return j == 0 ? w.x : (j == 1 ? w.y : (j == 2 ? w.z : w.w));
    }
    // numpy-legacy random_sample layout: 53 bits from two words
    __device__ __forceinline__ double next_double()
    {
        uint32_t a = next() >> 5, b = next() >> 6;
        return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
    }
    // uniform integer in [0, max]: masked rejection, one word per attempt; max == 0 draws nothing
    __device__ __forceinline__ uint32_t bounded(uint32_t max)
    {
        if (max == 0) return 0;
        uint32_t mask = 0xffffffffu >> __builtin_clz(max);
        uint32_t v;
        do { v = next() & mask; } while (v > max);
        return v;
    }
};

// The same word sequence served 64 WORDS at a time: lane l holds word `base + l` (the four lanes of a quad compute
// the same Philox block and keep one component each), so a draw is a single v_readlane with a uniform lane index
// and no branch. Used by the generators, whose long sequential loops (maze growth, spawn picks) would otherwise
// pay a full 10-round Philox on the one active wave every fourth draw.
struct VStream {
    uint32_t k0, k1, episode, env, stream;
    uint32_t ctr;       // next word index (wave-uniform)
    uint32_t base;      // first word held in the lanes (multiple of 64), 0xffffffff = nothing loaded
    uint32_t wv;        // per-lane word `base + lane`
    int lane;
    __device__ __forceinline__ void init(uint32_t k0_, uint32_t k1_, uint32_t ep, uint32_t env_, uint32_t s,
                                         uint32_t ctr_, int lane_)
    {
        k0 = k0_; k1 = k1_; episode = ep; env = env_; stream = s; ctr = ctr_; base = 0xffffffffu; lane = lane_;
        wv = 0u;
    }
    // make the 64-word block that holds word `i` the one in the lanes
    __device__ __forceinline__ void ensure(uint32_t i)
    {
        if ((i & ~63u) != base) {
            base = i & ~63u;
            const u32x4 w = philox4x32_10(k0, k1, (base >> 2) + (uint32_t)(lane >> 2), episode, env, stream);
            const int j = lane & 3;
            wv = j == 0 ? w.x : (j == 1 ? w.y : (j == 2 ? w.z : w.w));
        }
    }
    __device__ __forceinline__ uint32_t next()
    {
        const uint32_t i = uni(ctr);
        ctr = i + 1u;
        ensure(i);
        return __builtin_amdgcn_readlane(wv, (int)(i & 63u));
    }
    __device__ __forceinline__ double next_double()
    {
        uint32_t a = next() >> 5, b = next() >> 6;
        return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
    }
    __device__ __forceinline__ uint32_t bounded(uint32_t max)
    {
        if (max == 0) return 0;
        uint32_t mask = 0xffffffffu >> __builtin_clz(max);
        uint32_t v;
        do { v = next() & mask; } while (v > max);
        return v;
    }
};

// ---- keyed permutation of [0, 6400) (oracle: orc_perm6400) --------------------------------------------
__device__ __forceinline__ uint32_t fmix32(uint32_t h)
{
    h ^= h >> 16; h *= 0x85EBCA6Bu;
    h ^= h >> 13; h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}
__device__ __forceinline__ uint32_t perm6400(const uint32_t (&rk)[8], uint32_t i)
{
    uint32_t a = i / 80u, b = i - a * 80u;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        uint32_t f = __umulhi(fmix32(b * 0x9E3779B1u + rk[r]), 80u);
        uint32_t t = a + f;
        if (t >= 80u) t -= 80u;
        a = b; b = t;
    }
    return a * 80u + b;
}

// ---- map tile access (LDS) ------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t tile_bit(const uint32_t *tile, int r, int c)
{
    return (tile[r * kRowWords + (c >> 5)] >> (c & 31)) & 1u;
}
__device__ __forceinline__ uint32_t valid_mask_w2(int side) { return (1u << (side - 64)) - 1u; }

// Number of free (0) cells per row for rows lane and lane+64, exclusive prefix sums over the wave and
// the total; used to select the k-th free cell in np.where(maze == 0) row-major order
// (G/envs/generators.py:42-43,57-58).
struct FreeIndex {
    int z0, z1;       // zeros in row lane / lane+64
    int ex0, ex1;     // exclusive prefix of z0 over lanes / of z1 over lanes
    int total0, total; // sum z0 / sum z0+z1
};
__device__ __forceinline__ int wave_incl_scan(int v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}
__device__ __forceinline__ FreeIndex build_free_index(const uint32_t *tile, int side, int lane)
{
    FreeIndex f;
    uint32_t m2 = valid_mask_w2(side);
    auto zeros = [&](int row) -> int {
        if (row >= side) return 0;
        const uint32_t *w = tile + row * kRowWords;
        return side - (__popc(w[0]) + __popc(w[1]) + __popc(w[2] & m2));
    };
    f.z0 = zeros(lane);
    f.z1 = zeros(lane + 64);
    int in0 = wave_incl_scan(f.z0, lane), in1 = wave_incl_scan(f.z1, lane);
    f.ex0 = in0 - f.z0; f.ex1 = in1 - f.z1;
    f.total0 = __shfl(in0, 63, 64);
    f.total = f.total0 + __shfl(in1, 63, 64);
    return f;
}
// k-th free cell (row-major). Wave-uniform k in [0, total). Returns r | c << 8.
__device__ __forceinline__ uint32_t select_free(const uint32_t *tile, int side, const FreeIndex &f, int k, int lane)
{
    int row, kk;
    if (k < f.total0) {
        bool mine = (k >= f.ex0) && (k < f.ex0 + f.z0);
        int src = __ffsll((unsigned long long)__ballot(mine)) - 1;
        row = src; kk = k - __shfl(f.ex0, src, 64);
    } else {
        int k2 = k - f.total0;
        bool mine = (k2 >= f.ex1) && (k2 < f.ex1 + f.z1);
        int src = __ffsll((unsigned long long)__ballot(mine)) - 1;
        row = src + 64; kk = k2 - __shfl(f.ex1, src, 64);
    }
    row = uni(row); kk = uni(kk);
    const uint32_t *w = tile + row * kRowWords;
    uint32_t fr[3] = {~w[0], ~w[1], ~w[2] & valid_mask_w2(side)};
    int col = 0;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        int n = __popc(fr[j]);
        if (kk >= 0 && kk < n) {
            uint32_t m = fr[j];
            for (int q = 0; q < kk; q++) m &= m - 1u;
            col = j * 32 + (__ffs(m) - 1);
            kk = -1;
        } else if (kk >= 0) {
            kk -= n;
        }
    }
    return (uint32_t)row | ((uint32_t)col << 8);
}

// ---- generators (write the LDS tile in place) -----------------------------------------------------------
__device__ __forceinline__ void tile_clear(uint32_t *tile, int lane)
{
    reinterpret_cast<uint4 *>(tile)[lane] = make_uint4(0u, 0u, 0u, 0u);
}
__device__ __forceinline__ void tile_border(uint32_t *tile, int side, int lane)
{
    uint32_t last = 1u << (side - 1 - 64);
    for (int row = lane; row < side; row += 64) {
        uint32_t *w = tile + row * kRowWords;
        if (row == 0 || row == side - 1) { w[0] = 0xffffffffu; w[1] = 0xffffffffu; w[2] = (last << 1) - 1u; }
        else { w[0] |= 1u; w[2] |= last; }
    }
}

// RandomBlockMazeGenerator._generate_maze — G/envs/generators.py:157-176: exactly K = int(ratio * 6400)
// distinct interior cells, here the first K images of a keyed permutation; then the wall border.
template <class S>
__device__ __forceinline__ void gen_block(uint32_t *tile, int lane, S &ms, double ratio)
{
    tile_clear(tile, lane);
    int K = (int)(ratio * 6400.0);
    ms.ctr = 4; // round keys = MAP words 4..11
    uint32_t rk[8];
#pragma unroll
    for (int i = 0; i < 8; i++) rk[i] = ms.next();
    wave_lds_sync();
    for (int i = lane; i < K; i += 64) {
        uint32_t c = perm6400(rk, (uint32_t)i);
        uint32_t row = c / 80u + 1u, col = c - (c / 80u) * 80u + 1u;
        atomicOr(&tile[row * kRowWords + (col >> 5)], 1u << (col & 31u));
    }
    wave_lds_sync();
    tile_border(tile, 82, lane);
    wave_lds_sync();
}

// RandomMazeGenerator._generate_maze — G/envs/generators.py:115-145 (81x81). Inherently sequential (every step
// depends on the walls laid so far): one wave-uniform loop whose cost is the number of instructions per growth step
// times the ~5 cycles a lone wave needs per dependent instruction, plus every hand-over between the vector and the
// scalar unit. Restated for a short, almost purely scalar step:
//   * the walk lives on the 41x41 node grid (cells with even coordinates — the only cells the algorithm READS); the
//     position is ONE scalar p = y * 41 + x, which is also the BIT INDEX of the node in a 1681-bit map held by ONE
//     vector register (lane L = bits [32 L, 32 L + 32)): a node test is one v_readlane + a scalar bit test, a carve
//     is one predicated OR;
//   * an INTERIOR node has all four neighbours (generators.py:135-138 drops a neighbour only at x <= 1, x >= S - 2,
//     y <= 1, y >= S - 2, i.e. on the border nodes), the draw over 4 candidates is the low two bits of ONE random word
//     (masked rejection never rejects for max = 3) and the move is p += delta[d] from a byte table in a scalar
//     constant. The walk can only stand on a border node at its seed (border nodes are walls, so it never MOVES onto
//     one): those steps take the general path (presence mask by arithmetic, k-th present direction by a small loop);
//   * the carved mid-points are not written in the loop: every successful move is appended to a log in LDS (one
//     ds_write by lane 0, never waited for) as p | d << 11, and all mid-point walls are set in parallel afterwards;
//   * random words: the 64-block VStream (one readlane per draw).
// Same draws in the same order as before, hence the same mazes bit for bit (oracle PHILOX mode).
__device__ __forceinline__ uint32_t spread16(uint32_t v)   // bit i of the low 16 bits -> bit 2i
{
    v &= 0xffffu;
    v = (v | (v << 8)) & 0x00ff00ffu;
    v = (v | (v << 4)) & 0x0f0f0f0fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}
// one log entry per CARVING move, and a move only carves onto a free interior node, which it then fills: at most 39 * 39 entries
// whatever the level (level 0 / 1 stay below 47 x 24; t2d_create admits level <= 15, i.e. up to 480 seeds x 243 moves)
constexpr int kMazeLogMax = 39 * 39;
constexpr uint32_t kMazeDelta = 0x29d701ffu;   // bytes -1, +1, -41, +41: node (y, x-2), (y, x+2), (y-2, x), (y+2, x)
template <class S>
__device__ __forceinline__ void gen_maze(uint32_t *tile, int lane, S &ms, double ratio, uint32_t *log)
{
#ifdef T2D_EXP_NOMAZE
    const int complexity = 0, density = 0;        // timing probe: everything but the growth loop
#else
    const int complexity = uni((int)(ratio * 810.0));
    const int density = uni((int)(ratio * 1600.0));
#endif
    // node bit map, bit b = y * 41 + x in lane b >> 5: border walls (generators.py:127-128) = node rows / columns 0 and 40
    uint32_t zb = 0u;
#pragma unroll 1
    for (int k = 0; k < 32; k++) {
        const int b = lane * 32 + k, y = b / 41, x = b - y * 41;
        if (b < 1681 && (y == 0 || y == 40 || x == 0 || x == 40)) zb |= 1u << k;
    }
    int nlog = 0;
    // one growth step from node p to node q = p + delta: test the node bit, carve + log + move on a free node. The log
    // entry is the pair (p, q): the carved mid-point is the cell between them.
#define T2D_MAZE_MOVE(DELTA)                                                                                           \
    do {                                                                                                               \
        const int q = p + (DELTA);                                                                                     \
        const uint32_t word = __builtin_amdgcn_readlane(zb, q >> 5);                                                   \
        if (((word >> (q & 31)) & 1u) == 0u) {                        /* Z[y_, x_] == 0: carve */                      \
            zb |= lane == (q >> 5) ? 1u << (q & 31) : 0u;                                                              \
            log[nlog] = (uint32_t)p | ((uint32_t)q << 11);            /* by every lane: same word, no exec juggling   \
               (a per-lane dump address instead was measured: no difference) */                                      \
            nlog++;                                                                                                    \
            p = q;                                                                                                     \
            interior = true;                                          /* a free node is never on the border */         \
        }                                                                                                              \
    } while (0)
    for (int i = 0; i < density; i++) {
        const int sx = (int)ms.bounded(40u);                          // x's draw first (the tuple on generators.py:131)
        const int sy = (int)ms.bounded(40u);
        int p = uni(sy * 41 + sx);
        zb |= lane == (p >> 5) ? 1u << (p & 31) : 0u;                 // Z[y, x] = 1
        bool interior = sx != 0 && sx != 40 && sy != 0 && sy != 40;
        // presence mask of the seed's neighbours, in the order of the list on generators.py:135-138
        const uint32_t m0 = (uint32_t)(sx != 0) | ((uint32_t)(sx != 40) << 1) | ((uint32_t)(sy != 0) << 2) | ((uint32_t)(sy != 40) << 3);
        int j = 0;
        for (; j < complexity && !interior; j++) {                    // border seed: general draw until the walk leaves it
            uint32_t k = ms.bounded((uint32_t)__popc(m0) - 1u);
            uint32_t d = 0u;
            for (uint32_t mm = m0; ; mm >>= 1, d++)
                if (mm & 1u) { if (k == 0u) break; k--; }
            T2D_MAZE_MOVE((int)(int8_t)(kMazeDelta >> (8u * d)));
        }
        while (j < complexity) {                                       // interior: one word per move, block by block
            const uint32_t c = uni(ms.ctr);
            ms.ensure(c);
            // the 64 moves this block of words stands for, decoded by all lanes at once: the walk then needs one
            // v_readlane with a known index per move — off its dependent chain (position -> node word -> test)
            const int dv = (int)(int8_t)(kMazeDelta >> (8u * (ms.wv & 3u)));
            const int c0 = (int)(c & 63u), n = min(64 - c0, complexity - j);
            int k = 0;
            for (; k + 4 <= n; k += 4) {                               // (unrolled by hand: one loop branch per four moves)
                const int d0 = (int)__builtin_amdgcn_readlane((uint32_t)dv, c0 + k), d1 = (int)__builtin_amdgcn_readlane((uint32_t)dv, c0 + k + 1);
                const int d2 = (int)__builtin_amdgcn_readlane((uint32_t)dv, c0 + k + 2), d3 = (int)__builtin_amdgcn_readlane((uint32_t)dv, c0 + k + 3);
                T2D_MAZE_MOVE(d0);
                T2D_MAZE_MOVE(d1);
                T2D_MAZE_MOVE(d2);
                T2D_MAZE_MOVE(d3);
            }
            for (; k < n; k++) T2D_MAZE_MOVE((int)__builtin_amdgcn_readlane((uint32_t)dv, c0 + k));
            ms.ctr = c + (uint32_t)n;
            j += n;
        }
    }
#undef T2D_MAZE_MOVE
    // assemble the tile: even row 2y = node row y at the even columns (rows 0 / 80 are walls at the odd columns too);
    // then the logged mid-points
    tile_clear(tile, lane);
    wave_lds_sync();
    // node row y = bits [41 y, 41 y + 41) of the register bit map: up to three lanes' words (shuffles by every lane:
    // a source lane must be active)
    {
        const int yl = min(lane, 40), b0 = yl * 41, l0 = b0 >> 5, sh = b0 & 31;
        const uint32_t w0 = __shfl(zb, l0, 64), w1 = __shfl(zb, min(l0 + 1, 63), 64), w2 = __shfl(zb, min(l0 + 2, 63), 64);
        uint64_t row = (((uint64_t)w1 << 32) | w0) >> sh;
        if (sh != 0) row |= (uint64_t)w2 << (64 - sh);
        if (lane <= 40) {
            const bool edge = lane == 0 || lane == 40;
            uint32_t *r = tile + (2 * lane) * kRowWords;
            r[0] = edge ? 0xffffffffu : spread16((uint32_t)row);
            r[1] = edge ? 0xffffffffu : spread16((uint32_t)(row >> 16));
            r[2] = edge ? 0x1ffffu : (spread16((uint32_t)(row >> 32)) & 0x1ffffu);
        }
    }
    if (lane < 40) {      // odd rows: the border columns 0 and 80
        uint32_t *r = tile + (2 * lane + 1) * kRowWords;
        r[0] = 1u; r[2] = 0x10000u;
    }
    wave_lds_sync();
    for (int i = lane; i < nlog; i += 64) {
        const uint32_t e = log[i];                                    // nodes p and q (index y * 41 + x) of a carving move
        const uint32_t pp = e & 0x7ffu, qq = e >> 11;
        const uint32_t yp = pp / 41u, xp = pp - yp * 41u, yq = qq / 41u, xq = qq - yq * 41u;
        const uint32_t row = yp + yq, col = xp + xq;                   // the cell between them, in full-resolution cells
        atomicOr(&tile[row * kRowWords + (col >> 5)], 1u << (col & 31u));
    }
    wave_lds_sync();
}

// ---- scripted Ram target (G/envs/navigator.py:73-93) ----------------------------------------------------
// plan word: bits 0..19 ten 2-bit actions, 20..23 length, 24..27 cursor, 28 Nav plan-B flag.
__device__ __forceinline__ uint32_t plan_len(uint32_t p) { return (p >> 20) & 15u; }
__device__ __forceinline__ uint32_t plan_cur(uint32_t p) { return (p >> 24) & 15u; }
__device__ __forceinline__ uint32_t plan_act(uint32_t p, uint32_t i) { return (p >> (2u * i)) & 3u; }
template <class S>
__device__ __forceinline__ uint32_t plan_random(S &ts, uint32_t n)
{
    uint32_t p = 0;
    for (uint32_t i = 0; i < n; i++) p |= ts.bounded(3u) << (2u * i);
    return p | (n << 20);
}
template <class S>
__device__ __forceinline__ uint32_t ram_reset(S &ts)
{
    uint32_t n = 1u + ts.bounded(8u); // randint(1,10) is evaluated before choice(4, n)  (navigator.py:91)
    return plan_random(ts, n);
}
template <class S>
__device__ __forceinline__ uint32_t ram_step(uint32_t &plan, S &ts)
{
    uint32_t cur = plan_cur(plan), len = plan_len(plan);
    uint32_t action = plan_act(plan, cur);
    cur++;
    if (cur >= len) {
        if (ts.bounded(1u) == 0u) {                   // np.random.choice([0,1],1) == 0  (navigator.py:81)
            action = ts.bounded(3u);
            uint32_t n = 1u + ts.bounded(8u);
            plan = (action * 0x55555u & ((1u << (2u * n)) - 1u)) | (n << 20);
        } else {
            uint32_t n = 1u + ts.bounded(8u);
            plan = plan_random(ts, n);
        }
    } else {
        plan = (plan & 0xf0ffffffu) | (cur << 24);
    }
    return action;
}

// ---- Nav target: goal-rooted BFS direction field (device form of Navigator + AstarSolver) -----------------
// The reference plans with heap A* (G/envs/Astar_solver.py:121-149) and follows the plan open-loop
// (G/envs/navigator.py:11-41). A* returns A shortest path; its tie-breaks depend on Python heap/list
// comparison order and are not reproduced on the device. Instead (spec = oracle PHILOX mode, orc_bfs_field):
// dist-to-goal by 4-connected BFS over free cells, dir[cell] = first action in order up, down, left, right whose
// neighbour is one step closer. Following dir from any reachable cell is a shortest path (same length as A*).
//
// Wave-parallel bit BFS held entirely in registers: lane l owns row l (set A) and, for l < 18, row l+64 (set B),
// 3 words per row. One BFS level = OR of the frontier shifted up/down (neighbour lanes) and left/right (96-bit
// shifts), masked by free & ~visited. Direction planes: d0 = (code & 1), d1 = (code >> 1).
struct RowBits { uint32_t w[3]; };

// Neighbour-lane moves on the DPP crossbar (GFX9 wave_shr:1 / wave_shl:1): lane l reads lane l-1 / l+1; the edge
// lane gets 0. One VALU op each instead of a ds_bpermute round trip (the BFS needs 12 of them per level).
__device__ __forceinline__ uint32_t from_prev_lane(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t from_next_lane(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
}

__device__ __forceinline__ RowBits row_shl1(const RowBits &a) // cell c takes the value of cell c-1
{
    RowBits r;
    r.w[0] = a.w[0] << 1;
    r.w[1] = (a.w[1] << 1) | (a.w[0] >> 31);
    r.w[2] = (a.w[2] << 1) | (a.w[1] >> 31);
    return r;
}
__device__ __forceinline__ RowBits row_shr1(const RowBits &a) // cell c takes the value of cell c+1
{
    RowBits r;
    r.w[0] = (a.w[0] >> 1) | (a.w[1] << 31);
    r.w[1] = (a.w[1] >> 1) | (a.w[2] << 31);
    r.w[2] = a.w[2] >> 1;
    return r;
}

struct NavField {
    RowBits visA, visB;    // reachable cells
    RowBits d0A, d0B, d1A, d1B;
};

__device__ __forceinline__ uint32_t rowbits_get(const RowBits &a, const RowBits &b, int r, int c)
{
    // wave-uniform (r, c): fetch the bit from the owning lane
    const int j = c >> 5;
    const bool setb = r >= 64;
    const int owner = setb ? r - 64 : r;
    uint32_t va = j == 0 ? a.w[0] : (j == 1 ? a.w[1] : a.w[2]);
    uint32_t vb = j == 0 ? b.w[0] : (j == 1 ? b.w[1] : b.w[2]);
    uint32_t word = __shfl(setb ? vb : va, owner, 64);
    return (word >> (c & 31)) & 1u;
}

// The four patrol cells of the 'RPF' ids (MazeGenerator.static_goals, generators.py:12-19): (S/6, S/6), (5S/6, S/6),
// (5S/6, 5S/6), (S/6, 5S/6) as r | c << 8.
__device__ __forceinline__ uint32_t rpf_cell(int side, int i)
{
    const uint32_t lo = (uint32_t)(side / 6), hi = (uint32_t)(side * 5 / 6);
    const uint32_t r = (i == 1 || i == 2) ? hi : lo, c = (i >= 2) ? hi : lo;
    return r | (c << 8);
}

// clear_rpf: plan on the GENERATOR's map of an RPF env, i.e. the env's tile with the four patrol cells free
// (track_1v1.py:233-236). (qr, qc): the cell the plan will be followed FROM (-1: none); returns its BFS distance (-1 if
// unreachable or not asked for; by value — a nullable out-pointer kept the caller's variable in scratch memory). The flood stops at the level that reaches (qr, qc): every cell of a shortest path from there to the
// goal is closer to the goal, hence already labelled — the rest of the field would never be read (about half the levels
// on average).
__device__ __forceinline__ int bfs_dir_field(const uint32_t *tile, int side, int lane, int gr, int gc, NavField &f,
                                              bool clear_rpf = false, int qr = -1, int qc = -1)
{
    RowBits freeA, freeB, frA, frB;
    const uint32_t m2 = valid_mask_w2(side);
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const uint32_t vm = j == 2 ? m2 : 0xffffffffu;
        freeA.w[j] = (lane < side) ? (~tile[lane * kRowWords + j] & vm) : 0u;
        freeB.w[j] = (lane + 64 < side) ? (~tile[(lane + 64) * kRowWords + j] & vm) : 0u;
        frA.w[j] = 0u; frB.w[j] = 0u;
        f.d0A.w[j] = f.d0B.w[j] = f.d1A.w[j] = f.d1B.w[j] = 0u;
    }
    if (clear_rpf) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t cell = rpf_cell(side, i);
            const int r = (int)(cell & 0xffu), c = (int)(cell >> 8);
#pragma unroll
            for (int j = 0; j < 3; j++) {
                if (j == (c >> 5) && r < 64 && lane == r) freeA.w[j] |= 1u << (c & 31);
                if (j == (c >> 5) && r >= 64 && lane == r - 64) freeB.w[j] |= 1u << (c & 31);
            }
        }
    }
    int level = 0, found = (qr == gr && qc == gc) ? 0 : -1;
    {   // seed the frontier with the goal cell
        const uint32_t bit = 1u << (gc & 31);
        const int j = gc >> 5;
#pragma unroll
        for (int q = 0; q < 3; q++) {
            if (q == j && gr < 64 && lane == gr) frA.w[q] = bit;
            if (q == j && gr >= 64 && lane == gr - 64) frB.w[q] = bit;
        }
    }
    f.visA = frA; f.visB = frB;
    for (;;) {
        // Rows 64 .. side-1 (set B, 17-18 of the 81-82 rows) take no part in a level unless the frontier stands in them or in
        // row 63: a wave-uniform test per level skips their share of the work (about 40 % of a level's instructions) for the
        // larger part of most floods — a lone wave's instruction count is its latency.
        const bool actB = __ballot((frB.w[0] | frB.w[1] | frB.w[2]) != 0u || (lane == 63 && (frA.w[0] | frA.w[1] | frA.w[2]) != 0u)) != 0ull;
        RowBits upA, dnA, upB, dnB;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const uint32_t a_prev = from_prev_lane(frA.w[j]);       // row lane-1 (set A)
            const uint32_t a_next = from_next_lane(frA.w[j]);       // row lane+1 (set A)
            upA.w[j] = lane == 0 ? 0u : a_prev;                     // frontier cell above  -> action 0 (up)
            dnA.w[j] = a_next;                                      // frontier cell below  -> action 1 (down)
            upB.w[j] = 0u; dnB.w[j] = 0u;
        }
        if (actB) {
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const uint32_t b_prev = from_prev_lane(frB.w[j]);
                const uint32_t b_next = from_next_lane(frB.w[j]);
                const uint32_t a_last = __builtin_amdgcn_readlane(frA.w[j], 63);  // row 63
                const uint32_t b_first = __builtin_amdgcn_readlane(frB.w[j], 0);  // row 64
                if (lane == 63) dnA.w[j] = b_first;
                upB.w[j] = lane == 0 ? a_last : b_prev;
                dnB.w[j] = lane == 63 ? 0u : b_next;
            }
        }
        const RowBits lfA = row_shl1(frA), rtA = row_shr1(frA);     // frontier cell to the left / right
        uint32_t any = 0u;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const uint32_t open = freeA.w[j] & ~f.visA.w[j];
            const uint32_t u = upA.w[j] & open, d = dnA.w[j] & open & ~u;
            const uint32_t l = lfA.w[j] & open & ~(u | d), r = rtA.w[j] & open & ~(u | d | l);
            const uint32_t nw = u | d | l | r;
            f.d0A.w[j] |= d | r; f.d1A.w[j] |= l | r;
            f.visA.w[j] |= nw; frA.w[j] = nw; any |= nw;
        }
        if (actB) {
            const RowBits lfB = row_shl1(frB), rtB = row_shr1(frB);
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const uint32_t open = freeB.w[j] & ~f.visB.w[j];
                const uint32_t u = upB.w[j] & open, d = dnB.w[j] & open & ~u;
                const uint32_t l = lfB.w[j] & open & ~(u | d), r = rtB.w[j] & open & ~(u | d | l);
                const uint32_t nw = u | d | l | r;
                f.d0B.w[j] |= d | r; f.d1B.w[j] |= l | r;
                f.visB.w[j] |= nw; frB.w[j] = nw; any |= nw;
            }
        }
        level++;
        if (qr >= 0 && found < 0) {           // did this level reach the cell the plan starts from?
            uint32_t hit = 0u;
#pragma unroll
            for (int j = 0; j < 3; j++) {
                if (j == (qc >> 5) && qr < 64 && lane == qr) hit = frA.w[j] & (1u << (qc & 31));
                if (j == (qc >> 5) && qr >= 64 && lane == qr - 64) hit = frB.w[j] & (1u << (qc & 31));
            }
            if (__ballot(hit != 0u) != 0ull) found = level;
        }
        if (found >= 0) break;                // everything a path from (qr, qc) can touch is labelled
        if (__ballot(any != 0u) == 0ull) break;
    }
    return found;
}

// direction-plane tile in HBM: plane 0 = words 0..245, plane 1 = words 256..501 (same row layout as the map)
constexpr int kDirWords = 512;
__device__ __forceinline__ void store_dir_field(uint32_t *gdir, const NavField &f, int side, int lane)
{
#pragma unroll
    for (int j = 0; j < 3; j++) {
        if (lane < side) { gdir[lane * kRowWords + j] = f.d0A.w[j]; gdir[256 + lane * kRowWords + j] = f.d1A.w[j]; }
        if (lane + 64 < side) {
            gdir[(lane + 64) * kRowWords + j] = f.d0B.w[j];
            gdir[256 + (lane + 64) * kRowWords + j] = f.d1B.w[j];
        }
    }
}
// Prefetched plan of a Nav target (goal drawn ahead of time, field computed by the generator pass): the two direction
// planes as above plus the visited plane at words 512..757, so that reachability of the target's position can be
// checked when the plan is adopted.
constexpr int kPlanWords = 768;
__device__ __forceinline__ void store_plan_field(uint32_t *gp, const NavField &f, int side, int lane)
{
    store_dir_field(gp, f, side, lane);
#pragma unroll
    for (int j = 0; j < 3; j++) {
        if (lane < side) gp[512 + lane * kRowWords + j] = f.visA.w[j];
        if (lane + 64 < side) gp[512 + (lane + 64) * kRowWords + j] = f.visB.w[j];
    }
}
__device__ __forceinline__ uint32_t load_vis(const uint32_t *gp, int r, int c)
{
    return (gp[512 + r * kRowWords + (c >> 5)] >> (c & 31)) & 1u;
}

__device__ __forceinline__ uint32_t load_dir(const uint32_t *gdir, int r, int c)
{
    const int w = r * kRowWords + (c >> 5);
    return ((gdir[w] >> (c & 31)) & 1u) | (((gdir[256 + w] >> (c & 31)) & 1u) << 1);
}

// ---- rewards (G/envs/track_1v1.py:94-104), float64 in the reference's operation order ---------------------
__device__ __forceinline__ void reward_f64(uint32_t d2, double w_p, double &r_track, double &r_target)
{
    const double max_distance = 6.0;
    double distance = __dsqrt_rn((double)d2);
    double rt = 1.0 - __ddiv_rn(2.0 * distance, max_distance);
    rt = rt > -1.0 ? rt : -1.0;
    double over = distance - max_distance;
    over = over > 0.0 ? over : 0.0;
    double rg = -rt - __ddiv_rn(w_p * over, max_distance);
    rg = rg > -1.0 ? rg : -1.0;
    r_track = rt; r_target = rg;
}

} // namespace t2d
