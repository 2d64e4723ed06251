// track2d_hip.hip — kernels + C ABI (include/track2d.h) of the batched Track2D environment for MI355X.
//
// One fused kernel family, k_env<OP>, covers reset / step (+ in-launch auto-reset) / observe:
//   phase A, one wavefront per env: stage the env's 1 KiB bit-packed map tile in LDS with a single
//            coalesced 16 B/lane load, read the wave-uniform SoA state, apply the (scripted) actions with
//            wall tests on the LDS tile, integer d^2 -> float64 reward -> far counter / time limit -> done,
//            regenerate finished envs in place in LDS (Philox streams) and write the tile back;
//   phase B, whole 256-thread workgroup: expand the 4 envs' 2x13x13 crops from the LDS tiles into
//            16 B/lane coalesced f32 stores (the HBM-dominant part: 1352 B of 1723 B per env-step).
// Reference semantics: envs/gym-track2d/gym_track2d/envs/track_1v1.py:71-168,271-326 (cited per function in
// t2d_device.h and below). Bit-exact spec: the PHILOX mode of oracle/track2d_oracle.c.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/track2d.h"
#include "t2d_device.h"

namespace t2d {

struct DevState {
    uint32_t *maps;     // [N][256] bit-packed tiles
    uint32_t *pos;      // [N] tracker r | c<<8 | target r<<16 | c<<24
    uint32_t *goals;    // [N] goal0 r | c<<8 | goal1 r<<16 | c<<24
    uint32_t *cnt;      // [N] c_far | t<<8 | side<<24
    uint32_t *cfg;      // [N] map_type | target_mode<<2 | level<<5
    uint32_t *episode;  // [N]
    uint32_t *plan;     // [N] scripted-target plan word
    uint32_t *tctr;     // [N] TARGET stream word counter
    uint32_t *navgoal;  // [N] r | c<<8
    uint32_t *d2;       // [N] last squared distance
    uint32_t *dirf;     // [N][512] Nav direction planes (allocated only if some env has a Nav target)
    uint32_t *faults;   // [1]
    int n;
    uint32_t env_base, k0, k1;
    int max_steps, auto_reset;
};

enum : int { OP_STEP = 0, OP_RESET = 1, OP_OBSERVE = 2 };

__device__ __forceinline__ int load_action(const void *p, int dtype, int e, uint32_t *faults)
{
    long long v;
    if (dtype == T2D_ACT_U8) v = reinterpret_cast<const uint8_t *>(p)[e];
    else if (dtype == T2D_ACT_I32) v = reinterpret_cast<const int32_t *>(p)[e];
    else v = reinterpret_cast<const long long *>(p)[e];
    if (v < 0 || v > 3) { atomicOr(faults, 1u); v &= 3; }
    return (int)v;
}

// Navigator.reset / the re-plan branch of Navigator.step (navigator.py:43-63, :15-38): plan from (fr, fc) to navgoal;
// unreachable or empty plan -> resample the goal, the 6th failure -> plan B (10 random actions).
__device__ __forceinline__ void nav_plan(const uint32_t *tile, int side, int lane, int fr, int fc, const FreeIndex &fi,
                                         uint32_t &navgoal, Stream &ts, uint32_t &plan, NavField &f)
{
    int count_res = 0;
    bool planb = false;
    for (;;) {
        const int gr = (int)(navgoal & 0xffu), gc = (int)(navgoal >> 8);
        bfs_dir_field(tile, side, lane, gr, gc, f);
        const bool ok = rowbits_get(f.visA, f.visB, fr, fc) != 0u && !(fr == gr && fc == gc);
        if (ok) break;
        if (++count_res > 5) { planb = true; break; }
        navgoal = select_free(tile, side, fi, (int)ts.bounded((uint32_t)(fi.total - 1)), lane);
    }
    plan = planb ? (plan_random(ts, 10u) | (1u << 28)) : 0u;
}
__device__ __forceinline__ uint32_t nav_dir_from_regs(const NavField &f, int r, int c)
{
    return rowbits_get(f.d0A, f.d0B, r, c) | (rowbits_get(f.d1A, f.d1B, r, c) << 1);
}

// Track1v1Env.reset -> init_maze (track_1v1.py:134-168,218-240) for one env, executed by one wave on its
// LDS tile. All arguments are wave-uniform.
__device__ __forceinline__ void reset_env(const DevState &s, int e, uint32_t *tile, int lane, uint32_t cfg,
                                          uint32_t &pos, uint32_t &goals, uint32_t &cnt, uint32_t &episode,
                                          uint32_t &plan, uint32_t &tctr, uint32_t &navgoal, uint32_t &d2)
{
    const int map_type = cfg & 3, mode = (cfg >> 2) & 7, level = (cfg >> 5) & 15;
    const uint32_t genv = s.env_base + (uint32_t)e;
    episode += 1u;
    Stream ms;
    ms.init(s.k0, s.k1, episode, genv, STREAM_MAP, 0);
    int side = 82;
    if (map_type == MAP_MAZE) {
        double r = level > 0 ? (double)level * 0.02 : .03 * ms.next_double();
        gen_maze(tile, lane, ms, r);
        side = 81;
    } else if (map_type == MAP_BLOCK) {
        double r = level > 0 ? (double)level * 0.05 : 0.15 * ms.next_double();
        gen_block(tile, lane, ms, r);
    } else {
        gen_block(tile, lane, ms, 0.0);
    }
    const FreeIndex fi = build_free_index(tile, side, lane);
    const int n = fi.total;
    Stream ss;
    ss.init(s.k0, s.k1, episode, genv, STREAM_SPAWN, 0);
    uint32_t g0, g1;
    auto sample_goal2 = [&]() { // MazeGenerator.sample_goal(2), generators.py:38-51
        int i0 = (int)ss.bounded((uint32_t)(n - 1));
        int i1 = (int)ss.bounded((uint32_t)(n - 2));
        if (i1 >= i0) i1++;
        g0 = select_free(tile, side, fi, i0, lane);
        g1 = select_free(tile, side, fi, i1, lane);
    };
    sample_goal2();
    // sample_close_states(2, 1), generators.py:53-77 + get_around :82-94 (2x2 block up-left of the tracker)
    const uint32_t tr = select_free(tile, side, fi, (int)ss.bounded((uint32_t)(n - 1)), lane);
    const int r = (int)(tr & 0xffu), c = (int)(tr >> 8);
    const int x0 = max(0, r - 1), x1 = min(side - 1, r + 1), y0 = max(0, c - 1), y1 = min(side - 1, c + 1);
    int m = 0;
    for (int rr = x0; rr < x1; rr++)
        for (int cc = y0; cc < y1; cc++) m += (int)(tile_bit(tile, rr, cc) == 0u);
    int j = (int)ss.bounded((uint32_t)(m - 1));
    uint32_t tg = tr;
    for (int rr = x0; rr < x1; rr++)
        for (int cc = y0; cc < y1; cc++)
            if (tile_bit(tile, rr, cc) == 0u) {
                if (j == 0) tg = (uint32_t)rr | ((uint32_t)cc << 8);
                j--;
            }
    while (tr == g0 || tr == g1) sample_goal2(); // goal_test loop, track_1v1.py:239-240
    pos = tr | (tg << 16);
    goals = g0 | (g1 << 16);
    Stream ts;
    ts.init(s.k0, s.k1, episode, genv, STREAM_TARGET, 0);
    plan = 0;
    navgoal = g1;
    if (mode == TGT_RAM) plan = ram_reset(ts);
    if (mode == TGT_NAV) { // Navigator.reset (navigator.py:43-63): plan from the target spawn to goal_states[1]
        NavField nf;
        nav_plan(tile, side, lane, (int)(tg & 0xffu), (int)(tg >> 8), fi, navgoal, ts, plan, nf);
        if (((plan >> 28) & 1u) == 0u) store_dir_field(s.dirf + (size_t)e * kDirWords, nf, side, lane);
    }
    tctr = ts.ctr;
    cnt = (uint32_t)side << 24;
    const int dr = (int)(tg & 0xffu) - r, dc = (int)(tg >> 8) - c;
    d2 = (uint32_t)(dr * dr + dc * dc);
}

template <int OP, bool RANDOM>
__global__ __launch_bounds__(256) void k_env(DevState s, const void *act0, const void *act1, int act_dtype,
                                             const uint8_t *mask, float *obs, float *rew, uint8_t *done_out,
                                             uint32_t aseed_lo, uint32_t aseed_hi, uint32_t step_idx)
{
    __shared__ __attribute__((aligned(16))) uint32_t tiles[kWavesPerBlock][kTileWords];
    __shared__ uint32_t s_pos[kWavesPerBlock];
    __shared__ int s_side[kWavesPerBlock];

    const int lane = (int)(threadIdx.x & 63u);
    const int wave = uni((int)(threadIdx.x >> 6));
    const int e = (int)blockIdx.x * kWavesPerBlock + wave;

    if (e < s.n) {
        uint32_t *tile = tiles[wave];
        uint32_t *gtile = s.maps + (size_t)e * kTileWords;
        reinterpret_cast<uint4 *>(tile)[lane] = reinterpret_cast<const uint4 *>(gtile)[lane];
        uint32_t pos = s.pos[e], cnt = s.cnt[e];
        const uint32_t cfg = s.cfg[e];
        uint32_t goals = 0, episode = 0, plan = 0, tctr = 0, navgoal = 0, d2 = 0;
        const int mode = (int)((cfg >> 2) & 7u);
        bool do_reset = false, dirty = false, s_navgoal_dirty = false;
        wave_lds_sync();

        if (OP == OP_RESET) do_reset = (mask == nullptr) || (mask[e] != 0);

        if (OP == OP_STEP) {
            const uint32_t genv = s.env_base + (uint32_t)e;
            int side = (int)(cnt >> 24), c_far = (int)(cnt & 0xffu), t = (int)((cnt >> 8) & 0xffffu);
            int a_tr, a_tg;
            if (RANDOM) {
                u32x4 w = philox4x32_10(aseed_lo, aseed_hi, step_idx, 0u, genv, STREAM_ACTION);
                a_tr = (int)(w.x & 3u); a_tg = (int)(w.y & 3u);
            } else {
                a_tr = load_action(act0, act_dtype, e, s.faults);
                a_tg = act1 ? load_action(act1, act_dtype, e, s.faults) : 0;
            }
            if (mode == TGT_RAM) { // track_1v1.py:81-82
                plan = s.plan[e]; tctr = s.tctr[e]; episode = s.episode[e];
                Stream ts;
                ts.init(s.k0, s.k1, episode, genv, STREAM_TARGET, tctr);
                a_tg = (int)ram_step(plan, ts);
                tctr = ts.ctr;
                dirty = true;
            }
            int r0 = (int)(pos & 0xffu), c0 = (int)((pos >> 8) & 0xffu);
            int r1 = (int)((pos >> 16) & 0xffu), c1 = (int)(pos >> 24);
            if (mode == TGT_NAV) { // track_1v1.py:83-84 -> Navigator.step(old_state[1], ...) (navigator.py:11-41)
                plan = s.plan[e]; tctr = s.tctr[e]; episode = s.episode[e]; navgoal = s.navgoal[e];
                uint32_t *gdir = s.dirf + (size_t)e * kDirWords;
                Stream ts;
                ts.init(s.k0, s.k1, episode, genv, STREAM_TARGET, tctr);
                bool planb = ((plan >> 28) & 1u) != 0u;
                const bool exhausted = planb ? (plan_cur(plan) >= plan_len(plan))
                                             : (r1 == (int)(navgoal & 0xffu) && c1 == (int)(navgoal >> 8));
                uint32_t dir = 0;
                if (exhausted) {
                    const FreeIndex fi = build_free_index(tile, side, lane);
                    navgoal = select_free(tile, side, fi, (int)ts.bounded((uint32_t)(fi.total - 1)), lane);
                    NavField nf;
                    nav_plan(tile, side, lane, r1, c1, fi, navgoal, ts, plan, nf);
                    planb = ((plan >> 28) & 1u) != 0u;
                    if (!planb) { store_dir_field(gdir, nf, side, lane); dir = nav_dir_from_regs(nf, r1, c1); }
                    s_navgoal_dirty = true;
                } else if (!planb) {
                    dir = load_dir(gdir, r1, c1);
                }
                if (planb) {
                    const uint32_t cur = plan_cur(plan);
                    a_tg = (int)plan_act(plan, cur);
                    plan = (plan & 0xf0ffffffu) | ((cur + 1u) << 24);
                } else {
                    a_tg = (int)dir;
                }
                tctr = ts.ctr;
                dirty = true;
            }
            // _next_state (track_1v1.py:271-285): stay put iff the destination cell is a wall
            {
                int nr = r0 + (a_tr == 0 ? -1 : (a_tr == 1 ? 1 : 0)), nc = c0 + (a_tr == 2 ? -1 : (a_tr == 3 ? 1 : 0));
                if (tile_bit(tile, nr, nc) == 0u) { r0 = nr; c0 = nc; }
                nr = r1 + (a_tg == 0 ? -1 : (a_tg == 1 ? 1 : 0)); nc = c1 + (a_tg == 2 ? -1 : (a_tg == 3 ? 1 : 0));
                if (tile_bit(tile, nr, nc) == 0u) { r1 = nr; c1 = nc; }
            }
            pos = (uint32_t)r0 | ((uint32_t)c0 << 8) | ((uint32_t)r1 << 16) | ((uint32_t)c1 << 24);
            const int dr = r1 - r0, dc = c1 - c0;
            d2 = (uint32_t)(dr * dr + dc * dc);
            const double w_p = mode == TGT_PZR ? 1.0 : (mode == TGT_FAR ? -0.5 : 0.0); // track_1v1.py:147-152
            double rt, rg;
            reward_f64(d2, w_p, rt, rg);
            c_far = d2 <= 36u ? 0 : min(c_far + 1, 255);  // distance <= 6 (track_1v1.py:106-109)
            int dn = c_far > 10;
            t = min(t + 1, 65535);
            if (s.max_steps > 0 && t >= s.max_steps) dn = 1; // gym TimeLimit
            cnt = (uint32_t)c_far | ((uint32_t)t << 8) | ((uint32_t)side << 24);
            if (lane == 0) {
                reinterpret_cast<float2 *>(rew)[e] = make_float2((float)rt, (float)rg);
                done_out[e] = (uint8_t)dn;
            }
            do_reset = dn && s.auto_reset;
        }

        if (do_reset) {
            episode = s.episode[e];
            reset_env(s, e, tile, lane, cfg, pos, goals, cnt, episode, plan, tctr, navgoal, d2);
            wave_lds_sync();
            reinterpret_cast<uint4 *>(gtile)[lane] = reinterpret_cast<const uint4 *>(tile)[lane];
        }
        if (lane == 0 && OP != OP_OBSERVE) {
            if (OP == OP_STEP || do_reset) { s.pos[e] = pos; s.cnt[e] = cnt; s.d2[e] = d2; }
            if (do_reset) { s.goals[e] = goals; s.episode[e] = episode; s.navgoal[e] = navgoal; }
            if (do_reset || dirty) { s.plan[e] = plan; s.tctr[e] = tctr; }
            if (s_navgoal_dirty && !do_reset) s.navgoal[e] = navgoal;
        }
        if (lane == 0) { s_pos[wave] = pos; s_side[wave] = (int)(cnt >> 24); }
    }
    __syncthreads();

    // ---- phase B: _get_obs (track_1v1.py:287-326) for the block's envs, 16 B per lane per store ----------
    if (obs == nullptr) return;
    const int first = (int)blockIdx.x * kWavesPerBlock;
    const int nloc = min(kWavesPerBlock, s.n - first);
    const int nflt = nloc * kObsPerEnv;
    float *out = obs + (size_t)first * kObsPerEnv;
    for (int q = (int)threadIdx.x; q * 4 < nflt; q += (int)blockDim.x) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int f = q * 4 + j;
            const int el = f / kObsPerEnv;
            const int rem = f - el * kObsPerEnv;
            const int ag = rem >= 169 ? 1 : 0;
            const int ci = rem - ag * 169;
            const int y = ci / 13, x = ci - y * 13;
            const int eli = min(el, nloc - 1);
            const uint32_t p = s_pos[eli];
            const int side = s_side[eli];
            const int tr_r = (int)(p & 0xffu), tr_c = (int)((p >> 8) & 0xffu);
            const int tg_r = (int)((p >> 16) & 0xffu), tg_c = (int)(p >> 24);
            const int rr = (ag ? tg_r : tr_r) - T2D_POB + y, cc = (ag ? tg_c : tr_c) - T2D_POB + x;
            float val = 1.0f; // np.pad(..., constant_values=1) outside the map (track_1v1.py:321)
            if ((unsigned)rr < (unsigned)side && (unsigned)cc < (unsigned)side) {
                val = (float)tile_bit(tiles[eli], rr, cc);
                if (rr == tr_r && cc == tr_c) val = 2.0f;          // tracker, track_1v1.py:300-305
                if (rr == tg_r && cc == tg_c) val = 4.0f;          // target painted last
                if (ci == 84) val = ag ? 4.0f : 2.0f;              // own cell re-painted, :313
            }
            v[j] = val;
        }
        if (q * 4 + 3 < nflt) reinterpret_cast<float4 *>(out)[q] = make_float4(v[0], v[1], v[2], v[3]);
        else
            for (int j = 0; j < 4; j++)
                if (q * 4 + j < nflt) out[q * 4 + j] = v[j];
    }
}

__global__ void k_reward_table(const uint32_t *d2, int n, double w_p, float *r_track, float *r_target)
{
    int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    double a, b;
    reward_f64(d2[i], w_p, a, b);
    r_track[i] = (float)a; r_target[i] = (float)b;
}

} // namespace t2d

// =====================================================================================================
// Host side: C ABI
// =====================================================================================================
using namespace t2d;

struct t2d_handle {
    DevState s;
    int device;
    bool reset_done;
    uint32_t random_step;
};

static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess) return fail(T2D_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e));  \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    bool changed = false;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) changed = (hipSetDevice(dev) == hipSuccess);
    }
    ~DeviceGuard()
    {
        if (changed) (void)hipSetDevice(prev);
    }
};

extern "C" const char *t2d_last_error(void) { return g_err; }
extern "C" int t2d_abi_version(void) { return T2D_ABI_VERSION; }

extern "C" int t2d_num_envs(const t2d_handle *h) { return h ? h->s.n : T2D_ERR_INVALID; }

extern "C" int t2d_create(const t2d_config *cfg, t2d_handle **out)
{
    if (!cfg || !out) return fail(T2D_ERR_INVALID, "t2d_create: null argument");
    if (cfg->abi_version != T2D_ABI_VERSION)
        return fail(T2D_ERR_INVALID, "t2d_create: abi_version %u, library is %d", cfg->abi_version, T2D_ABI_VERSION);
    if (cfg->num_envs <= 0) return fail(T2D_ERR_INVALID, "t2d_create: num_envs must be > 0");
    if (cfg->max_episode_steps < 0 || cfg->max_episode_steps > 65535)
        return fail(T2D_ERR_INVALID, "t2d_create: max_episode_steps must be in [0, 65535]");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(T2D_ERR_NO_DEVICE, "t2d_create: no HIP device visible");
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(T2D_ERR_INVALID, "t2d_create: device %d out of range (%d visible)", cfg->device, ndev);
    const int n = cfg->num_envs;
    std::vector<uint32_t> hcfg((size_t)n);
    bool has_nav = false;
    for (int i = 0; i < n; i++) {
        uint32_t mt = cfg->map_type_per_env ? cfg->map_type_per_env[i] : cfg->map_type;
        uint32_t tm = cfg->target_mode_per_env ? cfg->target_mode_per_env[i] : cfg->target_mode;
        uint32_t lv = cfg->level_per_env ? cfg->level_per_env[i] : cfg->level;
        if (mt > T2D_MAP_EMPTY) return fail(T2D_ERR_INVALID, "t2d_create: map_type %u (env %d)", mt, i);
        if (tm > T2D_TGT_RAM) return fail(T2D_ERR_INVALID, "t2d_create: target_mode %u (env %d)", tm, i);
        has_nav = has_nav || tm == T2D_TGT_NAV;
        if (lv > 15) return fail(T2D_ERR_INVALID, "t2d_create: level %u (env %d)", lv, i);
        hcfg[(size_t)i] = mt | (tm << 2) | (lv << 5);
    }
    DeviceGuard guard(cfg->device);
    t2d_handle *h = new (std::nothrow) t2d_handle();
    if (!h) return fail(T2D_ERR_INVALID, "t2d_create: out of host memory");
    std::memset(&h->s, 0, sizeof(h->s));
    h->device = cfg->device;
    h->reset_done = false;
    h->random_step = 0;
    DevState &s = h->s;
    s.n = n; s.env_base = cfg->env_id_base;
    s.k0 = (uint32_t)cfg->seed; s.k1 = (uint32_t)(cfg->seed >> 32);
    s.max_steps = cfg->max_episode_steps; s.auto_reset = cfg->auto_reset ? 1 : 0;
    const size_t nb = (size_t)n * sizeof(uint32_t);
    uint32_t **arrs[] = {&s.pos, &s.goals, &s.cnt, &s.cfg, &s.episode, &s.plan, &s.tctr, &s.navgoal, &s.d2};
    hipError_t err = hipMalloc((void **)&s.maps, (size_t)n * kTileWords * sizeof(uint32_t));
    if (err == hipSuccess) err = hipMemset(s.maps, 0, (size_t)n * kTileWords * sizeof(uint32_t));
    for (auto a : arrs) {
        if (err == hipSuccess) err = hipMalloc((void **)a, nb);
        if (err == hipSuccess) err = hipMemset(*a, 0, nb);
    }
    if (has_nav) {
        if (err == hipSuccess) err = hipMalloc((void **)&s.dirf, (size_t)n * kDirWords * sizeof(uint32_t));
        if (err == hipSuccess) err = hipMemset(s.dirf, 0, (size_t)n * kDirWords * sizeof(uint32_t));
    }
    if (err == hipSuccess) err = hipMalloc((void **)&s.faults, sizeof(uint32_t));
    if (err == hipSuccess) err = hipMemset(s.faults, 0, sizeof(uint32_t));
    if (err == hipSuccess) err = hipMemcpy(s.cfg, hcfg.data(), nb, hipMemcpyHostToDevice);
    if (err != hipSuccess) {
        t2d_destroy(h);
        return fail(T2D_ERR_HIP, "t2d_create: device allocation failed: %s", hipGetErrorString(err));
    }
    *out = h;
    return T2D_OK;
}

extern "C" int t2d_destroy(t2d_handle *h)
{
    if (!h) return T2D_OK;
    DeviceGuard guard(h->device);
    DevState &s = h->s;
    void *ptrs[] = {s.maps, s.pos, s.goals, s.cnt, s.cfg, s.episode, s.plan, s.tctr, s.navgoal, s.d2, s.dirf, s.faults};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    delete h;
    return T2D_OK;
}

static inline dim3 env_grid(int n) { return dim3((unsigned)((n + kWavesPerBlock - 1) / kWavesPerBlock)); }

extern "C" int t2d_reset(t2d_handle *h, const uint8_t *mask_dev, float *obs_dev, void *stream)
{
    if (!h) return fail(T2D_ERR_INVALID, "t2d_reset: null handle");
    DeviceGuard guard(h->device);
    hipLaunchKernelGGL((k_env<OP_RESET, false>), env_grid(h->s.n), dim3(256), 0, (hipStream_t)stream, h->s,
                       nullptr, nullptr, 0, mask_dev, obs_dev, nullptr, nullptr, 0u, 0u, 0u);
    HIP_TRY(hipGetLastError());
    if (mask_dev == nullptr) h->reset_done = true;
    return T2D_OK;
}

extern "C" int t2d_step(t2d_handle *h, const void *act_tracker_dev, const void *act_target_dev, int act_dtype,
                        float *obs_dev, float *rew_dev, uint8_t *done_dev, void *stream)
{
    if (!h) return fail(T2D_ERR_INVALID, "t2d_step: null handle");
    if (!act_tracker_dev || !rew_dev || !done_dev) return fail(T2D_ERR_INVALID, "t2d_step: null buffer");
    if (act_dtype < T2D_ACT_U8 || act_dtype > T2D_ACT_I64) return fail(T2D_ERR_INVALID, "t2d_step: act_dtype %d", act_dtype);
    if (!h->reset_done) return fail(T2D_ERR_STATE, "t2d_step: call t2d_reset (all envs) or t2d_inject first");
    DeviceGuard guard(h->device);
    hipLaunchKernelGGL((k_env<OP_STEP, false>), env_grid(h->s.n), dim3(256), 0, (hipStream_t)stream, h->s,
                       act_tracker_dev, act_target_dev, act_dtype, nullptr, obs_dev, rew_dev, done_dev, 0u, 0u, 0u);
    HIP_TRY(hipGetLastError());
    return T2D_OK;
}

extern "C" int t2d_step_random(t2d_handle *h, int steps, uint64_t action_seed, float *obs_dev, float *rew_dev,
                               uint8_t *done_dev, void *stream)
{
    if (!h) return fail(T2D_ERR_INVALID, "t2d_step_random: null handle");
    if (!rew_dev || !done_dev || steps < 0) return fail(T2D_ERR_INVALID, "t2d_step_random: bad argument");
    if (!h->reset_done) return fail(T2D_ERR_STATE, "t2d_step_random: reset first");
    DeviceGuard guard(h->device);
    for (int i = 0; i < steps; i++) {
        hipLaunchKernelGGL((k_env<OP_STEP, true>), env_grid(h->s.n), dim3(256), 0, (hipStream_t)stream, h->s,
                           nullptr, nullptr, 0, nullptr, obs_dev, rew_dev, done_dev, (uint32_t)action_seed,
                           (uint32_t)(action_seed >> 32), h->random_step++);
    }
    HIP_TRY(hipGetLastError());
    return T2D_OK;
}

extern "C" int t2d_observe(t2d_handle *h, float *obs_dev, void *stream)
{
    if (!h || !obs_dev) return fail(T2D_ERR_INVALID, "t2d_observe: null argument");
    DeviceGuard guard(h->device);
    hipLaunchKernelGGL((k_env<OP_OBSERVE, false>), env_grid(h->s.n), dim3(256), 0, (hipStream_t)stream, h->s,
                       nullptr, nullptr, 0, nullptr, obs_dev, nullptr, nullptr, 0u, 0u, 0u);
    HIP_TRY(hipGetLastError());
    return T2D_OK;
}

static int check_range(const t2d_handle *h, int first, int count, const char *who)
{
    if (!h) return fail(T2D_ERR_INVALID, "%s: null handle", who);
    if (first < 0 || count < 0 || first + count > h->s.n)
        return fail(T2D_ERR_INVALID, "%s: env range [%d, %d) outside [0, %d)", who, first, first + count, h->s.n);
    return T2D_OK;
}

extern "C" int t2d_inject(t2d_handle *h, int first, int count, int side, const uint8_t *maze_host,
                          const int32_t *pos_host, const int32_t *goals_host, void *stream)
{
    int rc = check_range(h, first, count, "t2d_inject");
    if (rc) return rc;
    if (side != 81 && side != 82) return fail(T2D_ERR_INVALID, "t2d_inject: side must be 81 or 82");
    if (!maze_host || !pos_host) return fail(T2D_ERR_INVALID, "t2d_inject: null buffer");
    if (count == 0) return T2D_OK;
    DeviceGuard guard(h->device);
    std::vector<uint32_t> tiles((size_t)count * kTileWords, 0u), pos((size_t)count), goals((size_t)count, 0u),
        cnt((size_t)count), zero((size_t)count, 0u), d2((size_t)count), navgoal((size_t)count);
    for (int i = 0; i < count; i++) {
        const uint8_t *m = maze_host + (size_t)i * side * side;
        uint32_t *t = tiles.data() + (size_t)i * kTileWords;
        for (int r = 0; r < side; r++)
            for (int c = 0; c < side; c++)
                if (m[r * side + c]) t[r * kRowWords + (c >> 5)] |= 1u << (c & 31);
        const int32_t *p = pos_host + (size_t)i * 4;
        for (int k = 0; k < 4; k++)
            if (p[k] < 0 || p[k] >= side) return fail(T2D_ERR_INVALID, "t2d_inject: position outside the map (env %d)", first + i);
        if (m[p[0] * side + p[1]] || m[p[2] * side + p[3]])
            return fail(T2D_ERR_INVALID, "t2d_inject: agent placed on a wall (env %d)", first + i);
        pos[(size_t)i] = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
        if (goals_host) {
            const int32_t *g = goals_host + (size_t)i * 4;
            goals[(size_t)i] = ((uint32_t)g[0] & 0xff) | (((uint32_t)g[1] & 0xff) << 8) | (((uint32_t)g[2] & 0xff) << 16) | (((uint32_t)g[3] & 0xff) << 24);
        }
        cnt[(size_t)i] = (uint32_t)side << 24;
        navgoal[(size_t)i] = (uint32_t)p[2] | ((uint32_t)p[3] << 8); // Nav: "standing on the goal" -> re-plan at the next step
        const int dr = p[2] - p[0], dc = p[3] - p[1];
        d2[(size_t)i] = (uint32_t)(dr * dr + dc * dc);
    }
    hipStream_t st = (hipStream_t)stream;
    DevState &s = h->s;
    const size_t nb = (size_t)count * sizeof(uint32_t);
    HIP_TRY(hipMemcpyAsync(s.maps + (size_t)first * kTileWords, tiles.data(), tiles.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(s.pos + first, pos.data(), nb, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(s.goals + first, goals.data(), nb, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(s.cnt + first, cnt.data(), nb, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(s.d2 + first, d2.data(), nb, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(s.navgoal + first, navgoal.data(), nb, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(s.plan + first, zero.data(), nb, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (first == 0 && count == s.n) h->reset_done = true;
    return T2D_OK;
}

extern "C" int t2d_inject_plan(t2d_handle *h, int env, const int32_t *plan_host, int len, int cursor, void *stream)
{
    int rc = check_range(h, env, 1, "t2d_inject_plan");
    if (rc) return rc;
    if (!plan_host || len < 1 || len > 10 || cursor < 0 || cursor >= len)
        return fail(T2D_ERR_INVALID, "t2d_inject_plan: len in [1,10], cursor in [0,len)");
    uint32_t p = ((uint32_t)len << 20) | ((uint32_t)cursor << 24);
    for (int i = 0; i < len; i++) {
        if (plan_host[i] < 0 || plan_host[i] > 3) return fail(T2D_ERR_INVALID, "t2d_inject_plan: action %d", plan_host[i]);
        p |= (uint32_t)plan_host[i] << (2 * i);
    }
    DeviceGuard guard(h->device);
    HIP_TRY(hipMemcpyAsync(h->s.plan + env, &p, sizeof(p), hipMemcpyHostToDevice, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return T2D_OK;
}

static int fetch(const uint32_t *dev, int first, int count, std::vector<uint32_t> &host, hipStream_t st)
{
    host.resize((size_t)count);
    HIP_TRY(hipMemcpyAsync(host.data(), dev + first, (size_t)count * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    return T2D_OK;
}

extern "C" int t2d_get_state(t2d_handle *h, int first, int count, int32_t *pos_host, int32_t *goals_host,
                             int32_t *c_far_host, int32_t *t_host, uint32_t *episode_host, int32_t *side_host,
                             uint32_t *d2_host, void *stream)
{
    int rc = check_range(h, first, count, "t2d_get_state");
    if (rc) return rc;
    if (count == 0) return T2D_OK;
    DeviceGuard guard(h->device);
    hipStream_t st = (hipStream_t)stream;
    std::vector<uint32_t> pos, goals, cnt, ep, d2;
    if ((rc = fetch(h->s.pos, first, count, pos, st))) return rc;
    if ((rc = fetch(h->s.goals, first, count, goals, st))) return rc;
    if ((rc = fetch(h->s.cnt, first, count, cnt, st))) return rc;
    if ((rc = fetch(h->s.episode, first, count, ep, st))) return rc;
    if ((rc = fetch(h->s.d2, first, count, d2, st))) return rc;
    HIP_TRY(hipStreamSynchronize(st));
    for (int i = 0; i < count; i++) {
        for (int k = 0; k < 4; k++) {
            if (pos_host) pos_host[i * 4 + k] = (int32_t)((pos[(size_t)i] >> (8 * k)) & 0xffu);
            if (goals_host) goals_host[i * 4 + k] = (int32_t)((goals[(size_t)i] >> (8 * k)) & 0xffu);
        }
        if (c_far_host) c_far_host[i] = (int32_t)(cnt[(size_t)i] & 0xffu);
        if (t_host) t_host[i] = (int32_t)((cnt[(size_t)i] >> 8) & 0xffffu);
        if (side_host) side_host[i] = (int32_t)(cnt[(size_t)i] >> 24);
        if (episode_host) episode_host[i] = ep[(size_t)i];
        if (d2_host) d2_host[i] = d2[(size_t)i];
    }
    return T2D_OK;
}

extern "C" int t2d_get_maps(t2d_handle *h, int first, int count, uint8_t *maps_host, void *stream)
{
    int rc = check_range(h, first, count, "t2d_get_maps");
    if (rc) return rc;
    if (!maps_host) return fail(T2D_ERR_INVALID, "t2d_get_maps: null buffer");
    if (count == 0) return T2D_OK;
    DeviceGuard guard(h->device);
    hipStream_t st = (hipStream_t)stream;
    std::vector<uint32_t> tiles((size_t)count * kTileWords);
    HIP_TRY(hipMemcpyAsync(tiles.data(), h->s.maps + (size_t)first * kTileWords, tiles.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (int i = 0; i < count; i++) {
        const uint32_t *t = tiles.data() + (size_t)i * kTileWords;
        uint8_t *m = maps_host + (size_t)i * T2D_MAX_SIDE * T2D_MAX_SIDE;
        for (int r = 0; r < T2D_MAX_SIDE; r++)
            for (int c = 0; c < T2D_MAX_SIDE; c++)
                m[r * T2D_MAX_SIDE + c] = (uint8_t)((t[r * kRowWords + (c >> 5)] >> (c & 31)) & 1u);
    }
    return T2D_OK;
}

extern "C" int t2d_get_target(t2d_handle *h, int first, int count, int32_t *plan_host, int32_t *len_host,
                              int32_t *cursor_host, int32_t *navgoal_host, void *stream)
{
    int rc = check_range(h, first, count, "t2d_get_target");
    if (rc) return rc;
    if (count == 0) return T2D_OK;
    DeviceGuard guard(h->device);
    hipStream_t st = (hipStream_t)stream;
    std::vector<uint32_t> plan, ng;
    if ((rc = fetch(h->s.plan, first, count, plan, st))) return rc;
    if ((rc = fetch(h->s.navgoal, first, count, ng, st))) return rc;
    HIP_TRY(hipStreamSynchronize(st));
    for (int i = 0; i < count; i++) {
        uint32_t p = plan[(size_t)i];
        if (plan_host)
            for (int k = 0; k < 10; k++) plan_host[i * 10 + k] = (int32_t)((p >> (2 * k)) & 3u);
        if (len_host) len_host[i] = (int32_t)((p >> 20) & 15u);
        if (cursor_host) cursor_host[i] = (int32_t)((p >> 24) & 15u);
        if (navgoal_host) { navgoal_host[i * 2] = (int32_t)(ng[(size_t)i] & 0xffu); navgoal_host[i * 2 + 1] = (int32_t)((ng[(size_t)i] >> 8) & 0xffu); }
    }
    return T2D_OK;
}

extern "C" int t2d_get_faults(t2d_handle *h, uint32_t *faults_host, void *stream)
{
    if (!h || !faults_host) return fail(T2D_ERR_INVALID, "t2d_get_faults: null argument");
    DeviceGuard guard(h->device);
    HIP_TRY(hipMemcpyAsync(faults_host, h->s.faults, sizeof(uint32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return T2D_OK;
}

extern "C" int t2d_reward_table(t2d_handle *h, const uint32_t *d2_dev, int n, double w_p, float *r_track_dev,
                                float *r_target_dev, void *stream)
{
    if (!h || !d2_dev || !r_track_dev || !r_target_dev || n < 0) return fail(T2D_ERR_INVALID, "t2d_reward_table: bad argument");
    if (n == 0) return T2D_OK;
    DeviceGuard guard(h->device);
    hipLaunchKernelGGL(k_reward_table, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d2_dev, n,
                       w_p, r_track_dev, r_target_dev);
    HIP_TRY(hipGetLastError());
    return T2D_OK;
}
