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
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/atr_policy.h"
#include "../../include/track2d.h"
#include "../../include/track2d_np.h"
#include "atr_cell.h"
#include "coop_gemm.h"
#include "t2d_device.h"

#ifndef T2D_EXP
#define T2D_EXP 0
#endif

namespace t2d {

struct DevState {
    // ---- current episode (read/written by the step kernel) ----
    uint32_t *maps;     // [N][256] bit-packed tiles
    uint32_t *pos;      // [N] tracker r | c<<8 | target r<<16 | c<<24
    uint32_t *goals;    // [N] goal0 r | c<<8 | goal1 r<<16 | c<<24
    uint32_t *cnt;      // [N] c_far | t<<8 | side<<24
    uint32_t *cfg;      // [N] map_type | target_mode<<2 | level<<5
    uint32_t *episode;  // [N]
    uint32_t *plan;     // [N] scripted-target plan word
    uint32_t *tctr;     // [N] TARGET stream word counter
    uint32_t *navgoal;  // [N] r | c<<8
    uint32_t *nav2;     // [N] RPF patrol: virtual position r | c<<8 | remaining plan steps<<16 (14 bits) | vector<<30
    uint32_t *d2;       // [N] last squared distance
    uint32_t *dirf;     // [N][512] Nav direction planes (only if some env has a Nav target)
    // ---- the next TWO episodes of every env, generated ahead of time by k_gen. Every array is [2][N]...: slot p holds the
    // lowest not yet started episode number of parity p (episode[e] + 1 sits in slot (episode[e] + 1) & 1, episode[e] + 2 in
    // the other one). An episode lasts >= 11 steps, so an env uses at most two slots in any 20 steps: ONE generator pass per
    // 20 steps (one A3C rollout) keeps both valid.
    uint32_t *n_maps, *n_pos, *n_goals, *n_plan, *n_tctr, *n_navgoal, *n_nav2, *n_d2, *n_dirf;
    uint32_t *n_win;    // [2][N][32] the 2 x 13 window rows (13 map bits each) of that episode's FIRST observation
    uint32_t *gen_req;  // [2][N] 0 = slot valid; s > 0 = consumed at step stamp s, to be regenerated
    // ---- Nav targets: the NEXT plan of the current episode, prepared ahead of time by the generator pass ----
    // A queue of up to TWO prepared plans per env and EPISODE (the next goal's plan and the one after it), in three rotating
    // sets indexed by episode % 3: the current episode's queue and those the generator prepared for the two pre-generated next
    // episodes never share storage. Queue slot q of episode ep of env e = index ((ep % 3) * 2 + q) * N + e.
    uint32_t *p_field;  // [3][2][N][768] direction planes + visited plane of the BFS rooted at p_goal
    uint32_t *p_goal;   // [3][2][N] r | c<<8
    uint32_t *p_tctr;   // [3][2][N] TARGET stream word counter after drawing p_goal
    uint32_t *p_state;  // [3][N] bits 0-1: prepared plans (0..2), bit 2: queue head slot, bit 3: the last prepared plan does
                        //     not reach its start cell (it will be re-made inline, with more draws: nothing can follow it)
    // ---- Maze maps grown AHEAD of the generator pass (k_pregrow; handles with Maze envs only, else null) ----
    // A maze is a pure function of (seed, global env id, episode number, level): its growth — the one long serial chain of a
    // generated Maze episode — need not wait for the pass that builds the rest of the episode around it. Ring of four per env,
    // entry episode % 4; g_ep = the episode number whose maze the entry holds (kPoolEmpty: none / being written). An entry is
    // only ever overwritten once its episode number is <= the env's current one (dead: started long ago), and the entries
    // wanted next (current + 3, current + 4) then always fall on dead ones — so the pass and k_pregrow need no ordering
    // between them beyond the tag's release / acquire: a pass that finds another tag grows the maze itself, as before.
    uint32_t *np_mt;    // [N][kNpStateWords] per-env MT19937 state (t2d_np_attach: the generator draws from numpy-legacy streams), else null
    unsigned char *np_nav;  // [N][kNavBytes] Navigator plan + A* scratch (t2d_np_attach on a handle with Nav targets), else null
    uint32_t *g_maps;   // [4][N][256]
    uint32_t *g_ep;     // [4][N]
    uint32_t *pg_stats; // [4] pass: mazes taken from the pool, mazes grown inline; k_pregrow: mazes grown, entries left alone
    uint32_t *faults;   // [1]
    const float2 *rew_lut;    // [3][kLutN] (r_track, r_target) as float32(float64 formula), by w_p class and d^2
    int n;
    uint32_t env_base, k0, k1;
    int max_steps, auto_reset;
    int amask;                // highest action: 3 ('VonNeumann') or 7 ('Moore', track_1v1.py:243-249)
    int obs_full, obs_side;   // obs_type 'Full': every env writes [2][obs_side][obs_side] floats
    int nt_obs;               // large batches: observations are streamed past the caches (non-temporal stores)
};

enum : int { OP_STEP = 0, OP_RESET = 1, OP_OBSERVE = 2 };
constexpr uint32_t kPoolEmpty = 0xffffffffu;
constexpr int kNpStateWords = 640;           // 624 words of MT19937 state, the read position at [624], padding
// ... of which two words carry per-env facts of a numpy-stream handle: [625] the squared distance of the env's last TERMINAL
// step (info['distance'] of a step whose in-launch auto-reset has already replaced d2), [626] != 0: the env's target is the
// reference's RamAgent, stepped from this stream by k_ram_np (t2d_np_attach on a handle with T2D_TGT_RAM envs)
constexpr int kNpTermD2 = 625, kNpRamFlag = 626;     // [626]: 1 = RamAgent, 2 = Navigator (heap A*), stepped by k_tgt_np
// ... and, Navigator envs: [627] plan length, [628] plan cursor (a_i), [629] nav goal (r | c << 8). The plan's actions and the
// A* search's arrays live in a per-env scratch block (DevState::np_nav, t2d_np_attach):
constexpr int kNpPlanLen = 627, kNpPlanCur = 628, kNpNavGoal = 629, kNpRpfVector = 630;     // ([626] = 3: the RPF patrol Navigator)
constexpr int kNavPlanCap = 4096, kNavNodeCap = 32768, kNavHeapCap = 16384, kNavCells = 82 * 82;
// per-env scratch layout (bytes): plan u8 [4096] | nodes u64 [32768] | heap f f64 [16384] | heap node u32 [16384] |
// in_frontier i32 [6724 -> 6728] | explored u8 [6724 -> 6728]
constexpr size_t kNavOffNodes = kNavPlanCap, kNavOffHf = kNavOffNodes + (size_t)kNavNodeCap * 8,
                 kNavOffHn = kNavOffHf + (size_t)kNavHeapCap * 8, kNavOffInf = kNavOffHn + (size_t)kNavHeapCap * 4,
                 kNavOffExp = kNavOffInf + (size_t)6728 * 4, kNavBytes = kNavOffExp + 6728;
constexpr uint32_t kFaultNavOverflow = 16u;      // fault bit 4: the device A* ran out of node / heap / plan space
// The reward is a pure function of the integer squared distance (<= 2 * 81^2) and the mode's w_p in {0, 1, -0.5}
// (track_1v1.py:96-104,147-152): the float64 formula is evaluated once per handle into a table by the same
// reward_f64 device code the exhaustive parity test checks against the oracle; the step kernel then replaces a
// dependent f64 sqrt/divide chain that every lane would execute redundantly by one 8-byte scalar load.
constexpr int kLutN = 2 * 81 * 81 + 1;

// Action fetch in two halves so that the load is issued at the top of the kernel, next to the state and tile loads
// (one memory round trip for all of them), and only checked where the action is used.
template <int ADT> __device__ __forceinline__ long long load_action_raw(const void *p, int e)
{
    if (ADT == T2D_ACT_U8) return reinterpret_cast<const uint8_t *>(p)[e];
    if (ADT == T2D_ACT_I32) return reinterpret_cast<const int32_t *>(p)[e];
    return reinterpret_cast<const long long *>(p)[e];
}
__device__ __forceinline__ int check_action(long long v, int amask, uint32_t *faults)
{
    if (v < 0 || v > amask) { atomicOr(faults, 1u); v &= amask; }
    return (int)v;
}
// _next_state's transition table (track_1v1.py:275-279): the four VonNeumann moves are the first four of the Moore
// table {0:(-1,0) 1:(+1,0) 2:(0,-1) 3:(0,+1) 4:(-1,+1) 5:(+1,+1) 6:(-1,-1) 7:(+1,-1)}; two bits per action and axis.
__device__ __forceinline__ int move_dy(int a) { return (int)((0x8858u >> (2 * a)) & 3u) - 1; }
__device__ __forceinline__ int move_dx(int a) { return (int)((0x0a85u >> (2 * a)) & 3u) - 1; }

// Navigator.reset / the re-plan branch of Navigator.step (navigator.py:43-63, :15-38): plan from (fr, fc) to navgoal;
// unreachable or empty plan -> resample the goal, the 6th failure -> plan B (10 random actions).
// RPF: goals cycle the four patrol cells without random draws (generators.py:48-50), the plan is made on the
// generator's map (patrol cells free) and nav2 receives the open-loop plan: virtual position = (fr, fc), remaining =
// BFS distance, vector. (rpf + reference instead of a nullable pointer: the latter kept nav2 in scratch memory.)
template <class S>
__device__ __forceinline__ void nav_plan(const uint32_t *tile, int side, int lane, int fr, int fc, const FreeIndex &fi,
                                         uint32_t &navgoal, S &ts, uint32_t &plan, NavField &f, bool rpf, uint32_t &nav2)
{
    int count_res = 0;
    bool planb = false;
    uint32_t vector = rpf ? (nav2 >> 30) : 0u;
    int dist = -1;
    for (;;) {
        const int gr = (int)(navgoal & 0xffu), gc = (int)(navgoal >> 8);
        dist = bfs_dir_field(tile, side, lane, gr, gc, f, rpf, fr, fc);   // stops once (fr, fc) is reached
        const bool ok = rowbits_get(f.visA, f.visB, fr, fc) != 0u && !(fr == gr && fc == gc);
        if (ok) break;
        if (++count_res > 5) { planb = true; break; }
        if (rpf) { vector = (vector + 1u) & 3u; navgoal = rpf_cell(side, (int)vector); }
        else navgoal = select_free(tile, side, fi, (int)ts.bounded((uint32_t)(fi.total - 1)), lane);
    }
    plan = planb ? (plan_random(ts, 10u) | (1u << 28)) : 0u;
    if (rpf)
        nav2 = (uint32_t)fr | ((uint32_t)fc << 8) | ((planb ? 0u : ((uint32_t)dist & 0x3fffu)) << 16) | (vector << 30);
}
__device__ __forceinline__ uint32_t nav_dir_from_regs(const NavField &f, int r, int c)
{
    return rowbits_get(f.d0A, f.d0B, r, c) | (rowbits_get(f.d1A, f.d1B, r, c) << 1);
}

__device__ __forceinline__ int side_of_cfg(uint32_t cfg) { return (cfg & 3u) == (uint32_t)MAP_MAZE ? 81 : 82; }

// Nav targets: prepare the next plans on the CURRENT map, off the step kernel's critical path. A plan is needed when the
// target stands on its goal (navigator.py:15); the Navigator then draws a new goal (navigator.py:17 — the TARGET stream has no
// other consumer in between, so drawing it NOW keeps the stream order) and plans from where it stands, i.e. from the goal of
// the plan before. Up to two plans are queued: the second one starts at the first one's goal, from the stream position the
// first one's draw left — valid as long as the first one turns out usable (reachable start, goal != start), which is known
// here; otherwise the step kernel will re-plan inline with further draws and nothing can be prepared beyond it (bit 3).
// The step kernel adopts the queue head if the target's position is reachable in it and differs from its goal, else re-plans
// inline from the same already-drawn goal and drops the queue. Reads the env's live state: only ever runs in order on the
// caller's stream, between two step launches (inside the in-order generator launch, or as its own launch when the generator
// is forked). With a pass every 20 steps an inline re-plan needs two goals reached within those 20 steps.
__device__ __forceinline__ uint32_t pq_count(uint32_t ps) { return ps & 3u; }
__device__ __forceinline__ uint32_t pq_head(uint32_t ps) { return (ps >> 2) & 1u; }
__device__ __forceinline__ uint32_t pq_pop(uint32_t ps)      // the head plan was adopted
{
    const uint32_t c = pq_count(ps) - 1u;
    return c == 0u ? 0u : (c | ((pq_head(ps) ^ 1u) << 2) | (ps & 8u));
}
__device__ __forceinline__ size_t pq_index(const DevState &s, uint32_t episode, uint32_t qslot, int e)
{
    return (size_t)((episode % 3u) * 2u + qslot) * s.n + e;
}
__device__ __forceinline__ size_t pq_state_index(const DevState &s, uint32_t episode, int e)
{
    return (size_t)(episode % 3u) * s.n + e;
}
// Top the plan queue of episode `episode` of env e up to two plans, on the map in `tile` (LDS; free index `fi`). ps: the
// queue's state; (start, ctr): where the target will stand when the next plan is needed and the stream position its goal is
// drawn from — the current goal / counter, or those of the one plan already queued; carried in registers from there on.
// Returns the new state (the caller stores it).
__device__ __forceinline__ uint32_t nav_fill_queue(const DevState &s, int e, uint32_t episode, const uint32_t *tile, int side,
                                                    const FreeIndex &fi, int lane, uint32_t ps, uint32_t start, uint32_t ctr)
{
    const uint32_t genv = s.env_base + (uint32_t)e;
    while (pq_count(ps) < 2u && (ps & 8u) == 0u) {
        const uint32_t cnt = pq_count(ps), head = pq_head(ps);
        const size_t slot = pq_index(s, episode, (head + cnt) & 1u, e);
        Stream ts;
        ts.init(s.k0, s.k1, episode, genv, STREAM_TARGET, ctr);
        const uint32_t g2 = select_free(tile, side, fi, (int)ts.bounded((uint32_t)(fi.total - 1)), lane);
        NavField nf;
        // the flood may stop once it has reached the start cell; if the target is elsewhere when the plan is needed (plan B),
        // the adoption test in the step kernel (visited plane) sends it to the inline re-plan
        const int sr = (int)(start & 0xffu), sc = (int)(start >> 8);
        bfs_dir_field(tile, side, lane, (int)(g2 & 0xffu), (int)(g2 >> 8), nf, false, sr, sc);
        store_plan_field(s.p_field + slot * kPlanWords, nf, side, lane);
        const bool usable = rowbits_get(nf.visA, nf.visB, sr, sc) != 0u && g2 != start;
        if (lane == 0) { s.p_goal[slot] = g2; s.p_tctr[slot] = ts.ctr; }
        ps = (cnt + 1u) | (head << 2) | (usable ? 0u : 8u);
        start = g2; ctr = ts.ctr;
    }
    return ps;
}
__device__ __forceinline__ void nav_prefetch(const DevState &s, int e, uint32_t *tile, int lane)
{
    const uint32_t episode = s.episode[e];
    const size_t si = pq_state_index(s, episode, e);
    uint32_t ps = s.p_state[si];
    if (pq_count(ps) >= 2u || (ps & 8u) != 0u) return;
    reinterpret_cast<uint4 *>(tile)[lane] = reinterpret_cast<const uint4 *>(s.maps + (size_t)e * kTileWords)[lane];
    wave_lds_sync();
    const int side = (int)(s.cnt[e] >> 24);
    const FreeIndex fi = build_free_index(tile, side, lane);
    const size_t hs = pq_index(s, episode, pq_head(ps), e);
    const uint32_t start = pq_count(ps) == 0u ? s.navgoal[e] : s.p_goal[hs];
    const uint32_t ctr = pq_count(ps) == 0u ? s.tctr[e] : s.p_tctr[hs];
    ps = nav_fill_queue(s, e, episode, tile, side, fi, lane, ps, start, ctr);
    if (lane == 0) s.p_state[si] = ps;
    wave_lds_sync();
}

// The maze of (global env id, episode): the MAP stream's first draw is the density (init_maze, track_1v1.py:219-224), the rest
// is the growth — a pure function of its arguments, shared by the generator pass and k_pregrow.
__device__ __forceinline__ void grow_maze(const DevState &s, uint32_t genv, uint32_t episode, int level, uint32_t *tile,
                                          uint32_t *mlog, int lane)
{
    VStream ms;
    ms.init(s.k0, s.k1, episode, genv, STREAM_MAP, 0, lane);
    const double r = level > 0 ? (double)level * 0.02 : .03 * ms.next_double();
    gen_maze(tile, lane, ms, r, mlog);
}

// Track1v1Env.reset -> init_maze (track_1v1.py:134-168,218-240) for one env and one episode number, executed by
// one wave on an LDS tile. All arguments are wave-uniform. `gdir` receives the Nav direction planes.
// DEFER_NAV: stop before the Nav target's plans (the caller makes them: k_gen_nav spreads the three floods over three waves);
// fi_out receives the free-cell index of the finished map either way.
template <bool NAV, bool DEFER_NAV = false>
__device__ __forceinline__ void generate_episode(const DevState &s, int e, uint32_t *tile, uint32_t *mlog, int lane, uint32_t cfg,
                                                 uint32_t episode, uint32_t *gdir, uint32_t &pos, uint32_t &goals,
                                                 uint32_t &plan, uint32_t &tctr, uint32_t &navgoal, uint32_t &d2,
                                                 uint32_t &nav2, FreeIndex &fi_out, uint32_t *stamps = nullptr)
{
    // T2D_EXP == 9 (probe build only, tools/gen_timeline_probe.py): s_memtime stamps of the generator's phases
#define T2D_GSTAMP(i) do { if (T2D_EXP == 9 && lane == 0) stamps[i] = (uint32_t)__builtin_readcyclecounter(); } while (0)
    T2D_GSTAMP(0);
    nav2 = 0u;
    const int map_type = cfg & 3, mode = (cfg >> 2) & 7, level = (cfg >> 5) & 15;
    const uint32_t genv = s.env_base + (uint32_t)e;
    VStream ms;
    ms.init(s.k0, s.k1, episode, genv, STREAM_MAP, 0, lane);
    const int side = side_of_cfg(cfg);
    if (map_type == MAP_MAZE) {
        bool pooled = false;
        if (s.g_maps) {      // grown ahead of time by k_pregrow? (the tag is published after the tile: acquire it)
            const size_t po = (size_t)(episode & 3u) * s.n + e;
            pooled = __hip_atomic_load(s.g_ep + po, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == episode;
            if (pooled) {
                reinterpret_cast<uint4 *>(tile)[lane] = reinterpret_cast<const uint4 *>(s.g_maps + po * kTileWords)[lane];
                wave_lds_sync();
            }
            if (lane == 0) atomicAdd(s.pg_stats + (pooled ? 0 : 1), 1u);
        }
        if (!pooled) grow_maze(s, genv, episode, level, tile, mlog, lane);
    } else if (map_type == MAP_BLOCK) {
        double r = level > 0 ? (double)level * 0.05 : 0.15 * ms.next_double();
        gen_block(tile, lane, ms, r);
    } else {
        gen_block(tile, lane, ms, 0.0);
    }
    T2D_GSTAMP(1);
    const FreeIndex fi = build_free_index(tile, side, lane);
    fi_out = fi;
    const int n = fi.total;
    T2D_GSTAMP(2);
    VStream ss;
    ss.init(s.k0, s.k1, episode, genv, STREAM_SPAWN, 0, lane);
    uint32_t g0, g1;
    auto sample_goal2 = [&]() { // MazeGenerator.sample_goal(2), generators.py:38-51
        int i0 = (int)ss.bounded((uint32_t)(n - 1));
        int i1 = (int)ss.bounded((uint32_t)(n - 2));
        if (i1 >= i0) i1++;
        g0 = select_free(tile, side, fi, i0, lane);
        g1 = select_free(tile, side, fi, i1, lane);
    };
    // RPF (static goals): both goals = patrol cell 1, tracker spawn = patrol cell 0, no draws for either; the spawn
    // window is searched on the generator's map, where the patrol cells are free (generators.py:12-19,48-50,68)
    const bool rpf = mode == TGT_RPF;
    if (rpf) g0 = g1 = rpf_cell(side, 1);
    else sample_goal2();
    // sample_close_states(2, 1), generators.py:53-77 + get_around :82-94 (2x2 block up-left of the tracker)
    const uint32_t tr = rpf ? rpf_cell(side, 0) : select_free(tile, side, fi, (int)ss.bounded((uint32_t)(n - 1)), lane);
    const int r = (int)(tr & 0xffu), c = (int)(tr >> 8);
    const int x0 = max(0, r - 1), x1 = min(side - 1, r + 1), y0 = max(0, c - 1), y1 = min(side - 1, c + 1);
    auto gen_free = [&](int rr, int cc) { return tile_bit(tile, rr, cc) == 0u || (rpf && rr == r && cc == c); };
    int m = 0;
    for (int rr = x0; rr < x1; rr++)
        for (int cc = y0; cc < y1; cc++) m += (int)gen_free(rr, cc);
    int j = (int)ss.bounded((uint32_t)(m - 1));
    uint32_t tg = tr;
    for (int rr = x0; rr < x1; rr++)
        for (int cc = y0; cc < y1; cc++)
            if (gen_free(rr, cc)) {
                if (j == 0) tg = (uint32_t)rr | ((uint32_t)cc << 8);
                j--;
            }
    while (!rpf && (tr == g0 || tr == g1)) sample_goal2(); // goal_test loop, track_1v1.py:239-240
    pos = tr | (tg << 16);
    goals = g0 | (g1 << 16);
    T2D_GSTAMP(3);
    VStream ts;
    ts.init(s.k0, s.k1, episode, genv, STREAM_TARGET, 0, lane);
    plan = 0;
    navgoal = g1;
    if (mode == TGT_RAM) plan = ram_reset(ts);
    if (NAV && (mode == TGT_NAV || rpf) && !(DEFER_NAV && !rpf)) { // Navigator.reset (navigator.py:43-63): plan from the target spawn to goal_states[1]
        NavField nf;
        nav2 = 1u << 30;                   // RPF: sample_goal(2) at reset advanced the patrol vector to 1
        nav_plan(tile, side, lane, (int)(tg & 0xffu), (int)(tg >> 8), fi, navgoal, ts, plan, nf, rpf, nav2);
        if (!rpf) nav2 = 0u;
        if (((plan >> 28) & 1u) == 0u) store_dir_field(gdir, nf, side, lane);
        if (!rpf) {
            // ... and the two plans AFTER it, into the new episode's own queue (set episode % 3): a fresh episode's first path
            // is often shorter than the 20 steps to the next generator pass, and then the step kernel would have to re-plan
            // inline (measured: every inline re-plan of a random-policy batch came from an episode <= 20 steps old)
            const uint32_t ps = nav_fill_queue(s, e, episode, tile, side, fi, lane, 0u, navgoal, ts.ctr);
            if (lane == 0) s.p_state[pq_state_index(s, episode, e)] = ps;
        }
    }
    tctr = ts.ctr;
    const int dr = (int)(tg & 0xffu) - r, dc = (int)(tg >> 8) - c;
    d2 = (uint32_t)(dr * dr + dc * dc);
    T2D_GSTAMP(4);
#undef T2D_GSTAMP
}

// The 13 map bits [c - 6, c + 6] of one row given as its three words (ones outside the map: np.pad(..., 1),
// track_1v1.py:316-322): one v_alignbit over the window [ones | w0 | w1 | w2' | ones].
__device__ __forceinline__ uint32_t window_row_bits(uint32_t w0, uint32_t w1, uint32_t w2, int side, int c)
{
    const uint32_t W2 = w2 | ~valid_mask_w2(side);           // columns >= side read as 1
    const int sp = c - T2D_POB + 32;
    const int j = sp >> 5;
    const uint32_t lo = j == 0 ? 0xffffffffu : (j == 1 ? w0 : (j == 2 ? w1 : W2));
    const uint32_t hi = j == 0 ? w0 : (j == 1 ? w1 : (j == 2 ? W2 : 0xffffffffu));
    return __builtin_amdgcn_alignbit(hi, lo, (uint32_t)(sp & 31)) & 0x1fffu;
}

// The map side of a generated next-episode slot: the tile and the window rows of the episode's first observation, so that the
// step kernel that switches to this episode needs no dependent map fetch (k_step2 reads them speculatively when a done is
// possible). One wave; the tile is complete in LDS.
__device__ __forceinline__ void store_slot_map(const DevState &s, size_t so, const uint32_t *tile, int lane, uint32_t cfg, uint32_t pos)
{
    reinterpret_cast<uint4 *>(s.n_maps + so * kTileWords)[lane] = reinterpret_cast<const uint4 *>(tile)[lane];
    if (lane < 2 * T2D_WIN) {
        const int ag = lane >= T2D_WIN ? 1 : 0, y = lane - ag * T2D_WIN;
        const int gside = side_of_cfg(cfg);
        const int ar = (int)((pos >> (16 * ag)) & 0xffu), ac = (int)((pos >> (16 * ag + 8)) & 0xffu);
        const int rr = ar - T2D_POB + y;
        uint32_t bits = 0x1fffu;
        if ((unsigned)rr < (unsigned)gside) {
            const uint32_t *w = tile + rr * kRowWords;
            bits = window_row_bits(w[0], w[1], w[2], gside, ac);
        }
        s.n_win[so * 32 + lane] = bits;
    }
}

// Generator kernel: fills the "next episode" slot (episode[e] + 1) of every env whose slot was consumed at a step
// stamp in [lo, hi] (or of every env when force != 0). Launched by the host once per stamp window: a consumed slot
// cannot be needed again for 11 steps (done needs 11 consecutive far steps, track_1v1.py:106-111), so generation is
// off the step kernel's critical path. Everything it reads (gen_req, cfg, episode of the envs it serves) was written
// before its launch and is not touched by the step launches of the following window; everything it writes is read
// only after the window's join (t2d_handle::gen_*): it may therefore run on the library's side stream, under the
// step launches and policy kernels of the next window.
template <bool NAV, bool PREFETCH>
__global__ __launch_bounds__(256) void k_gen(DevState s, uint32_t lo, uint32_t hi, int force)
{
    __shared__ __attribute__((aligned(16))) uint32_t tiles[kWavesPerBlock][kTileWords];
    __shared__ uint32_t mlogs[kWavesPerBlock][kMazeLogMax];     // move log of the maze generator (t2d_device.h gen_maze)
    const int lane = (int)(threadIdx.x & 63u);
    const int wave = uni((int)(threadIdx.x >> 6));
    const int idx = (int)blockIdx.x * kWavesPerBlock + wave;       // (slot, env): the first N waves serve slot 0, the next N
    if (idx >= (PREFETCH ? 3 : 2) * s.n) return;                    // slot 1; PREFETCH: N more waves prepare Nav plans
    uint32_t *tile = tiles[wave];
    if (PREFETCH && idx >= 2 * s.n) {       // (waves of their own: a prefetch and a generation are each one wave's serial chain)
        const int ep = idx - 2 * s.n;
        if (!force && (int)((s.cfg[ep] >> 2) & 7u) == TGT_NAV) nav_prefetch(s, ep, tile, lane);
        return;
    }
    const int slot = idx >= s.n ? 1 : 0, e = idx - slot * s.n;
    const size_t so = (size_t)slot * s.n + e;
    const uint32_t req = s.gen_req[so];
    const uint32_t cfg = s.cfg[e];
    const bool need_gen = force || (req >= lo && req <= hi && req != 0u);
    if (!need_gen) return;
    // the episode this slot is to hold: the lowest number above the env's current one with the slot's parity
    const uint32_t cur = s.episode[e];
    const uint32_t target = ((cur + 1u) & 1u) == (uint32_t)slot ? cur + 1u : cur + 2u;
    uint32_t pos, goals, plan, tctr, navgoal, d2, nav2;
    uint32_t *gdir = NAV ? s.n_dirf + so * kDirWords : nullptr;
#if T2D_EXP == 9
    __shared__ uint32_t gst[kWavesPerBlock][8];
    if (lane == 0) gst[wave][6] = (uint32_t)__builtin_readcyclecounter();            // kernel entry of this wave (incl. prefetch)
    FreeIndex fi_unused;
    generate_episode<NAV>(s, e, tile, mlogs[wave], lane, cfg, target, gdir, pos, goals, plan, tctr, navgoal, d2, nav2, fi_unused, gst[wave]);
    wave_lds_sync();
    if (lane == 0) { gst[wave][5] = (uint32_t)__builtin_readcyclecounter(); for (int i = 0; i < 7; i++) tile[246 + i] = gst[wave][i]; }
#else
    FreeIndex fi_unused;
    generate_episode<NAV>(s, e, tile, mlogs[wave], lane, cfg, target, gdir, pos, goals, plan, tctr, navgoal, d2, nav2, fi_unused);
#endif
    wave_lds_sync();
    store_slot_map(s, so, tile, lane, cfg, pos);
    if (lane == 0) {
        s.n_pos[so] = pos; s.n_goals[so] = goals; s.n_plan[so] = plan; s.n_tctr[so] = tctr;
        s.n_navgoal[so] = navgoal; s.n_d2[so] = d2; s.gen_req[so] = 0u;
        if (NAV) s.n_nav2[so] = nav2;
    }
}

// The generator pass of handles with Nav targets (navigator.py:5-70): ONE WORKGROUP per (slot, env) instead of one wave.
// A generated Nav episode needs three goal-rooted floods — Navigator.reset's plan (target spawn -> goal_states[1]) and the two
// plans queued behind it (nav_fill_queue) — and on one wave they were a serial chain behind the map (Maze: <= 110 us of growth
// + 3 x 43 us). They only LOOK dependent: plan k + 1 starts where plan k's goal draw left the TARGET stream, and that position
// depends on plan k's flood only when the flood FAILS (unreachable goal or goal == start: re-draws). So the common case is
// speculated: wave 0 builds the map, picks spawns and goals and draws the two further goals as if every plan succeeded at once;
// then waves 0 / 1 / 2 run the three floods side by side on the shared LDS tile while wave 3 writes the slot's map out; wave 0
// validates afterwards and, if the FIRST plan failed (the only case in which the speculated draws are wrong), redoes the serial
// chain exactly as k_gen does — the results are the oracle's either way, bit for bit (same draws in the same order).
// PREFETCH: further blocks top up the plan queues of the running episodes, one wave per env (nav_prefetch), as in k_gen.
template <bool PREFETCH>
__global__ __launch_bounds__(256) void k_gen_nav(DevState s, uint32_t lo, uint32_t hi, int force)
{
    __shared__ __attribute__((aligned(16))) uint32_t tiles[kWavesPerBlock][kTileWords];
    __shared__ uint32_t mlog[kMazeLogMax];
    __shared__ uint32_t xch[16];
    const int lane = (int)(threadIdx.x & 63u);
    const int wave = uni((int)(threadIdx.x >> 6));
    const int b = (int)blockIdx.x;
    if (b >= 2 * s.n) {
        if (!PREFETCH) return;
        const int ep = (b - 2 * s.n) * kWavesPerBlock + wave;
        if (ep < s.n && !force && (int)((s.cfg[ep] >> 2) & 7u) == TGT_NAV) nav_prefetch(s, ep, tiles[wave], lane);
        return;
    }
    const int slot = b >= s.n ? 1 : 0, e = b - slot * s.n;
    const size_t so = (size_t)slot * s.n + e;
    const uint32_t req = s.gen_req[so];
    const uint32_t cfg = s.cfg[e];
    if (!(force || (req >= lo && req <= hi && req != 0u))) return;          // (uniform over the workgroup)
    const uint32_t cur = s.episode[e];
    const uint32_t target = ((cur + 1u) & 1u) == (uint32_t)slot ? cur + 1u : cur + 2u;
    uint32_t *tile = tiles[0];
    uint32_t *gdir = s.n_dirf + so * kDirWords;
    const int mode = (int)((cfg >> 2) & 7u), side = side_of_cfg(cfg);
    uint32_t pos = 0u, goals = 0u, plan = 0u, tctr = 0u, navgoal = 0u, d2 = 0u, nav2 = 0u;
    FreeIndex fi;
    if (mode != TGT_NAV) {        // other targets of a mixed handle (and RPF): one wave, the whole episode, as k_gen
        if (wave != 0) return;
        generate_episode<true>(s, e, tile, mlog, lane, cfg, target, gdir, pos, goals, plan, tctr, navgoal, d2, nav2, fi);
        wave_lds_sync();
        store_slot_map(s, so, tile, lane, cfg, pos);
        if (lane == 0) {
            s.n_pos[so] = pos; s.n_goals[so] = goals; s.n_plan[so] = plan; s.n_tctr[so] = tctr;
            s.n_navgoal[so] = navgoal; s.n_d2[so] = d2; s.gen_req[so] = 0u; s.n_nav2[so] = nav2;
        }
        return;
    }
    const uint32_t genv = s.env_base + (uint32_t)e;
#if T2D_EXP == 9     // probe build (tools/gen_nav_timeline_probe.py): s_memtime stamps of wave 0, left in the slot's spare tile words
    __shared__ uint32_t gst[16];
#define T2D_NSTAMP(i) do { if (wave == 0 && lane == 0) gst[i] = (uint32_t)__builtin_readcyclecounter(); } while (0)
#else
    uint32_t *gst = nullptr;
#define T2D_NSTAMP(i) do { } while (0)
#endif
    T2D_NSTAMP(6);
    if (wave == 0) {
        generate_episode<true, true>(s, e, tile, mlog, lane, cfg, target, gdir, pos, goals, plan, tctr, navgoal, d2, nav2, fi, gst);
        // the goals of the two queued plans, drawn as nav_fill_queue will draw them if the first plan needs no re-draw
        Stream t1;
        t1.init(s.k0, s.k1, target, genv, STREAM_TARGET, tctr);
        const uint32_t g2 = select_free(tile, side, fi, (int)t1.bounded((uint32_t)(fi.total - 1)), lane);
        const uint32_t c1 = t1.ctr;
        Stream t2;
        t2.init(s.k0, s.k1, target, genv, STREAM_TARGET, c1);
        const uint32_t g3 = select_free(tile, side, fi, (int)t2.bounded((uint32_t)(fi.total - 1)), lane);
        if (lane == 0) { xch[0] = pos; xch[1] = navgoal; xch[2] = g2; xch[3] = g3; xch[4] = c1; xch[5] = t2.ctr; }
    }
    T2D_NSTAMP(7);
    __syncthreads();
    const uint32_t x_pos = xch[0], x_g1 = xch[1], x_g2 = xch[2], x_g3 = xch[3];
    bool ok0 = false;
    if (wave == 3) {
        store_slot_map(s, so, tile, lane, cfg, x_pos);
    } else {
        // wave 0: spawn -> g1 (Navigator.reset); wave 1: g1 -> g2 (queue slot 0); wave 2: g2 -> g3 (queue slot 1)
        const uint32_t goal = wave == 0 ? x_g1 : (wave == 1 ? x_g2 : x_g3);
        const uint32_t start = wave == 0 ? (x_pos >> 16) : (wave == 1 ? x_g1 : x_g2);
        const int gr = (int)(goal & 0xffu), gc = (int)((goal >> 8) & 0xffu), sr = (int)(start & 0xffu), sc = (int)((start >> 8) & 0xffu);
        NavField nf;
        bfs_dir_field(tile, side, lane, gr, gc, nf, false, sr, sc);
        const bool usable = rowbits_get(nf.visA, nf.visB, sr, sc) != 0u && !(sr == gr && sc == gc);
        if (wave == 0) {
            ok0 = usable;
            if (usable) store_dir_field(gdir, nf, side, lane);
        } else {
            store_plan_field(s.p_field + pq_index(s, target, (uint32_t)(wave - 1), e) * kPlanWords, nf, side, lane);
            if (lane == 0) xch[8 + wave] = usable ? 1u : 0u;
        }
    }
    T2D_NSTAMP(8);
    __syncthreads();
    T2D_NSTAMP(9);
    if (wave != 0) return;
    uint32_t ps;
    if (ok0) {      // the speculation held: what nav_plan + nav_fill_queue would have left behind
        const bool u1 = xch[9] != 0u, u2 = xch[10] != 0u;
        ps = u1 ? (2u | (u2 ? 0u : 8u)) : (1u | 8u);
        if (lane == 0) {
            s.p_goal[pq_index(s, target, 0u, e)] = x_g2; s.p_tctr[pq_index(s, target, 0u, e)] = xch[4];
            s.p_goal[pq_index(s, target, 1u, e)] = x_g3; s.p_tctr[pq_index(s, target, 1u, e)] = xch[5];
        }
    } else {        // rare (an unreachable or coinciding first goal): the serial chain, re-draws and all
        NavField nf;
        VStream ts;
        ts.init(s.k0, s.k1, target, genv, STREAM_TARGET, tctr, lane);
        uint32_t nav2_unused = 0u;
        nav_plan(tile, side, lane, (int)((x_pos >> 16) & 0xffu), (int)(x_pos >> 24), fi, navgoal, ts, plan, nf, false, nav2_unused);
        if (((plan >> 28) & 1u) == 0u) store_dir_field(gdir, nf, side, lane);
        ps = nav_fill_queue(s, e, target, tile, side, fi, lane, 0u, navgoal, ts.ctr);
        tctr = ts.ctr;
    }
    if (lane == 0) {
        s.p_state[pq_state_index(s, target, e)] = ps;
        s.n_pos[so] = pos; s.n_goals[so] = goals; s.n_plan[so] = plan; s.n_tctr[so] = tctr;
        s.n_navgoal[so] = navgoal; s.n_d2[so] = d2; s.gen_req[so] = 0u; s.n_nav2[so] = 0u;
    }
#if T2D_EXP == 9
    if (lane == 0) {
        gst[5] = (uint32_t)__builtin_readcyclecounter();
        for (int i = 0; i < 10; i++) s.n_maps[so * kTileWords + 246 + i] = gst[i];
    }
#endif
#undef T2D_NSTAMP
}

// ---- numpy-exact episodes on the device (t2d_np_attach) ------------------------------------------------------------------
// The reference draws everything from numpy's legacy global stream (generators.py:28,44,61,90,131,140,166; track_1v1.py:
// 223,229): MT19937 words turned into doubles (random_sample: two words), bounded integers (masked rejection: one word per
// attempt) and, for every choice(n, k, replace=False), a WHOLE Fisher-Yates permutation(n) of which the first k entries are
// used — four permutations of ~6000 per Block reset. One wavefront per env restates that draw for draw: the MT state (624
// words) and the permutation (u16[6400]) live in LDS, the twist is done 64 words at a time, the shuffles are the serial loops
// they are (~0.7 ms each: this is the parity mode, environment.NumpyVecEnv(device_generators=True), not the throughput path).
struct NpStream {
    uint32_t *mt;      // LDS [624]
    int pos, lane;
    // the twist, in ten passes of 64: word k needs the OLD k + 1 (its neighbour in the same pass: everybody reads before anybody
    // writes) and word (k + 397) % 624 — old for k < 227, and for k >= 227 the new k - 227, written at least two passes earlier
    __device__ __forceinline__ void refill()
    {
#pragma unroll 1
        for (int p = 0; p < 10; p++) {
            const int k = p * 64 + lane;
            uint32_t nv = 0u;
            if (k < 624) {
                const uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
                nv = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            wave_lds_sync();
            if (k < 624) mt[k] = nv;
            wave_lds_sync();
        }
        pos = 0;
    }
    __device__ __forceinline__ uint32_t word()
    {
        if (pos >= 624) refill();
        uint32_t y = mt[pos++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return uni(y);
    }
    __device__ __forceinline__ double uniform()          // random_sample(): 53 bits out of two words
    {
        const uint32_t a = word() >> 5, b = word() >> 6;
        return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
    }
    __device__ __forceinline__ uint32_t upto(uint32_t top)   // uniform in [0, top]; top == 0 draws nothing
    {
        if (top == 0u) return 0u;
        uint32_t mask = top;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        uint32_t v;
        do { v = word() & mask; } while (v > top);
        return v;
    }
    __device__ __forceinline__ int randint(int low, int high) { return low + (int)upto((uint32_t)(high - 1 - low)); }
    __device__ __forceinline__ uint32_t bounded(uint32_t top) { return upto(top); }   // (the name ram_reset / ram_step draw by)
    // permutation(n): arange(n) shuffled from the top (i = n - 1 .. 1: swap with a uniform j in [0, i])
    __device__ __forceinline__ void permutation(int n, uint16_t *perm)
    {
        for (int i = lane; i < n; i += 64) perm[i] = (uint16_t)i;
        wave_lds_sync();
#pragma unroll 1
        for (int i = n - 1; i >= 1; i--) {
            const uint16_t a = perm[i];
            const int j = (int)upto((uint32_t)i);
            const uint16_t b = perm[j];
            wave_lds_sync();
            if (lane == 0) { perm[i] = b; perm[j] = a; }
            wave_lds_sync();
        }
    }
};

__device__ __forceinline__ void tile_set(uint32_t *tile, int r, int c, int lane)
{
    if (lane == 0) tile[r * kRowWords + (c >> 5)] |= 1u << (c & 31);
}

// One reset() of the reference env (track_1v1.py:134-168 -> init_maze :218-240) from the env's numpy stream.
// The four patrol cells of the RPF ids (static_goals, generators.py:12-19): (s/6, s/6), (5s/6, s/6), (5s/6, 5s/6), (s/6, 5s/6) as r | c << 8.
__device__ __forceinline__ uint32_t rpf_cand(int side, int k)
{
    const uint32_t lo = (uint32_t)(side / 6), hi = (uint32_t)(side * 5 / 6);
    const uint32_t r = (k == 1 || k == 2) ? hi : lo, c = (k >= 2) ? hi : lo;
    return r | (c << 8);
}
// Clear the patrol cells on the tile (the GENERATOR's map: track_1v1.py:233-236 copies the env's own map before static_goals frees
// them); returns which of them were walls, for rpf_restore.
__device__ __forceinline__ uint32_t rpf_clear(uint32_t *tile, int side, int lane)
{
    uint32_t saved = 0u;
    for (int k = 0; k < 4; k++) {
        const uint32_t cd = rpf_cand(side, k);
        saved |= tile_bit(tile, (int)(cd & 0xffu), (int)(cd >> 8)) << k;
    }
    wave_lds_sync();
    if (lane == 0)
        for (int k = 0; k < 4; k++) {
            const uint32_t cd = rpf_cand(side, k);
            tile[(cd & 0xffu) * kRowWords + ((cd >> 8) >> 5)] &= ~(1u << ((cd >> 8) & 31u));
        }
    wave_lds_sync();
    return saved;
}
__device__ __forceinline__ void rpf_restore(uint32_t *tile, int side, uint32_t saved, int lane)
{
    wave_lds_sync();
    for (int k = 0; k < 4; k++)
        if ((saved >> k) & 1u) { const uint32_t cd = rpf_cand(side, k); tile_set(tile, (int)(cd & 0xffu), (int)(cd >> 8), lane); }
    wave_lds_sync();
}

// rpf: the RPF ids — the tile is left as the GENERATOR's map (patrol cells freed: what the samplers and the Navigator see), `saved`
// says which of them the env's own map keeps as walls; `vector` = the patrol index static_goals starts at 0 and sample_goal advances.
__device__ __forceinline__ void generate_episode_np(NpStream &rs, uint16_t *perm, uint32_t *tile, int lane, uint32_t cfg,
                                                    uint32_t &pos, uint32_t &goals, uint32_t &d2, bool rpf = false,
                                                    uint32_t *saved = nullptr, uint32_t *vector = nullptr)
{
    const int map_type = cfg & 3, level = (cfg >> 5) & 15;
    const int side = side_of_cfg(cfg);
    tile_clear(tile, lane);
    wave_lds_sync();
    if (map_type == MAP_MAZE) {     // RandomMazeGenerator._generate_maze (generators.py:115-145), width = height = 80 -> 81 x 81
        const double r = level > 0 ? (double)level * 0.02 : .03 * rs.uniform();
        const int S = 81;
        const int complexity = (int)(r * (5.0 * (S + S))), density = (int)(r * (double)((S / 2) * (S / 2)));
        tile_border(tile, S, lane);
        wave_lds_sync();
#pragma unroll 1
        for (int i = 0; i < density; i++) {
            int x = rs.randint(0, S / 2 + 1) * 2;            // the tuple on generators.py:131 evaluates x's draw first
            int y = rs.randint(0, S / 2 + 1) * 2;
            tile_set(tile, y, x, lane);
            wave_lds_sync();
#pragma unroll 1
            for (int j = 0; j < complexity; j++) {
                int nr[4], nc[4], n = 0;
                if (x > 1) { nr[n] = y; nc[n] = x - 2; n++; }
                if (x < S - 2) { nr[n] = y; nc[n] = x + 2; n++; }
                if (y > 1) { nr[n] = y - 2; nc[n] = x; n++; }
                if (y < S - 2) { nr[n] = y + 2; nc[n] = x; n++; }
                if (n == 0) continue;
                const int k = rs.randint(0, n);
                int pr = nr[0], pc = nc[0];
#pragma unroll
                for (int q = 1; q < 4; q++) if (q == k) { pr = nr[q]; pc = nc[q]; }
                if (tile_bit(tile, pr, pc) == 0u) {
                    tile_set(tile, pr, pc, lane);
                    tile_set(tile, pr + (y - pr) / 2, pc + (x - pc) / 2, lane);
                    wave_lds_sync();
                    x = pc; y = pr;
                }
            }
        }
    } else {                        // RandomBlockMazeGenerator._generate_maze (generators.py:157-176); Empty: ratio 0
        const double r = map_type == MAP_BLOCK ? (level > 0 ? (double)level * 0.05 : 0.15 * rs.uniform()) : 0.0;
        const int K = (int)(r * 6400.0);
        rs.permutation(6400, perm);                          // (the whole shuffle is drawn even for K = 0)
        for (int i = lane; i < K; i += 64) {
            const uint32_t c = perm[i];
            const uint32_t row = c / 80u + 1u, col = c - (c / 80u) * 80u + 1u;
            atomicOr(&tile[row * kRowWords + (col >> 5)], 1u << (col & 31u));
        }
        wave_lds_sync();
        tile_border(tile, 82, lane);
        wave_lds_sync();
    }
    uint32_t vec = 0u;
    if (rpf) *saved = rpf_clear(tile, side, lane);
    const FreeIndex fi = build_free_index(tile, side, lane);
    const int n = fi.total;
    uint32_t g0, g1;
    auto sample_goal2 = [&]() {     // sample_goal(2): choice(len, 2, replace=False) = permutation(len)[:2]
        if (rpf) { vec = (vec + 1u) & 3u; g0 = g1 = rpf_cand(side, (int)vec); return; }     // (static: no draw, generators.py:48-50)
        rs.permutation(n, perm);
        const int i0 = uni((int)perm[0]), i1 = uni((int)perm[1]);
        g0 = select_free(tile, side, fi, i0, lane);
        g1 = select_free(tile, side, fi, i1, lane);
    };
    sample_goal2();
    // sample_close_states(2, 1): choice(len, 2, replace=False) of which only the first is used, then get_around, then
    // sample_state(0) = choice(len, 0, replace=False) — still a whole permutation
    rs.permutation(n, perm);                                 // (drawn in the static case too, then ignored: generators.py:61,68)
    const uint32_t tr = rpf ? rpf_cand(side, 0) : select_free(tile, side, fi, uni((int)perm[0]), lane);
    const int r = (int)(tr & 0xffu), c = (int)(tr >> 8);
    const int x0 = max(0, r - 1), x1 = min(side - 1, r + 1), y0 = max(0, c - 1), y1 = min(side - 1, c + 1);
    int m = 0;
    for (int rr = x0; rr < x1; rr++)
        for (int cc = y0; cc < y1; cc++) m += (int)(tile_bit(tile, rr, cc) == 0u);
    rs.permutation(m, perm);
    int j = uni((int)perm[0]);
    uint32_t tg = tr;
    for (int rr = x0; rr < x1; rr++)
        for (int cc = y0; cc < y1; cc++)
            if (tile_bit(tile, rr, cc) == 0u) {
                if (j == 0) tg = (uint32_t)rr | ((uint32_t)cc << 8);
                j--;
            }
    rs.permutation(n, perm);
    while (tr == g0 || tr == g1) sample_goal2();             // goal_test loop, track_1v1.py:239-240
    pos = tr | (tg << 16);
    goals = g0 | (g1 << 16);
    const int dr = (int)(tg & 0xffu) - r, dc = (int)(tg >> 8) - c;
    d2 = (uint32_t)(dr * dr + dc * dc);
    if (rpf) *vector = vec;
}

// ---- the reference's Navigator on the device: AstarSolver (Astar_solver.py:42-173) with heapq's exact sift order -----------
// What makes the reference's A* paths ITS paths are accidents of its implementation, all restated here (as in the host
// restatement csrc/np_mode.cpp, which the 49 reference searches of tests/golden/astar.npz pin): heap entries [f, node] compared
// like Python lists — by f = path cost + float64 Euclidean distance, ties by Node.__lt__ = smaller path cost (:30-32,55) —
// heapq._siftdown / _siftup's exact order, children in action order 0..3 with a wall bump returning the parent's state
// (skipped as explored, :138-145,170-171), and the INVERTED replace test (:146-147: a queued node that is CHEAPER than the
// child is replaced by it, every match in array order re-sifted). One lane runs the search (it is a sequential algorithm on
// a heap); nodes, heap and the two cell maps live in the env's scratch block in device memory.
struct AstarNp {
    unsigned long long *nodes;   // r (7) | c (7) << 7 | action (2) << 14 | cost (16) << 16 | prev (32) << 32
    double *hf;
    uint32_t *hn;
    int32_t *inf;                // cell -> node index or -1 (Frontier.state_nodes)
    uint8_t *expl;
    int nn, hs, gr, gc;
    bool overflow;
    __device__ __forceinline__ static int n_r(unsigned long long v) { return (int)(v & 127u); }
    __device__ __forceinline__ static int n_c(unsigned long long v) { return (int)((v >> 7) & 127u); }
    __device__ __forceinline__ static int n_act(unsigned long long v) { return (int)((v >> 14) & 3u); }
    __device__ __forceinline__ static int n_cost(unsigned long long v) { return (int)((v >> 16) & 0xffffu); }
    __device__ __forceinline__ static uint32_t n_prev(unsigned long long v) { return (uint32_t)(v >> 32); }
    __device__ __forceinline__ int add_node(int r, int c, int act, int cost, uint32_t prev)
    {
        if (nn >= kNavNodeCap) { overflow = true; return nn - 1; }
        nodes[nn] = (unsigned long long)r | ((unsigned long long)c << 7) | ((unsigned long long)act << 14) |
                    ((unsigned long long)cost << 16) | ((unsigned long long)prev << 32);
        return nn++;
    }
    __device__ __forceinline__ double f_of(int n) const      // :151-153: norm of an int vector = sqrt of an exact dot product
    {
        const unsigned long long v = nodes[n];
        const double dr = (double)(n_r(v) - gr), dc = (double)(n_c(v) - gc);
        return (double)n_cost(v) + sqrt(dr * dr + dc * dc);
    }
    __device__ __forceinline__ bool less(double fa, uint32_t na, double fb, uint32_t nb) const
    {
        if (fa != fb) return fa < fb;
        return n_cost(nodes[na]) < n_cost(nodes[nb]);
    }
    __device__ void sift_down(int start, int p)               // heapq._siftdown: towards the root
    {
        const double f = hf[p];
        const uint32_t n = hn[p];
        while (p > start) {
            const int parent = (p - 1) >> 1;
            if (!less(f, n, hf[parent], hn[parent])) break;
            hf[p] = hf[parent]; hn[p] = hn[parent];
            p = parent;
        }
        hf[p] = f; hn[p] = n;
    }
    __device__ void sift_up(int p)                            // heapq._siftup: to a leaf, then back
    {
        const int end = hs, start = p;
        const double f = hf[p];
        const uint32_t n = hn[p];
        int child = 2 * p + 1;
        while (child < end) {
            const int right = child + 1;
            if (right < end && !less(hf[child], hn[child], hf[right], hn[right])) child = right;
            hf[p] = hf[child]; hn[p] = hn[child];
            p = child;
            child = 2 * p + 1;
        }
        hf[p] = f; hn[p] = n;
        sift_down(start, p);
    }
    __device__ void push(int n)                               // Frontier.add
    {
        if (hs >= kNavHeapCap) { overflow = true; return; }
        hf[hs] = f_of(n); hn[hs] = (uint32_t)n;
        hs++;
        sift_down(0, hs - 1);
        const unsigned long long v = nodes[n];
        inf[n_r(v) * 82 + n_c(v)] = n;
    }
    __device__ int pop()                                      // Frontier.pop
    {
        const double lf = hf[hs - 1];
        const uint32_t ln = hn[hs - 1];
        hs--;
        uint32_t top = ln;
        if (hs > 0) {
            top = hn[0];
            hf[0] = lf; hn[0] = ln;
            sift_up(0);
        }
        const unsigned long long v = nodes[top];
        inf[n_r(v) * 82 + n_c(v)] = -1;
        return (int)top;
    }
    __device__ void replace(int n)                            // Frontier.replace: every match, in array order, re-sifted
    {
        const unsigned long long v = nodes[n];
        const int r = n_r(v), c = n_c(v);
        for (int i = 0; i < hs; i++) {
            const unsigned long long u = nodes[hn[i]];
            if (n_r(u) == r && n_c(u) == c) {
                hf[i] = f_of(n); hn[i] = (uint32_t)n;
                sift_down(0, i);
                inf[r * 82 + c] = n;
            }
        }
    }
};

// AstarSolver(start, [0,1,2,3], maze, goal).get_actions() by lane 0 of the wave; every lane gets the result: the plan's length
// (0: start == goal), or -1 when there is no path. plan: the env's action list (u8 [kNavPlanCap]).
__device__ int astar_np(const uint32_t *tile, unsigned char *scratch, uint32_t from, uint32_t goal, int lane, uint32_t *faults)
{
    unsigned char *plan = scratch;
    int32_t *inf = reinterpret_cast<int32_t *>(scratch + kNavOffInf);
    uint8_t *expl = scratch + kNavOffExp;
    for (int i = lane; i < kNavCells; i += 64) { inf[i] = -1; expl[i] = 0; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    wave_lds_sync();
    int result = -1;
    if (lane == 0) {
        AstarNp a;
        a.nodes = reinterpret_cast<unsigned long long *>(scratch + kNavOffNodes);
        a.hf = reinterpret_cast<double *>(scratch + kNavOffHf);
        a.hn = reinterpret_cast<uint32_t *>(scratch + kNavOffHn);
        a.inf = inf; a.expl = expl; a.nn = 0; a.hs = 0; a.overflow = false;
        a.gr = (int)(goal & 0xffu); a.gc = (int)((goal >> 8) & 0xffu);
        a.push(a.add_node((int)(from & 0xffu), (int)((from >> 8) & 0xffu), 0, 0, 0xffffffffu));
        int solution = -1;
        while (a.hs > 0 && !a.overflow) {
            const int n = a.pop();
            const unsigned long long v = a.nodes[n];
            const int r = AstarNp::n_r(v), c = AstarNp::n_c(v);
            if (r == a.gr && c == a.gc) { solution = n; break; }
            expl[r * 82 + c] = 1;
            const int cost = AstarNp::n_cost(v) + 1;
            for (int act = 0; act < 4; act++) {
                int cr = r + (act == 0 ? -1 : (act == 1 ? 1 : 0)), cc = c + (act == 2 ? -1 : (act == 3 ? 1 : 0));
                if (tile_bit(tile, cr, cc) != 0u) { cr = r; cc = c; }          // _next_state: a wall bump returns the same state
                const int k = cr * 82 + cc;
                const int q = inf[k];
                if (!expl[k] && q < 0) {
                    a.push(a.add_node(cr, cc, act, cost, (uint32_t)n));
                } else if (q >= 0 && AstarNp::n_cost(a.nodes[q]) < cost) {
                    a.replace(a.add_node(cr, cc, act, cost, (uint32_t)n));      // the reference's (inverted) test, :146-147
                }
            }
        }
        if (a.overflow) { atomicOr(faults, kFaultNavOverflow); solution = -1; }
        if (solution >= 0) {
            const int len = AstarNp::n_cost(a.nodes[solution]);       // unit step cost: the path's length
            if (len > kNavPlanCap) { atomicOr(faults, kFaultNavOverflow); }
            else {
                result = len;
                int i = len;
                for (uint32_t n = (uint32_t)solution; AstarNp::n_prev(a.nodes[n]) != 0xffffffffu; n = AstarNp::n_prev(a.nodes[n]))
                    plan[--i] = (unsigned char)AstarNp::n_act(a.nodes[n]);
            }
        }
    }
    return __builtin_amdgcn_readfirstlane(result);
}

// The planning loop shared by Navigator.reset and Navigator.step (navigator.py:22-38,46-62): A* to the goal; no path, or an
// empty one (the target stands on the goal) -> a fresh goal = sample_goal(1)[0] (a whole permutation of the free cells), up to
// six failures, then plan B = ten random actions. Leaves the plan in the env's scratch, its length, cursor 0 and the goal in
// the stream block's side words.
__device__ void nav_plan_np(NpStream &rs, uint16_t *perm, const uint32_t *tile, int side, unsigned char *scratch, uint32_t *side_words,
                            uint32_t from, uint32_t goal, int lane, uint32_t *faults, bool rpf = false, uint32_t vector = 0u)
{
    int count_res = 0, len;
    for (;;) {
        len = astar_np(tile, scratch, from, goal, lane, faults);
        if (len > 0) break;
        if (++count_res > 5) { len = -1; break; }
        if (rpf) { vector = (vector + 1u) & 3u; goal = rpf_cand(side, (int)vector); continue; }   // static sample_goal: the next patrol cell
        const FreeIndex fi = build_free_index(tile, side, lane);
        rs.permutation(fi.total, perm);                          // choice(len, size=1, replace=False)
        goal = select_free(tile, side, fi, uni((int)perm[0]), lane);
    }
    if (rpf && lane == 0) side_words[kNpRpfVector] = vector;
    if (len < 0) {                                               // plan B: np.random.choice(all_actions, 10)
        len = 10;
        for (int i = 0; i < 10; i++) {
            const uint32_t a = rs.bounded(3u);
            if (lane == 0) scratch[i] = (unsigned char)a;
        }
    }
    if (lane == 0) { side_words[kNpPlanLen] = (uint32_t)len; side_words[kNpPlanCur] = 0u; side_words[kNpNavGoal] = goal; }
}

// The generator pass of a numpy-stream handle: ONE wave per env, its consumed slots refilled in episode order (the stream is
// sequential: episode k + 1's draws follow episode k's).
constexpr int kNpWaves = 2;
// inter (handles whose Ram targets draw from the stream BETWEEN resets, RamAgent.step navigator.py:77-88): nothing may be
// generated ahead of time — the pass runs inside t2d_reset, BEFORE the reset launch, for exactly the envs that launch restarts
// (mask; null = all), makes the one episode they are about to start and ends it with RamAgent.reset's draws (navigator.py:90-93,
// after init_maze: track_1v1.py:134-144).
__global__ __launch_bounds__(64 * kNpWaves) void k_gen_np(DevState s, uint32_t lo, uint32_t hi, int force, const uint8_t *mask,
                                                          int inter)
{
    __shared__ __attribute__((aligned(16))) uint32_t tiles[kNpWaves][kTileWords];
    __shared__ uint32_t mts[kNpWaves][624];
    __shared__ uint16_t perms[kNpWaves][6400];
    const int lane = (int)(threadIdx.x & 63u);
    const int wave = uni((int)(threadIdx.x >> 6));
    const int e = (int)blockIdx.x * kNpWaves + wave;
    if (e >= s.n) return;
    const uint32_t cfg = s.cfg[e];
    const uint32_t cur = s.episode[e];
    const uint32_t req0 = s.gen_req[e], req1 = s.gen_req[(size_t)s.n + e];
    bool need0 = force || (req0 >= lo && req0 <= hi && req0 != 0u), need1 = force || (req1 >= lo && req1 <= hi && req1 != 0u);
    if (inter) {
        const bool mine = force && (mask == nullptr || mask[e] != 0);
        need0 = mine && ((cur + 1u) & 1u) == 0u; need1 = mine && ((cur + 1u) & 1u) == 1u;
    }
    if (!need0 && !need1) return;
    uint32_t *mt_g = s.np_mt + (size_t)e * kNpStateWords;
    const uint32_t tflag = inter ? uni(mt_g[kNpRamFlag]) : 0u;
    const bool ram = tflag == 1u, rpf = tflag == 3u, navig = tflag == 2u || rpf;
    NpStream rs;
    rs.mt = mts[wave]; rs.lane = lane;
    for (int i = lane; i < 624; i += 64) rs.mt[i] = mt_g[i];
    rs.pos = (int)mt_g[624];
    wave_lds_sync();
    // slot p holds the lowest episode number above the current one with parity p: the lower of the two first
    const int first = (int)((cur + 1u) & 1u);
#pragma unroll 1
    for (int t = 0; t < 2; t++) {
        const int slot = t == 0 ? first : 1 - first;
        if (!(slot == 0 ? need0 : need1)) continue;
        const size_t so = (size_t)slot * s.n + e;
        uint32_t pos, goals, d2;
        uint32_t rpf_saved = 0u, rpf_vec = 0u;
        generate_episode_np(rs, perms[wave], tiles[wave], lane, cfg, pos, goals, d2, rpf, &rpf_saved, &rpf_vec);
        const uint32_t plan0 = ram ? ram_reset(rs) : 0u;        // RamAgent.reset(): randint(1, 10), then choice(4, n)
        if (navig)      // Navigator.reset(init_states[1], goal_states[1], maze_generator) (track_1v1.py:139-141, navigator.py:43-63)
            nav_plan_np(rs, perms[wave], tiles[wave], side_of_cfg(cfg), s.np_nav + (size_t)e * kNavBytes, mt_g, pos >> 16, goals >> 16,
                        lane, s.faults, rpf, rpf_vec);
        if (rpf) rpf_restore(tiles[wave], side_of_cfg(cfg), rpf_saved, lane);     // the ENV's map keeps its walls on the patrol cells
        wave_lds_sync();
        store_slot_map(s, so, tiles[wave], lane, cfg, pos);
        if (lane == 0) {
            s.n_pos[so] = pos; s.n_goals[so] = goals; s.n_plan[so] = plan0; s.n_tctr[so] = 0u;
            s.n_navgoal[so] = goals >> 16; s.n_d2[so] = d2; s.gen_req[so] = 0u; s.n_nav2[so] = 0u;
        }
        wave_lds_sync();
    }
    for (int i = lane; i < 624; i += 64) mt_g[i] = rs.mt[i];
    if (lane == 0) mt_g[624] = (uint32_t)rs.pos;
}

// RamAgent.step() (navigator.py:77-88) for the Ram-target envs of a numpy-stream handle, BEFORE the step launch: the action
// the env step will take for the target (track_1v1.py:80-82 overrides action[1]) goes to out[e] in the caller's action dtype;
// every other env's entry is the caller's own. One wave per env: almost all of them read the plan word, emit the next planned
// action and leave; the env whose plan runs out on this step brings its MT19937 state into LDS and draws — the coin, on heads
// the action that OVERRIDES the one being returned and the run length, on tails a fresh length and plan — exactly the words
// the reference's global stream would have handed RamAgent at this point of the env's life.
// Navigator envs (flag 2; Navigator.step, navigator.py:11-41): the next action of the plan; a plan that is used up is replaced
// first — a fresh goal = sample_goal(1)[0] (a whole permutation of the free cells), then the planning loop from the target's
// current cell on the env's map (heap A*, retries, plan B: nav_plan_np).
__global__ __launch_bounds__(64 * kNpWaves) void k_ram_np(DevState s, const void *act_in, void *act_out, int adt)
{
    __shared__ uint32_t mts[kNpWaves][624];
    __shared__ __attribute__((aligned(16))) uint32_t tiles[kNpWaves][kTileWords];
    __shared__ uint16_t perms[kNpWaves][6400];
    const int lane = (int)(threadIdx.x & 63u);
    const int wave = uni((int)(threadIdx.x >> 6));
    const int e = (int)blockIdx.x * kNpWaves + wave;
    if (e >= s.n) return;
    uint32_t *mt_g = s.np_mt + (size_t)e * kNpStateWords;
    long long a = 0;
    const uint32_t tflag = uni(mt_g[kNpRamFlag]);
    if (tflag >= 2u) {
        const bool rpf = tflag == 3u;
        unsigned char *scratch = s.np_nav + (size_t)e * kNavBytes;
        uint32_t cur = uni(mt_g[kNpPlanCur]);
        if (cur >= uni(mt_g[kNpPlanLen])) {
            NpStream rs;
            rs.mt = mts[wave]; rs.lane = lane;
            for (int i = lane; i < 624; i += 64) rs.mt[i] = mt_g[i];
            rs.pos = (int)mt_g[624];
            reinterpret_cast<uint4 *>(tiles[wave])[lane] = reinterpret_cast<const uint4 *>(s.maps + (size_t)e * kTileWords)[lane];
            wave_lds_sync();
            const int side = (int)(uni(s.cnt[e]) >> 24);
            uint32_t goal, vec = 0u;
            if (rpf) {              // the patrol: plans are made on the generator's map, the next patrol cell is the goal (no draw)
                (void)rpf_clear(tiles[wave], side, lane);
                vec = (uni(mt_g[kNpRpfVector]) + 1u) & 3u;
                goal = rpf_cand(side, (int)vec);
            } else {
                const FreeIndex fi = build_free_index(tiles[wave], side, lane);
                rs.permutation(fi.total, perms[wave]);           // sample_goal(1): choice(len, size=1, replace=False)
                goal = select_free(tiles[wave], side, fi, uni((int)perms[wave][0]), lane);
            }
            nav_plan_np(rs, perms[wave], tiles[wave], side, scratch, mt_g, uni(s.pos[e]) >> 16, goal, lane, s.faults, rpf, vec);
            wave_lds_sync();
            for (int i = lane; i < 624; i += 64) mt_g[i] = rs.mt[i];
            if (lane == 0) mt_g[624] = (uint32_t)rs.pos;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            wave_lds_sync();
            cur = 0u;
        }
        a = (long long)scratch[cur];
        if (lane == 0) mt_g[kNpPlanCur] = cur + 1u;
    } else if (tflag == 0u) {
        if (act_in) a = adt == T2D_ACT_U8 ? (long long)reinterpret_cast<const uint8_t *>(act_in)[e]
                      : (adt == T2D_ACT_I32 ? (long long)reinterpret_cast<const int32_t *>(act_in)[e]
                                            : reinterpret_cast<const long long *>(act_in)[e]);
    } else {
        uint32_t plan = uni(s.plan[e]);
        if (plan_cur(plan) + 1u >= plan_len(plan)) {              // the plan ends with this action: RamAgent draws now
            NpStream rs;
            rs.mt = mts[wave]; rs.lane = lane;
            for (int i = lane; i < 624; i += 64) rs.mt[i] = mt_g[i];
            rs.pos = (int)mt_g[624];
            wave_lds_sync();
            a = (long long)ram_step(plan, rs);
            wave_lds_sync();
            for (int i = lane; i < 624; i += 64) mt_g[i] = rs.mt[i];
            if (lane == 0) mt_g[624] = (uint32_t)rs.pos;
        } else {                                                   // ram_step's other branch: the next planned action
            const uint32_t cur = plan_cur(plan);
            a = (long long)plan_act(plan, cur);
            plan = (plan & 0xf0ffffffu) | ((cur + 1u) << 24);
        }
        if (lane == 0) s.plan[e] = plan;
    }
    if (lane == 0) {
        if (adt == T2D_ACT_U8) reinterpret_cast<uint8_t *>(act_out)[e] = (uint8_t)a;
        else if (adt == T2D_ACT_I32) reinterpret_cast<int32_t *>(act_out)[e] = (int32_t)a;
        else reinterpret_cast<long long *>(act_out)[e] = a;
    }
}

// Grow the mazes the coming generator passes will ask for: for every Maze env the episodes current + 3 and current + 4 (the two
// pre-generated slots hold current + 1 and + 2; the pass that refills a consumed slot builds current + 3, then + 4). One wave
// per (which, env); almost all of them find their entry already there and leave. Runs beside anything: see DevState::g_maps.
__global__ __launch_bounds__(256) void k_pregrow(DevState s)
{
    __shared__ __attribute__((aligned(16))) uint32_t tiles[kWavesPerBlock][kTileWords];
    __shared__ uint32_t mlogs[kWavesPerBlock][kMazeLogMax];
    const int lane = (int)(threadIdx.x & 63u);
    const int wave = uni((int)(threadIdx.x >> 6));
    const int idx = (int)blockIdx.x * kWavesPerBlock + wave;
    if (idx >= 2 * s.n) return;
    const int which = idx >= s.n ? 1 : 0, e = idx - which * s.n;
    const uint32_t cfg = s.cfg[e];
    if ((int)(cfg & 3u) != MAP_MAZE) return;
    const uint32_t cur = __hip_atomic_load(s.episode + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t want = cur + 3u + (uint32_t)which;
    const size_t po = (size_t)(want & 3u) * s.n + e;
    const uint32_t tag = __hip_atomic_load(s.g_ep + po, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (an entry is re-used only when its episode has started — nobody will ask for it again — or when it holds nothing)
    if (tag == want || (tag != kPoolEmpty && tag > cur)) { if (lane == 0) atomicAdd(s.pg_stats + 3, 1u); return; }
    if (lane == 0) __hip_atomic_store(s.g_ep + po, kPoolEmpty, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t *tile = tiles[wave];
    grow_maze(s, s.env_base + (uint32_t)e, want, (int)((cfg >> 5) & 15u), tile, mlogs[wave], lane);
    wave_lds_sync();
    reinterpret_cast<uint4 *>(s.g_maps + po * kTileWords)[lane] = reinterpret_cast<const uint4 *>(tile)[lane];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");         // every lane's part of the tile before the tag
    wave_lds_sync();
    if (lane == 0) {
        __hip_atomic_store(s.g_ep + po, want, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        atomicAdd(s.pg_stats + 2, 1u);
    }
}

__global__ __launch_bounds__(256) void k_nav_prefetch(DevState s)
{
    __shared__ __attribute__((aligned(16))) uint32_t tiles[kWavesPerBlock][kTileWords];
    const int lane = (int)(threadIdx.x & 63u);
    const int wave = uni((int)(threadIdx.x >> 6));
    const int e = (int)blockIdx.x * kWavesPerBlock + wave;
    if (e >= s.n) return;
    if ((int)((s.cfg[e] >> 2) & 7u) != TGT_NAV) return;
    nav_prefetch(s, e, tiles[wave], lane);
}

// Parity hook: park a plan to a GIVEN goal in the prefetch slot of one Nav env (what k_nav_prefetch does with a goal it
// draws itself, minus the draw): the env's next step adopts it if the target stands on its current goal (t2d_inject leaves
// it there) and the goal is reachable.
__global__ __launch_bounds__(64) void k_nav_inject_goal(DevState s, int e, uint32_t goal)
{
    __shared__ __attribute__((aligned(16))) uint32_t tile[kTileWords];
    const int lane = (int)threadIdx.x;
    reinterpret_cast<uint4 *>(tile)[lane] = reinterpret_cast<const uint4 *>(s.maps + (size_t)e * kTileWords)[lane];
    wave_lds_sync();
    const int side = (int)(s.cnt[e] >> 24);
    NavField nf;
    const uint32_t cur = s.navgoal[e];
    bfs_dir_field(tile, side, lane, (int)(goal & 0xffu), (int)(goal >> 8), nf, false, (int)(cur & 0xffu), (int)(cur >> 8));
    const uint32_t episode = s.episode[e];
    const size_t q0 = pq_index(s, episode, 0u, e);
    store_plan_field(s.p_field + q0 * kPlanWords, nf, side, lane);
    if (lane == 0) { s.p_goal[q0] = goal; s.p_tctr[q0] = s.tctr[e]; s.p_state[pq_state_index(s, episode, e)] = 1u; }
}

// _get_obs / _get_partial_obs (track_1v1.py:287-326) for ONE env by ONE wave. Lanes 0..51 each expand one half
// row (7 + 6 cells) of the 2 x 13 crop rows from the LDS tile into an LDS staging row (bits -> bytes by a
// multiply spread, agent marks, v_cvt_f32_ubyteN); then all 64 lanes stream the 338 floats out as coalesced
// 8 B/lane stores (1352 B per env is 8-byte, not 16-byte, aligned).
__device__ __forceinline__ void emit_obs(const uint32_t *tile, float *stage, uint32_t pos, int side, int lane,
                                         float *gobs)
{
    if (lane < 52) {
        const int row = lane >> 1, h = lane & 1;
        const int ag = row >= T2D_WIN ? 1 : 0;
        const int y = row - ag * T2D_WIN;
        const int tr_r = (int)(pos & 0xffu), tr_c = (int)((pos >> 8) & 0xffu);
        const int tg_r = (int)((pos >> 16) & 0xffu), tg_c = (int)(pos >> 24);
        const int rr = (ag ? tg_r : tr_r) - T2D_POB + y;
        const int c0 = (ag ? tg_c : tr_c) - T2D_POB + h * 7;   // first column of this half row, in [-5, 81]
        uint32_t bits = 0x7fu;                                   // np.pad(..., 1): rows outside the map
        if ((unsigned)rr < (unsigned)side) {
            const uint32_t *w = tile + rr * kRowWords;
            const uint64_t lo = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
            const uint64_t hi = (uint64_t)w[2] | (~0ull << (side - 64));   // columns >= side read as 1
            uint64_t v;
            if (c0 < 0) v = (lo << (-c0)) | ((1ull << (-c0)) - 1ull);      // columns < 0 read as 1
            else if (c0 == 0) v = lo;
            else if (c0 < 64) v = (lo >> c0) | (hi << (64 - c0));
            else v = hi >> (c0 - 64);
            bits = (uint32_t)v & 0x7fu;
        }
        // bit k -> byte k (0/1)
        uint64_t bytes = ((uint64_t)bits * 0x0002040810204081ull) & 0x0101010101010101ull;
        if (rr == tr_r) { const int k = tr_c - c0; if (k >= 0 && k < 7) bytes = (bytes & ~(0xffull << (8 * k))) | (2ull << (8 * k)); }
        if (rr == tg_r) { const int k = tg_c - c0; if (k >= 0 && k < 7) bytes = (bytes & ~(0xffull << (8 * k))) | (4ull << (8 * k)); }
        if (y == T2D_POB && h == 0) bytes = (bytes & ~(0xffull << 48)) | ((ag ? 4ull : 2ull) << 48);  // own cell (:313)
        float *dst = stage + row * T2D_WIN + h * 7;
        const uint32_t b0 = (uint32_t)bytes, b1 = (uint32_t)(bytes >> 32);
        dst[0] = (float)(b0 & 0xffu); dst[1] = (float)((b0 >> 8) & 0xffu); dst[2] = (float)((b0 >> 16) & 0xffu);
        dst[3] = (float)(b0 >> 24); dst[4] = (float)(b1 & 0xffu); dst[5] = (float)((b1 >> 8) & 0xffu);
        if (h == 0) dst[6] = (float)((b1 >> 16) & 0xffu);
    }
    wave_lds_sync();
    const float2 *src = reinterpret_cast<const float2 *>(stage);
    float2 *out = reinterpret_cast<float2 *>(gobs);
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const int q = lane + 64 * i;
        if (q < kObsPerEnv / 2) out[q] = src[q];
    }
}

// _get_obs for obs_type 'Full' (track_1v1.py:288-290,295-307): both agents get the whole map, tracker cell = 2,
// target cell = 4 (painted last). One wave writes the env's two identical S x S planes, 256 B per store.
__device__ __forceinline__ void emit_full_obs(const uint32_t *tile, uint32_t pos, int side, int lane, float *gobs)
{
    const int tr = (int)(pos & 0xffu) * side + (int)((pos >> 8) & 0xffu);
    const int tg = (int)((pos >> 16) & 0xffu) * side + (int)(pos >> 24);
    const int cells = side * side;
    for (int idx = lane; idx < cells; idx += 64) {
        const int r = idx / side, c = idx - r * side;
        float v = (float)tile_bit(tile, r, c);
        if (idx == tr) v = 2.0f;
        if (idx == tg) v = 4.0f;
        gobs[idx] = v;
        gobs[cells + idx] = v;
    }
}

// MULTI (random-action rollouts only, no Nav targets): `nsteps` consecutive env steps per launch with the env's
// state held in registers / its map tile in LDS; step k writes its outputs to obs/rew/done + k * (per-step size) and is
// stamped stamp + k. nsteps <= gen_every, so at most one episode switch happens per env per launch (an episode lasts
// >= 11 steps) and the single pre-generated slot suffices. Results are identical to nsteps single-step launches.
// ADT: element type of the action tensors (compile-time, so that the action loads are straight-line code issued
// together with the state and tile loads); act1 is never null here (the host passes act0 again when the target is
// scripted — its value is then overridden by the Ram/Nav plan).
template <int OP, bool RANDOM, bool NAV, bool MULTI = false, int ADT = T2D_ACT_I64>
__global__ __launch_bounds__(256) void k_env(DevState s, const void *act0, const void *act1, int act_dtype,
                                             const uint8_t *mask, float *obs, float *rew, uint8_t *done_out,
                                             uint32_t aseed_lo, uint32_t aseed_hi, uint32_t step_idx, uint32_t stamp,
                                             int nsteps)
{
    static_assert(!MULTI || (OP == OP_STEP && RANDOM && !NAV), "MULTI is the fused random-action rollout");
    __shared__ __attribute__((aligned(16))) uint32_t tiles[kWavesPerBlock][kTileWords];
    __shared__ __attribute__((aligned(16))) float stages[kWavesPerBlock][kObsPerEnv + 2];

    const int lane = (int)(threadIdx.x & 63u);
    const int wave = uni((int)(threadIdx.x >> 6));
    const int e = (int)blockIdx.x * kWavesPerBlock + wave;
    if (e >= s.n) return;

    uint32_t *tile = tiles[wave];
    uint32_t *gtile = s.maps + (size_t)e * kTileWords;
    reinterpret_cast<uint4 *>(tile)[lane] = reinterpret_cast<const uint4 *>(gtile)[lane];
    uint32_t pos = s.pos[e], cnt = s.cnt[e];
    const uint32_t cfg = s.cfg[e];
    long long act_raw0 = 0, act_raw1 = 0;
    if (OP == OP_STEP && !RANDOM) {
        act_raw0 = load_action_raw<ADT>(act0, e);
        act_raw1 = load_action_raw<ADT>(act1, e);
    }
    uint32_t plan = 0, tctr = 0, navgoal = 0, d2 = 0, episode = 0;
    const int mode = (int)((cfg >> 2) & 7u);
    if (MULTI) { plan = s.plan[e]; tctr = s.tctr[e]; episode = s.episode[e]; }   // carried in registers across steps
    // Nav / RPF handles: the scripted target's state rides in the FIRST batch of loads (it was fetched inside the mode
    // branch, one dependent round trip later), so that only the direction word — whose address needs the position —
    // is left for a second one
    uint32_t nv_plan = 0, nv_tctr = 0, nv_goal = 0, nv_nav2 = 0, nv_episode = 0, nv_pstate = 0;
    if (NAV && OP == OP_STEP) {
        nv_plan = s.plan[e]; nv_tctr = s.tctr[e]; nv_goal = s.navgoal[e]; nv_nav2 = s.nav2[e]; nv_episode = s.episode[e];
        nv_pstate = s.p_state[pq_state_index(s, nv_episode, e)];
    }
    wave_lds_sync();

  for (int k = 0; k < (MULTI ? nsteps : 1); k++) {
    bool consume = false, dirty = false, navgoal_dirty = false;
    if (OP == OP_RESET) consume = (mask == nullptr) || (mask[e] != 0);

    if (OP == OP_STEP) {
        const uint32_t genv = s.env_base + (uint32_t)e;
        const int side = (int)(cnt >> 24);
        int c_far = (int)(cnt & 0xffu), t = (int)((cnt >> 8) & 0xffffu);
        int a_tr, a_tg;
        if (RANDOM) {
            u32x4 w = philox4x32_10(aseed_lo, aseed_hi, step_idx + (uint32_t)k, 0u, genv, STREAM_ACTION);
            a_tr = (int)(w.x & (uint32_t)s.amask); a_tg = (int)(w.y & (uint32_t)s.amask);
        } else {
            a_tr = check_action(act_raw0, s.amask, s.faults);
            a_tg = check_action(act_raw1, s.amask, s.faults);
        }
        if (mode == TGT_RAM) { // track_1v1.py:81-82
            if (!MULTI) { plan = s.plan[e]; tctr = s.tctr[e]; episode = s.episode[e]; }
            Stream ts;
            ts.init(s.k0, s.k1, episode, genv, STREAM_TARGET, tctr);
            a_tg = (int)ram_step(plan, ts);
            tctr = ts.ctr;
            dirty = true;
        }
        int r0 = (int)(pos & 0xffu), c0 = (int)((pos >> 8) & 0xffu);
        int r1 = (int)((pos >> 16) & 0xffu), c1 = (int)(pos >> 24);
        if (NAV && (mode == TGT_NAV || mode == TGT_RPF)) { // track_1v1.py:83-84 -> Navigator.step(old_state[1], ...) (navigator.py:11-41)
            plan = nv_plan; tctr = nv_tctr; navgoal = nv_goal;
            const bool rpf = mode == TGT_RPF;
            uint32_t nav2 = rpf ? nv_nav2 : 0u;
            uint32_t *gdir = s.dirf + (size_t)e * kDirWords;
            Stream ts;
            ts.init(s.k0, s.k1, nv_episode, genv, STREAM_TARGET, tctr);
            bool planb = ((plan >> 28) & 1u) != 0u;
            // RPF: the plan is an open-loop action list made on the generator's map (the env may hold walls on the
            // patrol cells): follow the field from a virtual position for exactly the planned number of steps
            const bool exhausted = planb ? (plan_cur(plan) >= plan_len(plan))
                                   : rpf ? (((nav2 >> 16) & 0x3fffu) == 0u)
                                         : (r1 == (int)(navgoal & 0xffu) && c1 == (int)(navgoal >> 8));
            int qr = rpf ? (int)(nav2 & 0xffu) : r1, qc = rpf ? (int)((nav2 >> 8) & 0xffu) : c1;
            const uint32_t dir_here = load_dir(gdir, qr, qc);      // issued now, used if the current plan still stands
            uint32_t dir = 0;
            bool adopted = false, have_goal = false;
            if (exhausted && !rpf && pq_count(nv_pstate) != 0u) {
                // a plan for the next goal was prepared by the generator pass (k_gen): adopt it if it is a valid plan
                // from here (reachable, not already on the goal) — else fall through to the inline re-plan, which
                // then starts from the same already-drawn goal (and voids whatever was prepared beyond it)
                const size_t hs = pq_index(s, nv_episode, pq_head(nv_pstate), e);
                const uint32_t *pf = s.p_field + hs * kPlanWords;
                const uint32_t g2 = s.p_goal[hs];
                ts.init(s.k0, s.k1, nv_episode, genv, STREAM_TARGET, s.p_tctr[hs]);
                navgoal = g2;
                have_goal = true;
                uint32_t new_ps = 0u;
                if (load_vis(pf, r1, c1) != 0u && !(r1 == (int)(g2 & 0xffu) && c1 == (int)(g2 >> 8))) {
                    const uint4 *src = reinterpret_cast<const uint4 *>(pf);
                    uint4 *dst = reinterpret_cast<uint4 *>(gdir);
                    dst[lane] = src[lane]; dst[lane + 64] = src[lane + 64];
                    dir = load_dir(pf, r1, c1);
                    plan = 0u; planb = false;
                    adopted = true;
                    navgoal_dirty = true;
                    new_ps = pq_pop(nv_pstate);
                }
                if (lane == 0) s.p_state[pq_state_index(s, nv_episode, e)] = new_ps;
            }
            if (exhausted && !adopted) {
                const FreeIndex fi = build_free_index(tile, side, lane);
                if (rpf) { nav2 = (nav2 & 0x3fffffffu) | ((((nav2 >> 30) + 1u) & 3u) << 30); navgoal = rpf_cell(side, (int)(nav2 >> 30)); }
                else if (!have_goal) navgoal = select_free(tile, side, fi, (int)ts.bounded((uint32_t)(fi.total - 1)), lane);
                NavField nf;
                nav_plan(tile, side, lane, r1, c1, fi, navgoal, ts, plan, nf, rpf, nav2);
                planb = ((plan >> 28) & 1u) != 0u;
                qr = r1; qc = c1;
                if (!planb) { store_dir_field(gdir, nf, side, lane); dir = nav_dir_from_regs(nf, qr, qc); }
                navgoal_dirty = true;
            } else if (!planb && !adopted) {
                dir = dir_here;
            }
            if (rpf && !planb) {   // advance the virtual position along the field, one planned step consumed
                qr += dir == 0u ? -1 : (dir == 1u ? 1 : 0); qc += dir == 2u ? -1 : (dir == 3u ? 1 : 0);
                const uint32_t rem = ((nav2 >> 16) & 0x3fffu) - 1u;
                nav2 = (uint32_t)qr | ((uint32_t)qc << 8) | ((rem & 0x3fffu) << 16) | (nav2 & 0xc0000000u);
            }
            if (rpf && lane == 0) s.nav2[e] = nav2;
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
            int nr = r0 + move_dy(a_tr), nc = c0 + move_dx(a_tr);
            if (tile_bit(tile, nr, nc) == 0u) { r0 = nr; c0 = nc; }
            nr = r1 + move_dy(a_tg); nc = c1 + move_dx(a_tg);
            if (tile_bit(tile, nr, nc) == 0u) { r1 = nr; c1 = nc; }
        }
        pos = (uint32_t)r0 | ((uint32_t)c0 << 8) | ((uint32_t)r1 << 16) | ((uint32_t)c1 << 24);
        const int dr = r1 - r0, dc = c1 - c0;
        d2 = (uint32_t)(dr * dr + dc * dc);
        // w_p = 1 (PZR), -0.5 (Far), else 0 (track_1v1.py:147-152) selects the table
        const float2 rwd = s.rew_lut[(mode == TGT_PZR ? kLutN : (mode == TGT_FAR ? 2 * kLutN : 0)) + (int)d2];
        c_far = d2 <= 36u ? 0 : min(c_far + 1, 255);  // distance <= 6 (track_1v1.py:106-109)
        int dn = c_far > 10;
        t = min(t + 1, 65535);
        if (s.max_steps > 0 && t >= s.max_steps) dn = 1; // gym TimeLimit
        cnt = (uint32_t)c_far | ((uint32_t)t << 8) | ((uint32_t)side << 24);
        if (lane == 0) {
            reinterpret_cast<float2 *>(rew)[(size_t)k * s.n + e] = rwd;
            done_out[(size_t)k * s.n + e] = (uint8_t)dn;
        }
        consume = dn && s.auto_reset;
    }

    if (OP != OP_OBSERVE && consume) {
        // Track1v1Env.reset(): switch to the pre-generated next episode (k_gen) — one 1 KiB tile copy + scalars
        if (!MULTI) episode = s.episode[e];
        const size_t so = (size_t)((episode + 1u) & 1u) * s.n + e;      // the slot that holds episode + 1
        const uint4 nt = reinterpret_cast<const uint4 *>(s.n_maps + so * kTileWords)[lane];
        reinterpret_cast<uint4 *>(tile)[lane] = nt;
        reinterpret_cast<uint4 *>(gtile)[lane] = nt;
        if (NAV && (mode == TGT_NAV || mode == TGT_RPF)) {
            const uint4 *src = reinterpret_cast<const uint4 *>(s.n_dirf + so * kDirWords);
            uint4 *dst = reinterpret_cast<uint4 *>(s.dirf + (size_t)e * kDirWords);
            dst[lane] = src[lane]; dst[lane + 64] = src[lane + 64];
        }
        if (OP == OP_STEP && s.np_mt && lane == 0) s.np_mt[(size_t)e * kNpStateWords + kNpTermD2] = d2;   // the finished step's distance
        pos = s.n_pos[so]; plan = s.n_plan[so]; tctr = s.n_tctr[so]; navgoal = s.n_navgoal[so]; d2 = s.n_d2[so];
        if (NAV && lane == 0) { s.nav2[e] = s.n_nav2[so]; s.p_state[pq_state_index(s, episode, e)] = 0u; }   // the finished episode's queue is void
        cnt = (uint32_t)side_of_cfg(cfg) << 24;
        episode += 1u;
        if (lane == 0) {
            s.goals[e] = s.n_goals[so]; s.episode[e] = episode; s.navgoal[e] = navgoal;
            s.gen_req[so] = stamp + (uint32_t)k;
        }
        wave_lds_sync();
    }
    if (lane == 0 && OP != OP_OBSERVE) {
        if (OP == OP_STEP || consume) { s.pos[e] = pos; s.cnt[e] = cnt; s.d2[e] = d2; }
        if (consume || dirty) { s.plan[e] = plan; s.tctr[e] = tctr; }
        if (navgoal_dirty && !consume) s.navgoal[e] = navgoal;
    }
    if (obs != nullptr) {
        if (s.obs_full)
            emit_full_obs(tile, pos, (int)(cnt >> 24), lane, obs + ((size_t)k * s.n + e) * 2 * s.obs_side * s.obs_side);
        else
            emit_obs(tile, stages[wave], pos, (int)(cnt >> 24), lane, obs + ((size_t)k * s.n + e) * kObsPerEnv);
    }
  }
}

// =====================================================================================================
// k_step2 — the step + observe kernel of every handle without Nav/RPF targets and with 'Partial' observations
// (all BASELINE configs but the Nav one). Same semantics as k_env<OP_STEP> (track_1v1.py:71-127,271-326), laid out
// for the two limits k_env hit: VALU issue at large N (~450 instructions per env-wave) and latency at N = 4096.
//
//   * TWO envs per wavefront: lane = slot(1) | agent(1) | k(4). An env pair's observations are 2 x 1352 B = 169 x
//     16 B, so every lane streams them out as three `global_store_dwordx4` (8-B stores run at 0.54-0.70x the 16-B
//     rate on this chip).
//   * Only the map rows the step can touch are fetched: lane k < 15 of an agent's group loads row r_old - 7 + k (12 B,
//     one dwordx3 load; rows outside the map read as walls). Those 15 rows contain the four move targets AND the 13
//     window rows of whichever cell the agent ends on, so the row loads depend on nothing but the state load: 360 B
//     per env instead of the 1 KiB tile, and no second dependent round trip for the wall test.
//   * The reward-table entries of the four possible outcomes (each agent moves or bumps) are fetched speculatively
//     together with the rows; the wall test is one wave ballot.
//   * Observation: the lane that HOLDS a row turns it into the 13 window bits (v_alignbit over the row words with
//     ones outside the map) and parks them as one dword in LDS; agent colours go into a zeroed 676-byte mark plane.
//     Each lane then builds its three 4-cell output groups straight from two row dwords and one mark dword
//     (bit -> byte spread by one 24-bit multiply, v_cvt_f32_ubyteN): no per-cell staging pass.
//   * The episode switch (rare) is a wave-uniform side path: whole-tile copy by all 64 lanes, then the rows are
//     re-read for the first observation of the next episode.
// OBS: 0 = f32 observations, 16-B stores (pointer and per-step stride 16-B aligned); 1 = f32, 4-B stores (any
// alignment); 2 = u8 observations [N,2,13,13] (the B_step = 709 variant of SURVEY 8d; the policy stem decodes them).
enum : int { OBS_F32_VEC4 = 0, OBS_F32_SCALAR = 1, OBS_U8 = 2 };
constexpr int kStage2Rows = 56;                     // 52 window rows of the pair (+1 read past the end, +pad)
constexpr int kStage2Words = kStage2Rows + 192;     // + 169 dwords of mark bytes (padded: every lane reads 3 of them)
#ifndef T2D_STEP2_WAVES
#define T2D_STEP2_WAVES 4
#endif
constexpr int kStep2Waves = T2D_STEP2_WAVES;        // waves (env pairs) per workgroup of k_step2

// One env PAIR's step + observe by one wavefront, split into three phases so that a caller can put independent work
// between them (k_act_step runs the policy's LSTM cells and draws between `rows` and `finish`, under the row loads):
//   init    lane roles; issues the state loads (pos, cnt, cfg [, plan, tctr, episode])
//   rows    needs the state: issues the 15 map-row loads of each agent and, for envs that can finish this step, the
//           speculative next-episode loads
//   finish  needs the two actions: scripted-target override, wall vote, reward, far counter / time limit, episode switch,
//           state write-back, observation
// k_step2 calls them back to back (per step of a MULTI launch: rows + finish).
// NAV: some env of the handle has the scripted Nav target (navigator.py:5-70): its action is one 2-bit read of the env's BFS
// direction planes at the target's cell; when the target stands on its goal the plan prepared by the generator pass is
// adopted (a 2 KiB copy whose source was fetched speculatively beside the map rows) or, rarely, re-made inline by the whole
// wave on an LDS tile (`tile`: 1 KiB per wave, NAV handles only).
template <bool MULTI, int OBS, bool RAM, bool NAV = false>
struct Step2 {
    static_assert(!(NAV && MULTI), "a re-plan rewrites the direction field other lanes re-read: one step per launch");
    int lane, e0, e, sl, ag, k;
    bool live, leader;
    uint32_t *st, *mk;
    uint32_t pos, cnt, cfg;
    int mode;
    bool ram;
    uint32_t plan, tctr, episode, d2, genv;
    const float2 *lut;
    const uint32_t *gmap;
    int cells;
    // rows phase
    int rowbase;
    uint32_t w0, w1, w2;
    bool maybe;
    unsigned long long mb;
    uint32_t sp_pos, sp_plan, sp_tctr, sp_d2, sp_goals, sp_navgoal, sp_episode, sp_win;
    uint4 sp_tile0, sp_tile1;
    // Nav target
    bool nav, nv_exh;
    uint32_t navgoal, pstate, nv_dA, nv_dB;                   // direction words at the target's cell (current plan)
    uint32_t sp_pgoal, sp_ptctr, sp_vis, sp_pdA, sp_pdB;      // prepared plan: goal, stream position, words at the target's cell
    uint4 sp_fld0, sp_fld1, sp_fld2, sp_fld3;                 // ... its direction planes (this lane's 64 B of the 2 KiB)
    uint4 sp_nd0, sp_nd1, sp_nd2, sp_nd3;                     // next episode's direction planes (episode switch)
    uint32_t *tile;
#if T2D_EXP == 6
    uint32_t *tstamp;   // timeline probe: s_memtime stamps of this wave, parked in the spare words 246..253 of the env's tile
#define T2D_STAMP(i) do { if (leader) tstamp[i] = (uint32_t)__builtin_readcyclecounter(); } while (0)
#else
#define T2D_STAMP(i) do { } while (0)
#endif

    __device__ __forceinline__ void init(const DevState &s, int e0_, int lane_, uint32_t *stage)
    {
        lane = lane_; e0 = e0_;
        sl = lane >> 5; ag = (lane >> 4) & 1; k = lane & 15;
        live = e0 + sl < s.n;                   // false only for slot 1 of the last pair of an odd batch
        e = live ? e0 + sl : e0;                // a dead slot shadows slot 0 (loads only; it never votes or stores)
        leader = live && ag == 0 && k == 15;    // one lane per env: reward table, rew/done, state write-back
        st = stage;
        mk = st + kStage2Rows;
        tile = nullptr;
#if T2D_EXP == 6
        tstamp = s.maps + (size_t)e * kTileWords + 246;
#endif
        T2D_STAMP(0);
        pos = s.pos[e]; cnt = s.cnt[e];
        cfg = s.cfg[e];
        plan = 0; tctr = 0; episode = 0; d2 = 0;
    }
    // second half of init, after the caller has issued ITS first loads (k_step2: the action loads): everything here
    // only consumes cfg / kernel arguments
    __device__ __forceinline__ void init2(const DevState &s, const void *obs, const float *rew, const uint8_t *done_out)
    {
        {   // every kernel argument the step needs, fetched NOW (one scalar round trip under the state loads' latency)
            // instead of lazily at first use, where each would put its own s_load + wait on the critical path
            const float2 *a0_ = s.rew_lut; const uint32_t *a1_ = s.maps, *a2_ = s.d2, *a3_ = s.faults;
            const int a4_ = s.max_steps, a5_ = s.auto_reset; const uint32_t a6_ = s.env_base;
            asm volatile("" ::"s"(a0_), "s"(a1_), "s"(a2_), "s"(a3_), "s"(a4_), "s"(a5_), "s"(a6_), "s"(obs), "s"(rew),
                         "s"(done_out));
        }
        mode = (int)((cfg >> 2) & 7u);
        // RAM = some env of the handle has the scripted Ram target; handles without one get a kernel without the plan /
        // Philox code (the step kernel's duration at N = 4096 is partly instruction-fetch latency: code size matters)
        ram = RAM && mode == TGT_RAM;
        nav = NAV && mode == TGT_NAV;
        navgoal = 0; pstate = 0;
        episode = s.episode[e];       // (also names the next-episode slot: parity of episode + 1)
        if (MULTI || ram || nav) { plan = s.plan[e]; tctr = s.tctr[e]; }
        if (nav) {      // (the queue state of all three sets: the right one is picked by episode % 3 without a dependent load)
            navgoal = s.navgoal[e];
            const uint32_t ps0 = s.p_state[e], ps1 = s.p_state[(size_t)s.n + e], ps2 = s.p_state[(size_t)2 * s.n + e];
            const uint32_t m3 = episode % 3u;
            pstate = m3 == 0u ? ps0 : (m3 == 1u ? ps1 : ps2);
        }
        genv = s.env_base + (uint32_t)e;
        lut = s.rew_lut + (mode == TGT_PZR ? kLutN : (mode == TGT_FAR ? 2 * kLutN : 0));
        gmap = s.maps + (size_t)e * kTileWords;
        cells = (e0 + 1 < s.n) ? 2 * kObsPerEnv : kObsPerEnv;
    }

    __device__ __forceinline__ void rows(const DevState &s)
    {
        const int side = (int)(cnt >> 24);
        const int c_far = (int)(cnt & 0xffu), t = (int)((cnt >> 8) & 0xffffu);
        const int r0 = (int)(pos & 0xffu), r1 = (int)((pos >> 16) & 0xffu);
        if (T2D_EXP == 6) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); T2D_STAMP(1); }
        // rows r_old - 7 .. r_old + 7 of this lane's agent: the move targets and every possible window row
        rowbase = (ag ? r1 : r0) - 7;
        w0 = 0xffffffffu; w1 = 0xffffffffu; w2 = 0xffffffffu;   // np.pad(..., 1): rows outside the map
        if (k < 15 && (unsigned)(rowbase + k) < (unsigned)side) {
            const uint32_t *rp = gmap + (rowbase + k) * kRowWords;
            w0 = rp[0]; w1 = rp[1]; w2 = rp[2];
        }
        // A done is only possible this step if the far counter stands at 10 or the time limit is one step away (:106-111,
        // TimeLimit): for those envs the next episode's scalars, the window rows of its first observation (n_win) and its map
        // tile are fetched NOW, beside the rows, so that the episode switch below adds no dependent memory round trip.
        nv_exh = false; nv_dA = 0; nv_dB = 0;
        sp_pgoal = 0; sp_ptctr = 0; sp_vis = 0; sp_pdA = 0; sp_pdB = 0;
        sp_fld0 = make_uint4(0u, 0u, 0u, 0u); sp_fld1 = sp_fld0; sp_fld2 = sp_fld0; sp_fld3 = sp_fld0;
        sp_nd0 = sp_fld0; sp_nd1 = sp_fld0; sp_nd2 = sp_fld0; sp_nd3 = sp_fld0;
        if (NAV && nav) {   // Navigator.step (navigator.py:11-41) on old_state[1] (track_1v1.py:84)
            const int c1 = (int)(pos >> 24);
            const bool planb = ((plan >> 28) & 1u) != 0u;
            nv_exh = planb ? (plan_cur(plan) >= plan_len(plan))
                           : (r1 == (int)(navgoal & 0xffu) && c1 == (int)(navgoal >> 8));
            const uint32_t *gdir = s.dirf + (size_t)e * kDirWords;
            const int w = r1 * kRowWords + (c1 >> 5);
            nv_dA = gdir[w]; nv_dB = gdir[256 + w];         // used if the current plan still stands
            if (nv_exh && pq_count(pstate) != 0u) {
                // the plan the generator pass prepared for the NEXT goal (the head of the env's plan queue), fetched now
                // (speculatively: it is adopted below if it reaches the target's cell) so that adopting it adds no dependent
                // round trip to the step
                const size_t hs = pq_index(s, episode, pq_head(pstate), e);
                const uint32_t *pf = s.p_field + hs * kPlanWords;
                sp_pgoal = s.p_goal[hs]; sp_ptctr = s.p_tctr[hs];
                sp_vis = pf[512 + w]; sp_pdA = pf[w]; sp_pdB = pf[256 + w];
                const uint4 *src = reinterpret_cast<const uint4 *>(pf) + (lane & 31);
                sp_fld0 = src[0]; sp_fld1 = src[32]; sp_fld2 = src[64]; sp_fld3 = src[96];
            }
        }
        maybe = live && s.auto_reset != 0 && (c_far >= 10 || (s.max_steps > 0 && t + 1 >= s.max_steps));
        mb = __ballot(maybe);
        sp_pos = 0; sp_plan = 0; sp_tctr = 0; sp_d2 = 0; sp_goals = 0; sp_navgoal = 0; sp_episode = 0; sp_win = 0x1fffu;
        sp_tile0 = make_uint4(0u, 0u, 0u, 0u); sp_tile1 = sp_tile0;
        if (__builtin_expect(mb != 0ull, 0)) {
            const uint32_t slot = (episode + 1u) & 1u;               // where this env's next episode sits
            const size_t so = (size_t)slot * s.n + e;
            if (maybe) {
                sp_pos = s.n_pos[so]; sp_plan = s.n_plan[so]; sp_tctr = s.n_tctr[so]; sp_d2 = s.n_d2[so];
                sp_goals = s.n_goals[so]; sp_navgoal = s.n_navgoal[so]; sp_episode = episode;
                if (k >= 1 && k <= T2D_WIN) sp_win = s.n_win[so * 32 + ag * T2D_WIN + (k - 1)];
                if (NAV && nav) {
                    const uint4 *src = reinterpret_cast<const uint4 *>(s.n_dirf + so * kDirWords) + (lane & 31);
                    sp_nd0 = src[0]; sp_nd1 = src[32]; sp_nd2 = src[64]; sp_nd3 = src[96];
                }
            }
            const size_t so0 = (size_t)__builtin_amdgcn_readlane(slot, 0) * s.n + e0;
            const size_t so1 = (size_t)__builtin_amdgcn_readlane(slot, 32) * s.n + e0 + 1;
            if ((mb & 0xffffffffull) != 0ull)
                sp_tile0 = reinterpret_cast<const uint4 *>(s.n_maps + so0 * kTileWords)[lane];
            if ((mb >> 32) != 0ull)
                sp_tile1 = reinterpret_cast<const uint4 *>(s.n_maps + so1 * kTileWords)[lane];
        }
    }

    // a_tr / a_tg: the two actions of this lane's env, already masked to the action table
    // returns the done flag of this lane's env (every lane of a slot computes it from the slot's replicated state)
    __device__ __forceinline__ int finish(const DevState &s, int a_tr, int a_tg, int it, void *obs, float *rew,
                                          uint8_t *done_out, uint32_t stamp)
    {
        const int side = (int)(cnt >> 24);
        int c_far = (int)(cnt & 0xffu), t = (int)((cnt >> 8) & 0xffffu);
        bool dirty = false;
        if (ram) { // track_1v1.py:81-82
            Stream ts;
            ts.init(s.k0, s.k1, episode, genv, STREAM_TARGET, tctr);
            a_tg = (int)ram_step(plan, ts);
            tctr = ts.ctr;
            dirty = true;
        }
        int r0 = (int)(pos & 0xffu), c0 = (int)((pos >> 8) & 0xffu);
        int r1 = (int)((pos >> 16) & 0xffu), c1 = (int)(pos >> 24);
        bool navgoal_dirty = false;
        if (NAV) {
            bool planb = ((plan >> 28) & 1u) != 0u, adopted = false, have_goal = false;
            uint32_t dir = ((nv_dA >> (c1 & 31)) & 1u) | (((nv_dB >> (c1 & 31)) & 1u) << 1);
            uint32_t *gdir = s.dirf + (size_t)e * kDirWords;
            if (nav && nv_exh && pq_count(pstate) != 0u) {
                // adopt the prepared plan if it is a valid plan from here (reachable, not already on its goal) — else the
                // inline re-plan below starts from the same already-drawn goal (k_env<NAV> does exactly this) and whatever
                // was prepared beyond it is void
                navgoal = sp_pgoal; tctr = sp_ptctr; have_goal = true;
                uint32_t new_ps = 0u;
                if (((sp_vis >> (c1 & 31)) & 1u) != 0u && !(r1 == (int)(sp_pgoal & 0xffu) && c1 == (int)(sp_pgoal >> 8))) {
                    uint4 *dst = reinterpret_cast<uint4 *>(gdir) + (lane & 31);
                    dst[0] = sp_fld0; dst[32] = sp_fld1; dst[64] = sp_fld2; dst[96] = sp_fld3;
                    dir = ((sp_pdA >> (c1 & 31)) & 1u) | (((sp_pdB >> (c1 & 31)) & 1u) << 1);
                    plan = 0u; planb = false; adopted = true;
                    navgoal_dirty = true;
                    new_ps = pq_pop(pstate);
                }
                if (leader) s.p_state[pq_state_index(s, episode, e)] = new_ps;
            }
            // rare: plan exhausted and nothing adoptable — Navigator's re-plan (navigator.py:15-38) by the whole wave, one slot
            // at a time, on the env's tile in LDS (wave-uniform copies of the slot's scalars in, results back to its lanes)
            const unsigned long long rm = __ballot(live && nav && nv_exh && !adopted);
            if (__builtin_expect(rm != 0ull, 0)) {
#ifdef T2D_COUNT_REPLANS
                {   // probe build: inline re-plans, split by the episode's age (bits 8-19: <= 20 steps old, 20-31: older)
                    const int age = (int)((cnt >> 8) & 0xffffu);
                    const unsigned long long young = __ballot(live && nav && nv_exh && !adopted && age <= 20);
                    if (lane == 0) atomicAdd(s.faults, ((uint32_t)__popcll(young & 0x100000001ull) << 8) +
                                                           ((uint32_t)__popcll(rm & ~young & 0x100000001ull) << 20));
                }
#endif
#pragma unroll 1
                for (int slot = 0; slot < 2; slot++) {
                    if (((rm >> (32 * slot)) & 1ull) == 0ull) continue;
                    const int src = 32 * slot;
                    const int es = e0 + slot;
                    const uint32_t u_pos = __builtin_amdgcn_readlane(pos, src), u_cnt = __builtin_amdgcn_readlane(cnt, src);
                    uint32_t u_goal = __builtin_amdgcn_readlane(navgoal, src), u_plan = __builtin_amdgcn_readlane(plan, src);
                    const uint32_t u_tctr = __builtin_amdgcn_readlane(tctr, src), u_ep = __builtin_amdgcn_readlane(episode, src);
                    const bool u_have = __builtin_amdgcn_readlane((uint32_t)have_goal, src) != 0u;
                    const int u_side = (int)(u_cnt >> 24), ur = (int)((u_pos >> 16) & 0xffu), uc = (int)(u_pos >> 24);
                    reinterpret_cast<uint4 *>(tile)[lane] = reinterpret_cast<const uint4 *>(s.maps + (size_t)es * kTileWords)[lane];
                    wave_lds_sync();
                    Stream ts;
                    ts.init(s.k0, s.k1, u_ep, s.env_base + (uint32_t)es, STREAM_TARGET, u_tctr);
                    const FreeIndex fi = build_free_index(tile, u_side, lane);
                    if (!u_have) u_goal = select_free(tile, u_side, fi, (int)ts.bounded((uint32_t)(fi.total - 1)), lane);
                    NavField nf;
                    uint32_t nav2_unused = 0u;
                    nav_plan(tile, u_side, lane, ur, uc, fi, u_goal, ts, u_plan, nf, false, nav2_unused);
                    const bool u_planb = ((u_plan >> 28) & 1u) != 0u;
                    uint32_t u_dir = 0u;
                    if (!u_planb) {
                        store_dir_field(s.dirf + (size_t)es * kDirWords, nf, u_side, lane);
                        u_dir = nav_dir_from_regs(nf, ur, uc);
                    }
                    wave_lds_sync();
                    if (sl == slot) {
                        navgoal = u_goal; plan = u_plan; tctr = ts.ctr; dir = u_dir; planb = u_planb;
                        navgoal_dirty = true;
                    }
                }
            }
            if (nav) {
                if (planb) {
                    const uint32_t cur = plan_cur(plan);
                    a_tg = (int)plan_act(plan, cur);
                    plan = (plan & 0xf0ffffffu) | ((cur + 1u) << 24);
                } else {
                    a_tg = (int)dir;
                }
                dirty = true;
            }
        }
        const int dy0 = move_dy(a_tr), dx0 = move_dx(a_tr);
        const int dy1 = move_dy(a_tg), dx1 = move_dx(a_tg);
        // the reward of each possible outcome (tracker moves / bumps) x (target moves / bumps): track_1v1.py:94-104
        float2 rw_mm = make_float2(0.f, 0.f), rw_ms = rw_mm, rw_sm = rw_mm, rw_ss = rw_mm;
        const int ddr = r1 - r0, ddc = c1 - c0;
        if (leader) {
            auto sq = [](int a, int b) { return a * a + b * b; };
            rw_mm = lut[sq(ddr + dy1 - dy0, ddc + dx1 - dx0)];
            rw_ms = lut[sq(ddr - dy0, ddc - dx0)];
            rw_sm = lut[sq(ddr + dy1, ddc + dx1)];
            rw_ss = lut[sq(ddr, ddc)];
        }
        // _next_state (track_1v1.py:271-285): the lane holding the destination row of its agent votes "wall"
        const int mdy = ag ? dy1 : dy0, mcc = (ag ? c1 : c0) + (ag ? dx1 : dx0);
        // (prvalues: `c ? w0 : w1` on members is an lvalue conditional, i.e. a load through a selected ADDRESS into the
        // struct, which keeps the whole struct in scratch memory)
        const uint32_t W0 = w0, W1 = w1, W2 = w2;
        const uint32_t wsel = (mcc >> 5) == 0 ? +W0 : ((mcc >> 5) == 1 ? +W1 : +W2);
        const bool vote = live && k == 7 + mdy && ((wsel >> (mcc & 31)) & 1u) != 0u;
        const unsigned long long bal = __ballot(vote);
        const uint32_t half = sl ? (uint32_t)(bal >> 32) : (uint32_t)bal;
        const bool wall_tr = (half & 0xffffu) != 0u, wall_tg = (half >> 16) != 0u;
        if (T2D_EXP == 6) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); T2D_STAMP(2); }
        if (T2D_EXP == 8) {   // latency probe: level-1 loads -> rows + reward table -> one store
            if (leader) reinterpret_cast<float2 *>(rew)[e] = wall_tr ? (wall_tg ? rw_ss : rw_sm) : (wall_tg ? rw_ms : rw_mm);
            return 0;
        }
        if (!wall_tr) { r0 += dy0; c0 += dx0; }
        if (!wall_tg) { r1 += dy1; c1 += dx1; }
        pos = (uint32_t)r0 | ((uint32_t)c0 << 8) | ((uint32_t)r1 << 16) | ((uint32_t)c1 << 24);
        {
            const int dr = r1 - r0, dc = c1 - c0;
            d2 = (uint32_t)(dr * dr + dc * dc);
        }
        c_far = d2 <= 36u ? 0 : min(c_far + 1, 255);  // distance <= 6 (track_1v1.py:106-109)
        int dn = c_far > 10;
        t = min(t + 1, 65535);
        if (s.max_steps > 0 && t >= s.max_steps) dn = 1; // gym TimeLimit
        cnt = (uint32_t)c_far | ((uint32_t)t << 8) | ((uint32_t)side << 24);
        if (leader) {
            const float2 rwd = wall_tr ? (wall_tg ? rw_ss : rw_sm) : (wall_tg ? rw_ms : rw_mm);
            reinterpret_cast<float2 *>(rew)[(size_t)it * s.n + e] = rwd;
            done_out[(size_t)it * s.n + e] = (uint8_t)dn;
        }
        const bool consume = live && dn != 0 && s.auto_reset != 0;

        const unsigned long long cm = __ballot(consume);
        bool switched = false;
        if (__builtin_expect(cm != 0ull, 0)) {
            // Track1v1Env.reset(): switch to the pre-generated next episode (k_gen). Everything it needs is already in
            // registers (fetched speculatively in `rows`); the tile copy into `maps` is a fire-and-forget store.
            if ((cm & 0xffffffffull) != 0ull)
                reinterpret_cast<uint4 *>(s.maps + (size_t)e0 * kTileWords)[lane] = sp_tile0;
            if ((cm >> 32) != 0ull)
                reinterpret_cast<uint4 *>(s.maps + (size_t)(e0 + 1) * kTileWords)[lane] = sp_tile1;
            if (consume) {
                switched = true;
                if (s.np_mt && leader) s.np_mt[(size_t)e * kNpStateWords + kNpTermD2] = d2;   // (numpy-stream handles: info['distance'])
                pos = sp_pos; plan = sp_plan; tctr = sp_tctr; d2 = sp_d2;
                cnt = (uint32_t)side_of_cfg(cfg) << 24;
                episode = sp_episode + 1u;
                if (NAV && nav) {       // the new episode's plan (made by k_gen at its reset); the prepared next plan is void
                    uint4 *dst = reinterpret_cast<uint4 *>(s.dirf + (size_t)e * kDirWords) + (lane & 31);
                    dst[0] = sp_nd0; dst[32] = sp_nd1; dst[64] = sp_nd2; dst[96] = sp_nd3;
                    navgoal = sp_navgoal; navgoal_dirty = false;
                    if (leader) s.p_state[pq_state_index(s, episode - 1u, e)] = 0u;    // the finished episode's queue is void
                }
                const size_t so = (size_t)(episode & 1u) * s.n + e;     // the slot just consumed (parity of the new episode)
                if (leader) {
                    s.goals[e] = sp_goals; s.episode[e] = episode; s.navgoal[e] = sp_navgoal;
                    s.gen_req[so] = stamp + (uint32_t)it;
                }
                r0 = (int)(pos & 0xffu); c0 = (int)((pos >> 8) & 0xffu);
                r1 = (int)((pos >> 16) & 0xffu); c1 = (int)(pos >> 24);
                rowbase = (ag ? r1 : r0) - 7;
                // later steps of a multi-step launch read the NEW map from its n_maps slot, which nobody writes during this
                // launch (the copy into `maps` above is for later launches): no store -> load hazard through the vector L1
                gmap = s.n_maps + so * kTileWords;
            }
        }
        if (leader) {
            s.pos[e] = pos; s.cnt[e] = cnt; s.d2[e] = d2;
            if (consume || dirty) { s.plan[e] = plan; s.tctr[e] = tctr; }
            if (NAV && navgoal_dirty && !consume) s.navgoal[e] = navgoal;
        }

        T2D_STAMP(3);
        if (obs != nullptr) {
            // _get_obs / _get_partial_obs (track_1v1.py:287-326) in closed form: own cell = own colour, the other agent's
            // colour where it falls in the window, else the map bit, outside the map 1
            const int oside = (int)(cnt >> 24);
            const int my_r = ag ? r1 : r0, my_c = ag ? c1 : c0, ot_r = ag ? r0 : r1, ot_c = ag ? c0 : c1;
            const int rr = rowbase + k, y = rr - (my_r - T2D_POB);
            if (k < 15 && (unsigned)y < (unsigned)T2D_WIN) {
                // the lane that holds a row extracts its 13 window bits; after an episode switch they come ready-made from
                // the next-episode slot (n_win, lane k <-> window row k - 1)
                uint32_t bits = switched ? sp_win : window_row_bits(w0, w1, w2, oside, my_c);
                if (y == T2D_POB) bits &= ~(1u << T2D_POB);                 // coloured cells carry no map bit
                const int xo = ot_c - (my_c - T2D_POB);
                if (rr == ot_r && (unsigned)xo < (unsigned)T2D_WIN) bits &= ~(1u << xo);
                st[sl * 26 + ag * 13 + y] = bits;
            }
            mk[lane] = 0u; mk[lane + 64] = 0u;
            if (lane < 169 - 128) mk[lane + 128] = 0u;
            if (k == 0) {   // the four (slot, agent) windows: own colour at the centre (:313), the other agent if inside
                uint8_t *mbp = reinterpret_cast<uint8_t *>(mk) + sl * kObsPerEnv + ag * (T2D_WIN * T2D_WIN)
                               + T2D_POB * T2D_WIN + T2D_POB;
                const int dr = ot_r - my_r, dc = ot_c - my_c;
                if ((unsigned)(dr + T2D_POB) < (unsigned)T2D_WIN && (unsigned)(dc + T2D_POB) < (unsigned)T2D_WIN)
                    mbp[dr * T2D_WIN + dc] = ag ? 2 : 4;
                mbp[0] = ag ? 4 : 2;                                        // written last: own colour wins when co-located
            }
            wave_lds_sync();
            T2D_STAMP(4);
            const size_t ebase = ((size_t)it * s.n + e0) * kObsPerEnv;   // first cell of the pair in the output
            // output groups of this lane: cells 4q .. 4q+3 of the pair's 676, q = lane + 64 i; cell p lives in window row
            // p / 13 (52 rows: slot, agent, y), column p % 13 (locals, not members: an array member indexed in a loop keeps
            // the whole struct out of registers)
            int rj[3], rx[3];
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const int p = 4 * (lane + 64 * i);
                rj[i] = (p * 1261) >> 14;               // p / 13, exact for p < 676
                rx[i] = p - 13 * rj[i];
            }
            // all LDS reads of the three groups first (independent: their latencies overlap), then the math, then the stores
            uint32_t glo[3], ghi[3], gmk[3];
#pragma unroll
            for (int i = 0; i < 3; i++) {
                glo[i] = st[rj[i]]; ghi[i] = st[rj[i] + 1];
                gmk[i] = mk[lane + 64 * i];                      // (the mark plane is padded to 192 dwords: always in bounds)
            }
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const int q = lane + 64 * i, p = 4 * q;
                const uint32_t nib = ((glo[i] | (ghi[i] << 13)) >> rx[i]) & 0xfu;
                const uint32_t bytes = (__umul24(nib, 0x204081u) & 0x01010101u) | gmk[i];   // bit c -> byte c, then colours
                if (p >= cells) continue;
                const int nvalid = min(4, cells - p);
                if (OBS == OBS_U8) {
                    uint8_t *o = reinterpret_cast<uint8_t *>(obs) + ebase + p;
                    if (nvalid == 4) {
                        if (s.nt_obs) __builtin_nontemporal_store(bytes, reinterpret_cast<uint32_t *>(o));
                        else *reinterpret_cast<uint32_t *>(o) = bytes;
                    } else for (int c = 0; c < nvalid; c++) o[c] = (uint8_t)(bytes >> (8 * c));
                } else {
                    float *o = reinterpret_cast<float *>(obs) + ebase + p;
                    const float f0 = (float)(bytes & 0xffu), f1 = (float)((bytes >> 8) & 0xffu);
                    const float f2 = (float)((bytes >> 16) & 0xffu), f3 = (float)(bytes >> 24);
                    if (OBS == OBS_F32_VEC4 && nvalid == 4) {
                        // beyond the Infinity Cache (hundreds of MB of observations per step) the stores stream past the
                        // caches: written once, read by another kernel much later
                        typedef float f32x4_t __attribute__((ext_vector_type(4)));
                        const f32x4_t v4 = {f0, f1, f2, f3};
                        if (s.nt_obs) __builtin_nontemporal_store(v4, reinterpret_cast<f32x4_t *>(o));
                        else *reinterpret_cast<float4 *>(o) = make_float4(f0, f1, f2, f3);
                    } else {
                        o[0] = f0;
                        if (nvalid > 1) o[1] = f1;
                        if (nvalid > 2) o[2] = f2;
                        if (nvalid > 3) o[3] = f3;
                    }
                }
            }
            if (MULTI) wave_lds_sync();   // the next step overwrites the stage
            T2D_STAMP(5);
            if (T2D_EXP == 6) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); T2D_STAMP(6); }
        }
        return dn;
    }
};

template <bool RANDOM, bool MULTI, int ADT, int OBS, bool RAM, bool NAV = false>
__global__ __launch_bounds__(64 * kStep2Waves) void k_step2(DevState s, const void *act0, const void *act1, void *obs, float *rew,
                                               uint8_t *done_out, uint32_t aseed_lo, uint32_t aseed_hi,
                                               uint32_t step_idx, uint32_t stamp, int nsteps)
{
    __shared__ __attribute__((aligned(16))) uint32_t stage2[kStep2Waves][kStage2Words];
    __shared__ __attribute__((aligned(16))) uint32_t navtiles[NAV ? kStep2Waves : 1][NAV ? kTileWords : 4];
    const int lane = (int)(threadIdx.x & 63u);
    const int wave = uni((int)(threadIdx.x >> 6));
    const int e0 = ((int)blockIdx.x * kStep2Waves + wave) * 2;
    if (e0 >= s.n) return;
    // T2D_EXP (compile-time, 0 in the product build): latency probes behind the N = 4096 numbers in DESIGN.md section 3 —
    // 5 = empty kernel (launch floor), 7 = state loads -> one store, 8 = state -> rows + reward table -> one store,
    // 6 = s_memtime timeline (tools/timeline_probe.py); built and timed by tools/exp_variants.sh.
    if (T2D_EXP == 5) { if (lane == 0) done_out[e0] = 0; return; }
    Step2<MULTI, OBS, RAM, NAV> S;
    S.init(s, e0, lane, stage2[wave]);
    if (NAV) S.tile = navtiles[wave];
    long long act_raw0 = 0, act_raw1 = 0;
    if (!RANDOM) {
        act_raw0 = load_action_raw<ADT>(act0, S.e);
        act_raw1 = load_action_raw<ADT>(act1, S.e);
    }
    if (T2D_EXP == 7) {   // latency probe: level-1 loads -> one store
        if (S.leader) done_out[S.e] = (uint8_t)((S.pos ^ S.cnt ^ S.cfg ^ (uint32_t)act_raw0 ^ (uint32_t)act_raw1) & 1u);
        return;
    }
    S.init2(s, obs, rew, done_out);
    for (int it = 0; it < (MULTI ? nsteps : 1); it++) {
        int a_tr, a_tg;
        if (RANDOM) {
            u32x4 w = philox4x32_10(aseed_lo, aseed_hi, step_idx + (uint32_t)it, 0u, S.genv, STREAM_ACTION);
            a_tr = (int)(w.x & (uint32_t)s.amask); a_tg = (int)(w.y & (uint32_t)s.amask);
        } else {
            if (S.leader && (act_raw0 < 0 || act_raw0 > s.amask || act_raw1 < 0 || act_raw1 > s.amask)) atomicOr(s.faults, 1u);
            a_tr = (int)(act_raw0 & s.amask); a_tg = (int)(act_raw1 & s.amask);
        }
        S.rows(s);
        S.finish(s, a_tr, a_tg, it, obs, rew, done_out, stamp);
        if (T2D_EXP == 8) return;
    }
}

// =====================================================================================================
// k_act_step — the END of a rollout step as one launch: both players' LSTM cells, actor heads and categorical draws
// (train.py:81-88 -> player_util.py:44-67 -> model.py:238-265 of the reference: tracker first, the tracker-aware target
// sees the tracker's fresh action) and, with those two actions still in registers, the env step + observation of
// k_step2. One wavefront = one env PAIR: lanes 0..31 serve env e0, lanes 32..63 env e0 + 1 — for the cells a lane owns
// four of the R = 128 hidden units of its env's row (the layout of atr::k_lstm_cell_fwd<true>: same expressions, same
// butterfly, same Philox key -> bit-identical states and actions), for the env step it is k_step2's slot | agent | k.
// What it replaces: two cell + head + draw launches, the action round trip through memory and the step launch — three
// dependent launches of 5-10 us each at every batch size. The env's state loads go out first and its map-row loads as
// soon as the state is there, so both fly under the cells' arithmetic.
// NA: compile-time number of actions (4 = every registered id; 8 = the 'Moore' table) — the head and draw code is sized to it
// The body of k_act_step for ONE env pair = one wavefront, as a function: k_act_step calls it from a workgroup of kStep2Waves
// waves (STAGE_EMB: the embedding rows are parked in LDS here, behind a workgroup barrier every wave reaches), k_coop_step from a
// workgroup whose earlier phases already staged them (STAGE_EMB = false: no barrier inside, waves without an env pair simply
// do not call). tid / nthreads: the caller's thread index and workgroup size (the staging loop's shape).
template <int OBS, bool RAM, bool ENV, int NA, bool NAV, bool STAGE_EMB, int NTHREADS>
__device__ __forceinline__ void act_pair(const DevState &s, const atr_act_step &a, void *obs, float *rew, uint8_t *done_out,
                                         uint32_t stamp, int e0, int lane, int tid, uint32_t *stage, uint32_t *navtile,
                                         float *emb_lds)
{
    using namespace atr;
    const int n = ENV ? s.n : a.N;
    // the embedding rows (A x 4R floats, 8 KB) go to LDS now: the target's cell reads row a_tracker the moment the tracker's
    // draw is known — an LDS read instead of a dependent trip to L2 in the middle of the kernel's serial chain
    constexpr int kEmbTrips = STAGE_EMB ? NA * 128 / NTHREADS : 1;
    float4 emb_st[kEmbTrips];
    if (STAGE_EMB && a.emb) {
#pragma unroll
        for (int i = 0; i < kEmbTrips; i++) emb_st[i] = ld4(a.emb + 4 * (tid + i * NTHREADS));
    }
    const bool active = e0 < n;         // (no early return: every wave reaches the workgroup barrier below)
    Step2<false, OBS, RAM, NAV> S;
    if (ENV && active) { S.init(s, e0, lane, stage); if (NAV) S.tile = navtile; }    // env state loads first ...
    const int sl = lane >> 5, q = lane & 31, j = q * 4;
    const bool live = e0 + sl < n;
    const int e = live ? e0 + sl : (active ? e0 : 0);
    constexpr int R = 128;
    // ... then everything both cells read, in one batch. The gate pre-activations are touched exactly once (written by the
    // GEMM just before, never read again): non-temporal loads keep them from displacing the weights in L2.
    const float k = a.done_prev ? (a.done_prev[e] == 0 ? 1.0f : 0.0f) : 1.0f;
    float4 pre[2][4], cp[2], aw[2][NA];
    auto ldnt = [](const float *p_) {
        const float4 *q_ = reinterpret_cast<const float4 *>(p_);
        return make_float4(__builtin_nontemporal_load(&q_->x), __builtin_nontemporal_load(&q_->y),
                           __builtin_nontemporal_load(&q_->z), __builtin_nontemporal_load(&q_->w));
    };
    // a.ig[0] == NULL (round 6): the tracker's cell already ran — as the epilogue of the step's LSTMCell product (atr_gate_cell,
    // csrc/gate_cell_hip.hip) — and h_out[0] / c_out[0] hold its fresh state: nothing of its gates is read here, its hidden row
    // is (wave-uniform branch: the argument is the same for every lane of the launch)
    const bool pre0 = a.ig[0] == nullptr;
    float4 h0_pre = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pre0) h0_pre = ld4(a.h_out[0] + (size_t)e * R + j);
#pragma unroll
    for (int p = 0; p < 2; p++) {
#pragma unroll
        for (int x = 0; x < NA; x++) aw[p][x] = ld4(a.actor_w[p] + x * R + j);
        if (p == 0 && pre0) continue;
        const float *ig = a.ig[p] + (size_t)e * 4 * R + j;
#pragma unroll
        for (int g = 0; g < 4; g++) pre[p][g] = ldnt(ig + g * R);
        if (a.hg[p]) {
            const float *hg = a.hg[p] + (size_t)e * 4 * R + j;
#pragma unroll
            for (int g = 0; g < 4; g++) pre[p][g] = fma4(k, ldnt(hg + g * R), pre[p][g]);
        }
        if (a.bias[p]) {
#pragma unroll
            for (int g = 0; g < 4; g++) pre[p][g] = fma4(1.0f, ld4(a.bias[p] + g * R + j), pre[p][g]);
        }
        cp[p] = ld4(a.c_prev[p] + (size_t)e * R + j);
    }
    const unsigned long long ctr = *a.counter;
    if (STAGE_EMB) {
        if (a.emb) {
#pragma unroll
            for (int i = 0; i < kEmbTrips; i++) st4(emb_lds + 4 * (tid + i * NTHREADS), emb_st[i]);
        }
        __syncthreads();
    }
    if (!active) return;
    if (ENV) { S.init2(s, obs, rew, done_out); S.rows(s); }   // (waits for the state only: vector loads return in order)
    int act[2] = {0, 0};
    float4 hkeep[2];
#pragma unroll
    for (int p = 0; p < 2; p++) {
        if (p == 1 && a.emb) {     // tracker-aware target: fc_action_tracker(one_hot(a_tracker)) projected through W_ih (model.py:193-194)
            const float *em = emb_lds + act[0] * 4 * R + j;
#pragma unroll
            for (int g = 0; g < 4; g++) pre[1][g] = fma4(1.0f, ld4(em + g * R), pre[1][g]);
        }
        CellOut o;
        if (p == 0 && pre0) o.h = h0_pre;
        else o = cell4(pre[p][0], pre[p][1], pre[p][2], pre[p][3], cp[p], k);
        hkeep[p] = o.h;
        if (live && !(p == 0 && pre0)) {
            st4(a.h_out[p] + (size_t)e * R + j, o.h);
            st4(a.c_out[p] + (size_t)e * R + j, o.c);
            if (a.acts[p]) {   // read next by the learner, a whole rollout later: streamed past the caches
                float *ac = a.acts[p] + (size_t)e * 4 * R + j;
                auto stnt = [](float *d_, const float4 &v_) {
                    __builtin_nontemporal_store(v_.x, d_); __builtin_nontemporal_store(v_.y, d_ + 1);
                    __builtin_nontemporal_store(v_.z, d_ + 2); __builtin_nontemporal_store(v_.w, d_ + 3);
                };
                stnt(ac, o.gi); stnt(ac + R, o.gf); stnt(ac + 2 * R, o.gg); stnt(ac + 3 * R, o.go);
            }
        }
        // actor head on the fresh row (the expressions and the butterfly of atr::head_logits, sized to NA)
        float logit[kMaxActions];
#pragma unroll
        for (int x = 0; x < kMaxActions; x++) {
            logit[x] = -INFINITY;
            if (x < NA) {
                const float4 w = aw[p][x];
                logit[x] = fmaf(o.h.x, w.x, fmaf(o.h.y, w.y, fmaf(o.h.z, w.z, o.h.w * w.w)));
            }
        }
#pragma unroll
        for (int msk = 1; msk < 32; msk <<= 1)
#pragma unroll
            for (int x = 0; x < NA; x++) logit[x] += __shfl_xor(logit[x], msk, 64);
        int mine = 0;
        if (q == 0) {
#pragma unroll
            for (int x = 0; x < NA; x++) logit[x] = logit[x] + a.actor_b[p][x];
            // row index e and ordinal + p: the numbers atr_lstm_cell_forward_act* draw for this row
            mine = draw_action(logit, NA, e, ctr, a.seed, a.ordinal + (unsigned)p);
            if (live) a.actions_out[(size_t)p * n + e] = (long long)mine;
        }
        act[p] = __shfl(mine, lane & 32, 64);
    }
    if (ENV) {
        const int dn = S.finish(s, act[0] & s.amask, act[1] & s.amask, 0, obs, rew, done_out, stamp);
        // the hidden rows as the NEXT step's LSTMCell GEMM reads them: zero for an env whose episode just ended (the reset() of
        // train.py:73-74 -> player_util.py:98-102), written into the h columns of that step's [features | k h] rows
        if (a.hm_out[0] && live) {
            const float km = dn ? 0.0f : 1.0f;
#pragma unroll
            for (int p = 0; p < 2; p++)
                st4(a.hm_out[p] + (size_t)e * a.hm_ld + j,
                    make_float4(km * hkeep[p].x, km * hkeep[p].y, km * hkeep[p].z, km * hkeep[p].w));
        }
    }
}

template <int OBS, bool RAM, bool ENV, int NA, bool NAV = false>
__global__ __launch_bounds__(64 * kStep2Waves) void k_act_step(DevState s, atr_act_step a, void *obs, float *rew,
                                                               uint8_t *done_out, uint32_t stamp)
{
    __shared__ __attribute__((aligned(16))) uint32_t stage2[kStep2Waves][kStage2Words];
    __shared__ __attribute__((aligned(16))) float emb_lds[NA * 4 * 128];   // the tracker-action embedding table
    __shared__ __attribute__((aligned(16))) uint32_t navtiles[NAV ? kStep2Waves : 1][NAV ? kTileWords : 4];
    const int lane = (int)(threadIdx.x & 63u);
    const int wave = uni((int)(threadIdx.x >> 6));
    const int e0 = ((int)blockIdx.x * kStep2Waves + wave) * 2;
    act_pair<OBS, RAM, ENV, NA, NAV, true, 64 * kStep2Waves>(s, a, obs, rew, done_out, stamp, e0, lane, (int)threadIdx.x, stage2[wave],
                                                             navtiles[NAV ? wave : 0], emb_lds);
}

// ---- the small-shard rollout step as ONE launch after the stem: fc + ReLU -> LSTMCell GEMM -> cells + heads + draws + env step ----
// At 512 / 1024 envs per GPU (the 8- and 4-GPU forms of the headline: 4096 envs sharded) a rollout step is four dependent
// launches of 8-10 us each, every one of them 2-3 us of matrix-pipe work behind a launch, a cold L2 and a drain. A layer
// boundary needs activations to cross between CUs, and a chip-wide barrier inside a kernel costs more than the launch it
// would replace (tools/microbench/grid_barrier.hip: 5.7 us — the agent-scope release / acquire is an L2 write-back +
// invalidate, because the eight XCDs have eight L2s). But nothing forces a layer's rows to cross XCDs: this kernel gives
// every XCD one eighth of the envs END TO END. Its 32 workgroups split each layer's tiles among themselves (coop_gemm.h),
// write them with plain stores — write-through to the ONE L2 they share — and meet at a barrier that lives in that L2: an
// atomic counter (RMW atomics execute in the L2) polled with sc1 loads (device scope: they miss the CU's L1). No L2
// write-back, no invalidate: 1 us per barrier (tools/microbench/xcd_barrier.hip), half a dependent launch. What makes the
// plain loads after the barrier safe without an L1 invalidate: a layer's activations are written once, in whole 128-byte
// lines (32 columns of a row), to addresses no CU has read since its L1 was invalidated at kernel start.
//   phase A  encoder fc + ReLU of both players (perception.py:81,90) into the feature columns of this step's [features | k h_prev] rows
//   phase B  both GEMMs of nn.LSTMCell of both players (model.py:110,137,172,203) as one K = F + R product over those rows
//            -> gate pre-activations (scratch that never leaves the L2 hot set)
//   phase C  act_pair: k_act_step's body, one wavefront per env pair of this XCD
// Which workgroup serves which XCD is read from the hardware (XCC_ID), its rank among that XCD's workgroups from a ticket;
// the ticket counter doubles as the barrier: every launch adds exactly 3 x per_xcd to it (claim + two arrivals), so the
// ticket modulo that period is the rank and ticket - rank the launch's base (64-bit: never wraps). Launches that use one
// counter block must not overlap (they are steps of one rollout: a dependent chain).
#define T2D_XCC_ID() (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7)       // hwreg(HW_REG_XCC_ID, 0, 4)
constexpr int kCoopSpinCap = 1 << 18;      // ~0.1 s of polling: a barrier that never completes ends in a fault flag, not a hang
constexpr uint32_t kFaultCoopPlacement = 2u, kFaultCoopTimeout = 4u, kFaultCoopShape = 8u;

struct CoopStep {
    const float *y[2], *fc_w[2], *fc_b[2], *w_cat[2];
    long long ldy[2];
    int kfc[2];
    float *fh, *gates;
    long long fh_pstride, fh_ld;
    int F;
    unsigned long long *ctl;     // [8 XCDs][16]: one counter per XCD, 128 bytes apart
    int per_xcd;
    unsigned long long *probe;   // nullable (tools/coop_step_timeline.py): [workgroups][8] s_memtime stamps at the phase boundaries
};
#define T2D_COOP_STAMP(i) do { if (c.probe && tid == 0) c.probe[(size_t)blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)

// The barrier in two halves, so that a workgroup can request what it needs next (weights, env state: nothing the other
// workgroups are still writing) while it waits.
//   arrive: every wave has its stores acknowledged by the L2 (gfx9 counts stores in vmcnt; a workgroup-scope release fence
//           emits no wait in this mode — checked in the ISA — so it is spelled out), then one atomic increment
//   wait:   thread 0 polls the counter
__device__ __forceinline__ void xcd_arrive(unsigned long long *ctr, int tid)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void xcd_wait(unsigned long long *ctr, unsigned long long target, uint32_t *faults, int tid)
{
    if (tid == 0) {
        // sc1 loads: device scope, they miss the CU's L1 and are served by the L2 the atomics execute in. (They queue behind
        // whatever vector loads wave 0 has in flight — the weight prefetch issued between the two halves, which the wave needs
        // next anyway. A scalar-load poll, s_load glc, would not, and was measured SLOWER: 3.0 against 1.9 us per barrier.)
        int spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < kCoopSpinCap)
            __builtin_amdgcn_s_sleep(1);
        if (spins >= kCoopSpinCap) atomicOr(faults, kFaultCoopTimeout);
    }
    __syncthreads();
}

template <int OBS, bool RAM, int NA, bool NAV>
__global__ __launch_bounds__(atr::kCoopThreads) void k_coop_step(DevState s, atr_act_step a, CoopStep c, void *obs, float *rew,
                                                                uint8_t *done_out, uint32_t stamp)
{
    using namespace atr;
    extern __shared__ __attribute__((aligned(16))) unsigned char coop_smem[];
    float *emb_lds = reinterpret_cast<float *>(coop_smem);                         // NA x 4R floats
    CoopLds &L = *reinterpret_cast<CoopLds *>(coop_smem + NA * 4 * 128 * sizeof(float));
    uint32_t *stage_base = reinterpret_cast<uint32_t *>(L.part);                   // phase C re-uses the partial tiles' space
    uint32_t *nav_base = stage_base + kCoopWaves * kStage2Words;
    __shared__ unsigned long long s_ticket;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
    const int xcc = (int)T2D_XCC_ID();
    unsigned long long *ctr = c.ctl + 16 * xcc;
    T2D_COOP_STAMP(0);
    if (tid == 0) s_ticket = __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (a.emb)
        for (int i = tid; i < NA * 128; i += kCoopThreads) st4(emb_lds + 4 * i, ld4(a.emb + 4 * i));
    __syncthreads();
    const int P = c.per_xcd;
    const unsigned long long period = 3ull * (unsigned long long)P, ticket = s_ticket;
    const int slot = (int)(ticket % period);
    const unsigned long long base = ticket - (unsigned long long)slot;
    if (slot >= P) {            // more workgroups on this XCD than planned (the dispatcher did not deal them round-robin)
        if (tid == 0) atomicOr(s.faults, kFaultCoopPlacement);
        return;
    }
    const int N = s.n, Rx = N >> 3, n_rt = Rx >> 4, row_x = xcc * Rx;
    constexpr int R4 = 4 * 128;
    T2D_COOP_STAMP(1);
    if (c.probe && tid == 0) c.probe[(size_t)blockIdx.x * 8 + 7] = ((unsigned long long)xcc << 32) | (unsigned)slot;
    CoopPipe pipe;
    // ---- phase A: fc + ReLU tiles of this XCD's rows, both players (player 0's units first: with equal counts every
    // workgroup gets the same mix of the two K's) ----
    {
        const int ct_n = c.F >> 5, upp = n_rt * ct_n, units = 2 * upp;
        const int mine = slot < units ? (units - slot + P - 1) / P : 0;
        if (tid < mine && tid < kCoopMaxUnits) {
            const int u = slot + tid * P, p = u / upp, idx = u - p * upp, rt = idx / ct_n, ct = idx - rt * ct_n;
            const int row0 = row_x + rt * 16;
            CoopUnit &U = L.units[tid];
            U.a = c.y[p] + (size_t)row0 * c.ldy[p]; U.lda = (int)c.ldy[p];
            U.w = c.fc_w[p] + (size_t)(ct * 32) * c.kfc[p]; U.ldw = c.kfc[p];
            U.bias = c.fc_b[p] + ct * 32;
            U.c = c.fh + (size_t)p * c.fh_pstride + (size_t)row0 * c.fh_ld + ct * 32; U.ldc = (int)c.fh_ld;
            U.K = c.kfc[p]; U.rows_valid = 16; U.relu = 1;
        }
        if (tid == 0) L.probe = c.probe ? c.probe + (size_t)gridDim.x * 8 + (size_t)blockIdx.x * 16 : nullptr;
        __syncthreads();
        const int nu = min(mine, kCoopMaxUnits);
        coop_plan(L, nu, tid, pipe);
        coop_prefetch_w(pipe);
        coop_prefetch_a(pipe);
        if (mine > kCoopMaxUnits || !coop_run(L, nu, tid, pipe)) { if (lane == 0) atomicOr(s.faults, kFaultCoopShape); }
    }
    T2D_COOP_STAMP(2);
    // ---- phase B: gate pre-activations = [features | k h_prev] [W_ih | W_hh]^T (bias added by the cell). Its weights are
    // requested BEFORE the barrier completes (nobody writes those), its rows after ----
    {
        const int ct_n = R4 >> 5, upp = n_rt * ct_n, units = 2 * upp;
        const int mine = slot < units ? (units - slot + P - 1) / P : 0;
        xcd_arrive(ctr, tid);            // (its __syncthreads also ends every thread's reads of phase A's unit table)
        if (tid < mine && tid < kCoopMaxUnits) {
            const int u = slot + tid * P, p = u / upp, idx = u - p * upp, ct = idx / n_rt, rt = idx - ct * n_rt;
            const int row0 = row_x + rt * 16;
            CoopUnit &U = L.units[tid];
            U.a = c.fh + (size_t)p * c.fh_pstride + (size_t)row0 * c.fh_ld; U.lda = (int)c.fh_ld;
            U.w = c.w_cat[p] + (size_t)(ct * 32) * c.fh_ld; U.ldw = (int)c.fh_ld;
            U.bias = nullptr;
            U.c = c.gates + ((size_t)p * N + row0) * R4 + ct * 32; U.ldc = R4;
            U.K = (int)c.fh_ld; U.rows_valid = 16; U.relu = 0;
        }
        if (tid == 0) L.probe = c.probe ? c.probe + (size_t)gridDim.x * 8 + (size_t)blockIdx.x * 16 + 8 : nullptr;
        __syncthreads();
        const int nu = min(mine, kCoopMaxUnits);
        coop_plan(L, nu, tid, pipe);
        coop_prefetch_w(pipe);
        xcd_wait(ctr, base + 2ull * (unsigned long long)P, s.faults, tid);
        T2D_COOP_STAMP(3);
        coop_prefetch_a(pipe);
        if (mine > kCoopMaxUnits || !coop_run(L, nu, tid, pipe)) { if (lane == 0) atomicOr(s.faults, kFaultCoopShape); }
    }
    T2D_COOP_STAMP(4);
    xcd_arrive(ctr, tid);
    xcd_wait(ctr, base + 3ull * (unsigned long long)P, s.faults, tid);
    T2D_COOP_STAMP(5);
    // ---- phase C: cells + heads + draws + env step, one wavefront per env pair of this XCD ----
    const int pi = slot + wave * P;
    if (pi < (Rx >> 1))
        act_pair<OBS, RAM, true, NA, NAV, false, kCoopThreads>(s, a, obs, rew, done_out, stamp, row_x + 2 * pi, lane, tid,
                                                               stage_base + wave * kStage2Words, nav_base + (NAV ? wave : 0) * kTileWords,
                                                               emb_lds);
    if (c.probe) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); T2D_COOP_STAMP(6); }
}

__global__ void k_build_reward_lut(float2 *lut)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= 3 * kLutN) return;
    const int cls = i / kLutN;
    double a, b;
    reward_f64((uint32_t)(i - cls * kLutN), cls == 1 ? 1.0 : (cls == 2 ? -0.5 : 0.0), a, b);
    lut[i] = make_float2((float)a, (float)b);
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
    bool reset_done;   // every env has a current episode
    std::vector<uint8_t> has_episode;   // per env: given one by t2d_inject (which may come env by env)
    int n_has_episode = 0;
    bool primed;       // every env has a valid next slot
    bool has_nav;
    bool has_ram;      // some env has the scripted Ram target (selects the step kernel variant with the plan code)
    bool has_rpf;      // some env has the RPF patrol target (an agent may then stand on a wall of the env's own map)
    uint32_t random_step;
    bool has_navmode;  // some env has the Nav target (its next plan is prefetched once per stamp window)
    bool has_rpfmode;  // some env has the RPF patrol target: those handles stay on the general step kernel (k_env)
    uint32_t gen_every; // shortest possible episode = longest admissible generator period, in steps
    // Step stamps run 1..cycle; the cycle is one window of gen_every steps (generator in order on the caller's
    // stream) or, with the asynchronous generator, two windows of gen_every / 2: the slots consumed in window w are
    // refilled by a launch forked onto gen_stream after w's last step and joined before w's next first step — one
    // full window later, which is still sooner than any of those slots can be needed again (2 * win <= gen_every).
    uint32_t win, cycle;
    uint32_t phase;     // steps launched in the current cycle = stamp of the last step launch
    bool gen_async;
    bool pending[2];    // a forked generator launch of window w has not been joined yet
    hipStream_t gen_stream;
    hipEvent_t ev_fork, ev_join[2];
    // atr_coop_env_step's XCD counters: kCoopCtlSets blocks of [8 XCDs][16] u64 (one counter per XCD, 128 bytes apart), one
    // block per distinct grid size (a counter's period is 3 x grid / 8: launches of different sizes must not share one)
    unsigned long long *coop_ctl = nullptr;
    int coop_grid[4] = {0, 0, 0, 0};
    // k_pregrow (Maze handles): forked launches run on pg_stream between a fork event on the caller's stream and a join the
    // next generator pass (or t2d_generator_join / t2d_flush) waits for; pg_auto: every generator pass forks one behind itself
    bool has_maze = false, pg_auto = false, pg_pending = false;
    // t2d_np_attach on a handle with Ram targets: their draws interleave with the resets (np_inter), so episodes are generated
    // inside t2d_reset and RamAgent.step runs as k_ram_np before every step launch, its actions in np_act ([N] of 8 bytes)
    bool np_inter = false;
    void *np_act = nullptr;
    hipStream_t pg_stream = nullptr;
    hipEvent_t ev_pg_fork = nullptr, ev_pg_join = nullptr;
};
constexpr int kCoopCtlSets = 4, kCoopCtlWords = 8 * 16;

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
extern "C" int t2d_config_size(void) { return (int)sizeof(t2d_config); }

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
    bool has_nav = false, has_rpf = false, has_ram = false, has_navmode = false, has_rpfmode = false;
    int n_maze = 0;
    if (cfg->obs_type > T2D_OBS_FULL) return fail(T2D_ERR_INVALID, "t2d_create: obs_type %u", cfg->obs_type);
    if (cfg->action_type > T2D_ACTIONS_MOORE) return fail(T2D_ERR_INVALID, "t2d_create: action_type %u", cfg->action_type);
    for (int i = 0; i < n; i++) {
        uint32_t mt = cfg->map_type_per_env ? cfg->map_type_per_env[i] : cfg->map_type;
        uint32_t tm = cfg->target_mode_per_env ? cfg->target_mode_per_env[i] : cfg->target_mode;
        uint32_t lv = cfg->level_per_env ? cfg->level_per_env[i] : cfg->level;
        if (mt > T2D_MAP_EMPTY) return fail(T2D_ERR_INVALID, "t2d_create: map_type %u (env %d)", mt, i);
        if (tm > T2D_TGT_EXT) return fail(T2D_ERR_INVALID, "t2d_create: target_mode %u (env %d)", tm, i);
        has_nav = has_nav || tm == T2D_TGT_NAV || tm == T2D_TGT_RPF;
        has_rpf = has_rpf || tm == T2D_TGT_RPF || tm == T2D_TGT_EXT;   // agents may stand on walls
        has_ram = has_ram || tm == T2D_TGT_RAM;
        has_navmode = has_navmode || tm == T2D_TGT_NAV;
        has_rpfmode = has_rpfmode || tm == T2D_TGT_RPF;
        n_maze += mt == T2D_MAP_MAZE;
        if (lv > 15) return fail(T2D_ERR_INVALID, "t2d_create: level %u (env %d)", lv, i);
        // the scripted targets plan in the four-move table: RamAgent would draw from 8 actions (navigator.py:74-75) and the
        // Navigator's A* fails outright in the reference (Astar_solver.py:138,163-169 index a 4-entry table with 8 actions)
        if (cfg->action_type == T2D_ACTIONS_MOORE && (tm == T2D_TGT_RAM || tm == T2D_TGT_NAV || tm == T2D_TGT_RPF))
            return fail(T2D_ERR_INVALID, "t2d_create: action_type Moore with a scripted target (env %d)", i);
        hcfg[(size_t)i] = mt | (tm << 2) | (lv << 5);
    }
    if (cfg->obs_type == T2D_OBS_FULL && n_maze != 0 && n_maze != n)
        return fail(T2D_ERR_INVALID, "t2d_create: obs_type Full needs one map side per handle (81 for Maze, else 82)");
    DeviceGuard guard(cfg->device);
    t2d_handle *h = new (std::nothrow) t2d_handle();
    if (!h) return fail(T2D_ERR_INVALID, "t2d_create: out of host memory");
    std::memset(&h->s, 0, sizeof(h->s));
    h->device = cfg->device;
    h->reset_done = false; h->primed = false; h->has_nav = has_nav; h->has_rpf = has_rpf; h->has_ram = has_ram;
    h->has_navmode = has_navmode; h->has_rpfmode = has_rpfmode; h->has_maze = n_maze > 0;
    h->random_step = 0; h->phase = 0;
    h->gen_async = false; h->pending[0] = h->pending[1] = false;
    h->gen_stream = nullptr; h->ev_fork = nullptr; h->ev_join[0] = h->ev_join[1] = nullptr;
    // An episode lasts L = min(11, max_episode_steps) steps or more (done needs 11 consecutive far steps, track_1v1.py:106-111,
    // or the TimeLimit), and every env has TWO pre-generated episodes: a slot consumed at step q is needed again only after
    // the other slot has been consumed too, i.e. not before step q + 2 L. Launching the generator every G <= 2 L - 2 steps, in
    // order on the caller's stream, therefore always refills a slot before its next use: G = 20 = one A3C rollout.
    {
        uint32_t L = 11u;
        if (cfg->max_episode_steps > 0 && (uint32_t)cfg->max_episode_steps < L) L = (uint32_t)cfg->max_episode_steps;
        h->gen_every = L >= 2u ? 2u * L - 2u : 1u;
    }
    h->win = h->cycle = h->gen_every;
    DevState &s = h->s;
    s.n = n; s.env_base = cfg->env_id_base;
    s.k0 = (uint32_t)cfg->seed; s.k1 = (uint32_t)(cfg->seed >> 32);
    s.max_steps = cfg->max_episode_steps; s.auto_reset = cfg->auto_reset ? 1 : 0;
    s.obs_full = cfg->obs_type == T2D_OBS_FULL ? 1 : 0;
    s.amask = cfg->action_type == T2D_ACTIONS_MOORE ? 7 : 3;
    s.obs_side = n_maze == n ? 81 : 82;
    {
        const char *nt = getenv("T2D_NT_OBS_MIN_ENVS");       // (experiment hook; default: from 8192 envs up — measured
        // neutral at 4096, +7 % at 16384, +15-30 % from 65536 up; non-temporal LOADS of the map rows changed nothing)
        s.nt_obs = n >= (nt ? atoi(nt) : 8192) ? 1 : 0;
    }
    const size_t nb = (size_t)n * sizeof(uint32_t), tb = (size_t)n * kTileWords * sizeof(uint32_t),
                 db = (size_t)n * kDirWords * sizeof(uint32_t);
    hipError_t err = hipSuccess;
    auto alloc = [&](uint32_t **p, size_t bytes) {
        if (err == hipSuccess) err = hipMalloc((void **)p, bytes);
        if (err == hipSuccess) err = hipMemset(*p, 0, bytes);
    };
    alloc(&s.maps, tb); alloc(&s.n_maps, 2 * tb);
    alloc(&s.n_win, (size_t)2 * n * 32 * sizeof(uint32_t));
    uint32_t **arrs[] = {&s.pos, &s.goals, &s.cnt, &s.cfg, &s.episode, &s.plan, &s.tctr, &s.navgoal, &s.d2, &s.nav2};
    for (auto a : arrs) alloc(a, nb);
    uint32_t **narrs[] = {&s.n_pos, &s.n_goals, &s.n_plan, &s.n_tctr, &s.n_navgoal, &s.n_d2, &s.gen_req, &s.n_nav2};
    for (auto a : narrs) alloc(a, 2 * nb);          // [2][N]: the two next-episode slots
    if (has_nav) {
        alloc(&s.dirf, db); alloc(&s.n_dirf, 2 * db);
        alloc(&s.p_field, (size_t)6 * n * kPlanWords * sizeof(uint32_t));      // three episode sets x two queued plans per env
        alloc(&s.p_goal, 6 * nb); alloc(&s.p_tctr, 6 * nb); alloc(&s.p_state, 3 * nb);
    }
    if (n_maze > 0 && cfg->obs_type != T2D_OBS_FULL) {
        alloc(&s.g_maps, (size_t)4 * tb);
        alloc(&s.g_ep, 4 * nb);
        if (err == hipSuccess) err = hipMemset(s.g_ep, 0xff, 4 * nb);      // kPoolEmpty
        alloc(&s.pg_stats, 4 * sizeof(uint32_t));
    }
    alloc(&s.faults, sizeof(uint32_t));
    alloc(reinterpret_cast<uint32_t **>(&h->coop_ctl), (size_t)kCoopCtlSets * kCoopCtlWords * sizeof(unsigned long long));
    float2 *lut = nullptr;
    if (err == hipSuccess) err = hipMalloc((void **)&lut, (size_t)3 * kLutN * sizeof(float2));
    s.rew_lut = lut;
    if (err == hipSuccess) {
        hipLaunchKernelGGL(k_build_reward_lut, dim3((3 * kLutN + 255) / 256), dim3(256), 0, 0, lut);
        err = hipGetLastError();
        if (err == hipSuccess) err = hipDeviceSynchronize();
    }
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
    void *ptrs[] = {s.maps, s.n_maps, s.pos, s.goals, s.cnt, s.cfg, s.episode, s.plan, s.tctr, s.navgoal, s.d2,
                    s.n_pos, s.n_goals, s.n_plan, s.n_tctr, s.n_navgoal, s.n_d2, s.gen_req, s.dirf, s.n_dirf, s.faults,
                    (void *)s.rew_lut, s.nav2, s.n_nav2, s.p_field, s.p_goal, s.p_tctr, s.p_state, s.n_win};
    if (h->gen_stream) {
        (void)hipStreamSynchronize(h->gen_stream);
        (void)hipStreamDestroy(h->gen_stream);
        (void)hipEventDestroy(h->ev_fork); (void)hipEventDestroy(h->ev_join[0]); (void)hipEventDestroy(h->ev_join[1]);
    }
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (h->coop_ctl) (void)hipFree(h->coop_ctl);
    if (h->np_act) (void)hipFree(h->np_act);
    for (void *p : {(void *)s.g_maps, (void *)s.g_ep, (void *)s.pg_stats, (void *)s.np_mt, (void *)s.np_nav})
        if (p) (void)hipFree(p);
    if (h->pg_stream) {
        (void)hipStreamSynchronize(h->pg_stream);
        (void)hipStreamDestroy(h->pg_stream);
        (void)hipEventDestroy(h->ev_pg_fork); (void)hipEventDestroy(h->ev_pg_join);
    }
    delete h;
    return T2D_OK;
}

static inline dim3 env_grid(int n) { return dim3((unsigned)((n + kWavesPerBlock - 1) / kWavesPerBlock)); }
constexpr int kGenNavMaxEnvs = 2048;

static void launch_gen(t2d_handle *h, hipStream_t st, uint32_t lo, uint32_t hi, int force, bool prefetch = false,
                       const uint8_t *mask = nullptr)
{
    if (h->s.np_mt) {                               // numpy-stream handle (t2d_np_attach): one wave per env, slots in episode order
        if (h->np_inter && !force) return;          // (interleaved streams: nothing is generated ahead of a reset)
        hipLaunchKernelGGL(k_gen_np, dim3((unsigned)((h->s.n + kNpWaves - 1) / kNpWaves)), dim3(64 * kNpWaves), 0, st, h->s, lo, hi,
                           force, mask, h->np_inter ? 1 : 0);
        return;
    }
    const dim3 grid = env_grid(2 * h->s.n);        // one wave per (slot, env)
    // Nav targets, small shards: one WORKGROUP per (slot, env), the three floods of a generated episode on three waves
    // (k_gen_nav: the pass is as long as its slowest generation — 238 -> 206 us at 1024 Maze + Nav envs). A large batch has
    // thousands of generations per pass and is bound by their throughput, where four waves per generation lose to one
    // (8192 envs: 927 against 784 us per pass): those keep k_gen.
    if (h->has_navmode && h->s.n <= kGenNavMaxEnvs) {
        const unsigned blocks = 2u * (unsigned)h->s.n + (prefetch ? (unsigned)((h->s.n + kWavesPerBlock - 1) / kWavesPerBlock) : 0u);
        if (prefetch) hipLaunchKernelGGL((k_gen_nav<true>), dim3(blocks), dim3(256), 0, st, h->s, lo, hi, force);
        else hipLaunchKernelGGL((k_gen_nav<false>), dim3(blocks), dim3(256), 0, st, h->s, lo, hi, force);
        return;
    }
    if (h->has_nav && prefetch) hipLaunchKernelGGL((k_gen<true, true>), env_grid(3 * h->s.n), dim3(256), 0, st, h->s, lo, hi, force);
    else if (h->has_nav) hipLaunchKernelGGL((k_gen<true, false>), grid, dim3(256), 0, st, h->s, lo, hi, force);
    else hipLaunchKernelGGL((k_gen<false, false>), grid, dim3(256), 0, st, h->s, lo, hi, force);
}

// Make `st` wait for every forked generator launch.
static int join_generator(t2d_handle *h, hipStream_t st)
{
    if (h->pg_pending) {                       // a forked k_pregrow
        HIP_TRY(hipStreamWaitEvent(st, h->ev_pg_join, 0));
        h->pg_pending = false;
    }
    for (int w = 0; w < 2; w++)
        if (h->pending[w]) {
            HIP_TRY(hipStreamWaitEvent(st, h->ev_join[w], 0));
            h->pending[w] = false;
        }
    return T2D_OK;
}

// k_pregrow on `st` itself (fork == 0: in order, e.g. the learner's stream of the pipelined schedule, beside the next rollout)
// or forked onto the handle's own stream, to be joined by the next generator pass / t2d_generator_join / t2d_flush.
static int launch_pregrow(t2d_handle *h, hipStream_t st, int fork)
{
    if (!h->s.g_maps || h->s.np_mt) return T2D_OK;      // (numpy-stream handles build their maps from the stream, in the pass)
    const dim3 grid = env_grid(2 * h->s.n);
    if (!fork) {
        hipLaunchKernelGGL(k_pregrow, grid, dim3(256), 0, st, h->s);
        HIP_TRY(hipGetLastError());
        return T2D_OK;
    }
    if (!h->pg_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&h->pg_stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&h->ev_pg_fork, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&h->ev_pg_join, hipEventDisableTiming));
    }
    if (h->pg_pending) {                       // one in flight at a time
        HIP_TRY(hipStreamWaitEvent(st, h->ev_pg_join, 0));
        h->pg_pending = false;
    }
    HIP_TRY(hipEventRecord(h->ev_pg_fork, st));
    HIP_TRY(hipStreamWaitEvent(h->pg_stream, h->ev_pg_fork, 0));
    hipLaunchKernelGGL(k_pregrow, grid, dim3(256), 0, h->pg_stream, h->s);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(h->ev_pg_join, h->pg_stream));
    h->pg_pending = true;
    return T2D_OK;
}

// Before a step launch: at the first step of a window, the slots consumed in that window one cycle ago must be back.
static int window_begin(t2d_handle *h, hipStream_t st)
{
    if (!h->gen_async || h->phase % h->win != 0u) return T2D_OK;
    const int w = (int)(h->phase / h->win);
    if (h->pending[w]) {
        HIP_TRY(hipStreamWaitEvent(st, h->ev_join[w], 0));
        h->pending[w] = false;
    }
    return T2D_OK;
}

// After the step launches (phase already advanced): at the last step of a window, refill what the window consumed.
static int window_end(t2d_handle *h, hipStream_t st)
{
    if (h->phase == 0u || h->phase % h->win != 0u) return T2D_OK;
    const uint32_t w = h->phase / h->win - 1u, lo = w * h->win + 1u, hi = (w + 1u) * h->win;
    if (!h->gen_async) {
        if (h->pg_pending) {                            // (a forked k_pregrow: its mazes are what this pass is about to take)
            HIP_TRY(hipStreamWaitEvent(st, h->ev_pg_join, 0));
            h->pg_pending = false;
        }
        launch_gen(h, st, lo, hi, 0, h->has_navmode);   // the Nav plan prefetch rides in the same launch (other waves)
        if (h->pg_auto) { int rc = launch_pregrow(h, st, 1); if (rc) return rc; }
    } else {
        if (h->has_navmode) hipLaunchKernelGGL(k_nav_prefetch, env_grid(h->s.n), dim3(256), 0, st, h->s);
        HIP_TRY(hipEventRecord(h->ev_fork, st));
        HIP_TRY(hipStreamWaitEvent(h->gen_stream, h->ev_fork, 0));
        launch_gen(h, h->gen_stream, lo, hi, 0);
        HIP_TRY(hipEventRecord(h->ev_join[w], h->gen_stream));
        h->pending[w] = true;
    }
    HIP_TRY(hipGetLastError());
    if (h->phase >= h->cycle) h->phase = 0;
    return T2D_OK;
}

static inline dim3 pair_grid(int n) { return dim3((unsigned)(((n + 1) / 2 + kStep2Waves - 1) / kStep2Waves)); }
static inline bool use_step2(const t2d_handle *h) { return !h->has_rpfmode && !h->s.obs_full; }

// k_step2 launcher. obs_kind: OBS_F32_* picked from the pointer / stride alignment, or OBS_U8.
template <bool RANDOM, bool MULTI>
static void launch_step2(t2d_handle *h, hipStream_t st, const void *a0, const void *a1, int adt, void *obs, bool u8,
                         float *rew, uint8_t *done, uint32_t slo, uint32_t shi, uint32_t sidx, uint32_t stamp, int nsteps)
{
    if (a1 == nullptr) a1 = a0;
    const bool vec4 = ((uintptr_t)obs & 15u) == 0u && (!MULTI || (h->s.n & 1) == 0);
    const int kind = u8 ? OBS_U8 : (vec4 ? OBS_F32_VEC4 : OBS_F32_SCALAR);
#define T2D_LAUNCH2(ADTV, KIND)                                                                                         \
    do {                                                                                                                \
        if constexpr (!MULTI) {                                                                                         \
            if (h->has_navmode) {   /* Nav handles: one variant with the Ram code too (per-env mixed target modes) */  \
                hipLaunchKernelGGL((k_step2<RANDOM, false, ADTV, KIND, true, true>), pair_grid(h->s.n),                 \
                                   dim3(64 * kStep2Waves), 0, st, h->s, a0, a1, obs, rew, done, slo, shi, sidx, stamp, nsteps); \
                break;                                                                                                  \
            }                                                                                                           \
        }                                                                                                               \
        if (h->has_ram)                                                                                                 \
            hipLaunchKernelGGL((k_step2<RANDOM, MULTI, ADTV, KIND, true>), pair_grid(h->s.n), dim3(64 * kStep2Waves), 0, \
                               st, h->s,   \
                               a0, a1, obs, rew, done, slo, shi, sidx, stamp, nsteps);                                  \
        else                                                                                                            \
            hipLaunchKernelGGL((k_step2<RANDOM, MULTI, ADTV, KIND, false>), pair_grid(h->s.n), dim3(64 * kStep2Waves), 0, \
                               st, h->s,  \
                               a0, a1, obs, rew, done, slo, shi, sidx, stamp, nsteps);                                  \
    } while (0)
#define T2D_LAUNCH2K(ADTV)                                                                                              \
    do {                                                                                                                \
        if (kind == OBS_U8) T2D_LAUNCH2(ADTV, OBS_U8);                                                                  \
        else if (kind == OBS_F32_VEC4) T2D_LAUNCH2(ADTV, OBS_F32_VEC4);                                                 \
        else T2D_LAUNCH2(ADTV, OBS_F32_SCALAR);                                                                         \
    } while (0)
    if constexpr (RANDOM) {
        T2D_LAUNCH2K(T2D_ACT_I64);
    } else {
        if (adt == T2D_ACT_I64) T2D_LAUNCH2K(T2D_ACT_I64);
        else if (adt == T2D_ACT_I32) T2D_LAUNCH2K(T2D_ACT_I32);
        else T2D_LAUNCH2K(T2D_ACT_U8);
    }
#undef T2D_LAUNCH2K
#undef T2D_LAUNCH2
}

template <int OP, bool RANDOM>
static void launch_env(t2d_handle *h, hipStream_t st, const void *a0, const void *a1, int adt, const uint8_t *mask,
                       float *obs, float *rew, uint8_t *done, uint32_t slo, uint32_t shi, uint32_t sidx, uint32_t stamp)
{
    if (OP == OP_STEP && use_step2(h)) {
        launch_step2<RANDOM, false>(h, st, a0, a1, adt, obs, false, rew, done, slo, shi, sidx, stamp, 1);
        return;
    }
    if (a1 == nullptr) a1 = a0;   // scripted target: the value is overridden in the kernel
#define T2D_LAUNCH(NAVF, ADTV)                                                                                          \
    hipLaunchKernelGGL((k_env<OP, RANDOM, NAVF, false, ADTV>), env_grid(h->s.n), dim3(256), 0, st, h->s, a0, a1, adt,   \
                       mask, obs, rew, done, slo, shi, sidx, stamp, 1)
    if (OP != OP_STEP || RANDOM || adt == T2D_ACT_I64) {
        if (h->has_nav) T2D_LAUNCH(true, T2D_ACT_I64); else T2D_LAUNCH(false, T2D_ACT_I64);
    } else if (adt == T2D_ACT_I32) {
        if (h->has_nav) T2D_LAUNCH(true, T2D_ACT_I32); else T2D_LAUNCH(false, T2D_ACT_I32);
    } else {
        if (h->has_nav) T2D_LAUNCH(true, T2D_ACT_U8); else T2D_LAUNCH(false, T2D_ACT_U8);
    }
#undef T2D_LAUNCH
}

// Regenerate every consumed next-episode slot, in order on `st` (after joining the forked launches), and restart
// the stamps.
static int flush_impl(t2d_handle *h, hipStream_t st)
{
    int rc = join_generator(h, st);
    if (rc) return rc;
    if (h->s.auto_reset && h->phase % h->win != 0u) {
        launch_gen(h, st, 1u, h->cycle, 0);
        HIP_TRY(hipGetLastError());
    }
    h->phase = 0;
    return T2D_OK;
}

extern "C" int t2d_flush(t2d_handle *h, void *stream)
{
    if (!h) return fail(T2D_ERR_INVALID, "t2d_flush: null handle");
    DeviceGuard guard(h->device);
    return flush_impl(h, (hipStream_t)stream);
}

extern "C" int t2d_generator_async(t2d_handle *h, int enable, void *stream)
{
    if (!h) return fail(T2D_ERR_INVALID, "t2d_generator_async: null handle");
    DeviceGuard guard(h->device);
    int rc = flush_impl(h, (hipStream_t)stream);
    if (rc) return rc;
    if (enable && h->gen_every < 2u)
        return fail(T2D_ERR_INVALID, "t2d_generator_async: episodes of %u step(s) leave no window to overlap", h->gen_every);
    if (enable && !h->gen_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&h->gen_stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&h->ev_join[0], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&h->ev_join[1], hipEventDisableTiming));
    }
    h->gen_async = enable != 0;
    h->win = h->gen_async ? h->gen_every / 2u : h->gen_every;
    h->cycle = h->gen_async ? 2u * h->win : h->win;
    return T2D_OK;
}

extern "C" int t2d_generator_join(t2d_handle *h, void *stream)
{
    if (!h) return fail(T2D_ERR_INVALID, "t2d_generator_join: null handle");
    DeviceGuard guard(h->device);
    return join_generator(h, (hipStream_t)stream);
}

extern "C" int t2d_generator_cycle(const t2d_handle *h) { return h ? (int)h->cycle : T2D_ERR_INVALID; }

extern "C" int t2d_pregrow(t2d_handle *h, int mode, void *stream)
{
    if (!h) return fail(T2D_ERR_INVALID, "t2d_pregrow: null handle");
    if (mode < 0 || mode > 3) return fail(T2D_ERR_INVALID, "t2d_pregrow: mode %d", mode);
    DeviceGuard guard(h->device);
    if (mode == T2D_PREGROW_AUTO_ON || mode == T2D_PREGROW_AUTO_OFF) { h->pg_auto = mode == T2D_PREGROW_AUTO_ON; return T2D_OK; }
    return launch_pregrow(h, (hipStream_t)stream, mode == T2D_PREGROW_FORK ? 1 : 0);
}

extern "C" int t2d_np_attach(t2d_handle *h, const uint32_t *states_host)
{
    if (!h || !states_host) return fail(T2D_ERR_INVALID, "t2d_np_attach: null argument");
    if (h->primed || h->reset_done) return fail(T2D_ERR_STATE, "t2d_np_attach: attach the streams before the first t2d_reset");
    if ((h->has_ram || h->has_navmode || h->has_rpfmode) && h->s.auto_reset)
        return fail(T2D_ERR_INVALID, "t2d_np_attach: a Ram / Nav / RPF target draws from the stream between resets, so its next episode "
                                     "cannot be generated ahead of the in-launch auto-reset: create the handle with auto_reset = 0 "
                                     "and restart finished envs with t2d_reset(mask = done)");
    DeviceGuard guard(h->device);
    const int n = h->s.n;
    std::vector<uint32_t> padded((size_t)n * kNpStateWords, 0u);
    for (int i = 0; i < n; i++) {
        if (states_host[(size_t)i * 625 + 624] > 624u) return fail(T2D_ERR_INVALID, "t2d_np_attach: env %d: read position > 624", i);
        std::memcpy(&padded[(size_t)i * kNpStateWords], states_host + (size_t)i * 625, 625 * sizeof(uint32_t));
    }
    if (h->has_ram || h->has_navmode || h->has_rpfmode) {
        // The Ram / Nav envs' target is stepped by k_ram_np from the env's own stream; to the step kernels their mode becomes "the
        // target's action comes from outside" (T2D_TGT_EXT: w_p = 0 either way, track_1v1.py:147-152) and neither the Philox Ram
        // code nor the BFS Navigator of the device generators is selected (has_ram / has_nav off)
        std::vector<uint32_t> cfg((size_t)n);
        HIP_TRY(hipMemcpy(cfg.data(), h->s.cfg, cfg.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (int i = 0; i < n; i++) {
            const uint32_t tm = (cfg[(size_t)i] >> 2) & 7u;
            if (tm == (uint32_t)TGT_RAM || tm == (uint32_t)TGT_NAV || tm == (uint32_t)TGT_RPF) {
                cfg[(size_t)i] = (cfg[(size_t)i] & ~(7u << 2)) | ((uint32_t)T2D_TGT_EXT << 2);
                padded[(size_t)i * kNpStateWords + kNpRamFlag] = tm == (uint32_t)TGT_RAM ? 1u : (tm == (uint32_t)TGT_NAV ? 2u : 3u);
            }
        }
        HIP_TRY(hipMemcpy(h->s.cfg, cfg.data(), cfg.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        if (!h->np_act) HIP_TRY(hipMalloc(&h->np_act, (size_t)n * 8));
        if ((h->has_navmode || h->has_rpfmode) && !h->s.np_nav) HIP_TRY(hipMalloc((void **)&h->s.np_nav, (size_t)n * kNavBytes));
        // (an RPF env's agents may stand on cells its own map keeps as walls: has_rpf stays, as for every handle with Ext envs)
        h->has_rpf = h->has_rpf || h->has_rpfmode;
        h->np_inter = true; h->has_ram = false; h->has_nav = false; h->has_navmode = false; h->has_rpfmode = false;
    }
    if (!h->s.np_mt) HIP_TRY(hipMalloc((void **)&h->s.np_mt, padded.size() * sizeof(uint32_t)));
    HIP_TRY(hipMemcpy(h->s.np_mt, padded.data(), padded.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    return T2D_OK;
}

// The device A* alone on a caller's maze (one wave; for the known-answer tests against the reference's searches).
__global__ __launch_bounds__(64) void k_astar_probe(const uint32_t *tile_g, unsigned char *scratch, uint32_t from, uint32_t goal,
                                                    int *len_out, uint32_t *faults)
{
    __shared__ __attribute__((aligned(16))) uint32_t tile[kTileWords];
    const int lane = (int)threadIdx.x;
    reinterpret_cast<uint4 *>(tile)[lane] = reinterpret_cast<const uint4 *>(tile_g)[lane];
    wave_lds_sync();
    const int len = astar_np(tile, scratch, from, goal, lane, faults);
    if (lane == 0) *len_out = len;
}

extern "C" int t2d_np_astar_device(int device, const uint8_t *maze, int32_t side, const int32_t start[2], const int32_t goal[2],
                                   int32_t *actions, int32_t max_len, int32_t *n, int32_t *solvable)
{
    if (!maze || !start || !goal || !actions || !n || !solvable || side < 3 || side > 82)
        return fail(T2D_ERR_INVALID, "t2d_np_astar_device: bad argument");
    DeviceGuard guard(device);
    std::vector<uint32_t> tile((size_t)kTileWords, 0u);
    for (int r = 0; r < side; r++)
        for (int c = 0; c < side; c++)
            if (maze[r * side + c]) tile[(size_t)(r * kRowWords + (c >> 5))] |= 1u << (c & 31);
    uint32_t *d_tile = nullptr, *d_faults = nullptr;
    unsigned char *d_scr = nullptr;
    int *d_len = nullptr;
    HIP_TRY(hipMalloc((void **)&d_tile, kTileWords * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void **)&d_scr, kNavBytes));
    HIP_TRY(hipMalloc((void **)&d_len, sizeof(int)));
    HIP_TRY(hipMalloc((void **)&d_faults, sizeof(uint32_t)));
    HIP_TRY(hipMemcpy(d_tile, tile.data(), kTileWords * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(d_faults, 0, sizeof(uint32_t)));
    hipLaunchKernelGGL(k_astar_probe, dim3(1), dim3(64), 0, nullptr, d_tile, d_scr, (uint32_t)start[0] | ((uint32_t)start[1] << 8),
                       (uint32_t)goal[0] | ((uint32_t)goal[1] << 8), d_len, d_faults);
    int len = -1;
    uint32_t faults = 0;
    hipError_t e1 = hipMemcpy(&len, d_len, sizeof(int), hipMemcpyDeviceToHost);
    hipError_t e2 = hipMemcpy(&faults, d_faults, sizeof(uint32_t), hipMemcpyDeviceToHost);
    std::vector<unsigned char> plan((size_t)(len > 0 ? len : 0));
    hipError_t e3 = len > 0 ? hipMemcpy(plan.data(), d_scr, (size_t)len, hipMemcpyDeviceToHost) : hipSuccess;
    (void)hipFree(d_tile); (void)hipFree(d_scr); (void)hipFree(d_len); (void)hipFree(d_faults);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) return fail(T2D_ERR_HIP, "t2d_np_astar_device: copy back failed");
    if (faults) return fail(T2D_ERR_STATE, "t2d_np_astar_device: the search ran out of node / heap / plan space (fault 0x%x)", faults);
    *solvable = len >= 0 ? 1 : 0;
    *n = len > 0 ? len : 0;
    for (int i = 0; i < len && i < max_len; i++) actions[i] = plan[(size_t)i];
    return T2D_OK;
}

extern "C" int t2d_np_terminal_d2(t2d_handle *h, int first, int count, uint32_t *d2_host, void *stream)
{
    if (!h || !d2_host) return fail(T2D_ERR_INVALID, "t2d_np_terminal_d2: null argument");
    if (!h->s.np_mt) return fail(T2D_ERR_STATE, "t2d_np_terminal_d2: no numpy streams attached (t2d_np_attach)");
    if (first < 0 || count < 0 || first + count > h->s.n) return fail(T2D_ERR_INVALID, "t2d_np_terminal_d2: range [%d, %d)", first, first + count);
    if (count == 0) return T2D_OK;
    DeviceGuard guard(h->device);
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemcpy2DAsync(d2_host, sizeof(uint32_t), h->s.np_mt + (size_t)first * kNpStateWords + kNpTermD2,
                             kNpStateWords * sizeof(uint32_t), sizeof(uint32_t), (size_t)count, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return T2D_OK;
}

extern "C" int t2d_pregrow_stats(t2d_handle *h, uint32_t stats_host[4], void *stream)
{
    if (!h || !stats_host) return fail(T2D_ERR_INVALID, "t2d_pregrow_stats: null argument");
    DeviceGuard guard(h->device);
    std::memset(stats_host, 0, 4 * sizeof(uint32_t));
    if (!h->s.pg_stats) return T2D_OK;
    HIP_TRY(hipMemcpyAsync(stats_host, h->s.pg_stats, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return T2D_OK;
}

// RamAgent.step() of the numpy-stream Ram envs, ahead of the step launch: the target's action of this step (track_1v1.py:80-82)
// for them, the caller's for everybody else, in the handle's own buffer (the caller's dtype).
static const void *np_ram_actions(t2d_handle *h, hipStream_t st, const void *a1, int adt)
{
    hipLaunchKernelGGL(k_ram_np, dim3((unsigned)((h->s.n + kNpWaves - 1) / kNpWaves)), dim3(64 * kNpWaves), 0, st, h->s, a1,
                       h->np_act, adt);
    return h->np_act;
}

template <bool RANDOM>
static int step_impl(t2d_handle *h, hipStream_t st, const void *a0, const void *a1, int adt, float *obs, float *rew,
                     uint8_t *done, uint32_t slo, uint32_t shi, uint32_t sidx)
{
    if (h->s.auto_reset) {
        int rc = window_begin(h, st);
        if (rc) return rc;
        h->phase++;
    }
    if (h->np_inter) {
        if (RANDOM) return fail(T2D_ERR_INVALID, "random-action rollouts are not defined for numpy-stream Ram handles");
        a1 = np_ram_actions(h, st, a1, adt);
    }
    launch_env<OP_STEP, RANDOM>(h, st, a0, a1, adt, nullptr, obs, rew, done, slo, shi, sidx, h->phase);
    HIP_TRY(hipGetLastError());
    if (h->s.auto_reset) return window_end(h, st);
    return T2D_OK;
}

extern "C" int t2d_reset(t2d_handle *h, const uint8_t *mask_dev, float *obs_dev, void *stream)
{
    if (!h) return fail(T2D_ERR_INVALID, "t2d_reset: null handle");
    DeviceGuard guard(h->device);
    hipStream_t st = (hipStream_t)stream;
    int rc = flush_impl(h, st);
    if (rc) return rc;
    if (mask_dev != nullptr && !h->reset_done)
        return fail(T2D_ERR_STATE, "t2d_reset: the first reset must cover every env (mask == NULL)");
    if (h->np_inter) {   // numpy streams with Ram targets: the episode each restarted env is about to start is drawn NOW, from where
        launch_gen(h, st, 0u, 0u, 1, false, mask_dev);      // its stream stands after the finished episode's step-time draws
        h->primed = true;
    } else if (!h->primed) { // first use: generate episode 1 into every next slot
        launch_gen(h, st, 0u, 0u, 1);
        h->primed = true;
    }
    launch_env<OP_RESET, false>(h, st, nullptr, nullptr, 0, mask_dev, obs_dev, nullptr, nullptr, 0u, 0u, 0u, 1u);
    launch_gen(h, st, 1u, 1u, 0); // refill the consumed next slots, in order on the caller's stream
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
    if (h->s.auto_reset && !h->primed) return fail(T2D_ERR_STATE, "t2d_step: auto_reset needs one t2d_reset before stepping");
    DeviceGuard guard(h->device);
    return step_impl<false>(h, (hipStream_t)stream, act_tracker_dev, act_target_dev, act_dtype, obs_dev, rew_dev,
                            done_dev, 0u, 0u, 0u);
}

extern "C" int t2d_step_u8(t2d_handle *h, const void *act_tracker_dev, const void *act_target_dev, int act_dtype,
                           uint8_t *obs_u8_dev, float *rew_dev, uint8_t *done_dev, void *stream)
{
    if (!h) return fail(T2D_ERR_INVALID, "t2d_step_u8: null handle");
    if (!act_tracker_dev || !rew_dev || !done_dev) return fail(T2D_ERR_INVALID, "t2d_step_u8: null buffer");
    if (act_dtype < T2D_ACT_U8 || act_dtype > T2D_ACT_I64) return fail(T2D_ERR_INVALID, "t2d_step_u8: act_dtype %d", act_dtype);
    if (!use_step2(h))
        return fail(T2D_ERR_INVALID, "t2d_step_u8: u8 observations exist for 'Partial' ids without the RPF target");
    if (((uintptr_t)obs_u8_dev & 3u) != 0u) return fail(T2D_ERR_INVALID, "t2d_step_u8: obs buffer must be 4-byte aligned");
    if (!h->reset_done) return fail(T2D_ERR_STATE, "t2d_step_u8: call t2d_reset (all envs) or t2d_inject first");
    if (h->s.auto_reset && !h->primed) return fail(T2D_ERR_STATE, "t2d_step_u8: auto_reset needs one t2d_reset before stepping");
    DeviceGuard guard(h->device);
    hipStream_t st = (hipStream_t)stream;
    if (h->s.auto_reset) {
        int rc = window_begin(h, st);
        if (rc) return rc;
        h->phase++;
    }
    if (h->np_inter) act_target_dev = np_ram_actions(h, st, act_target_dev, act_dtype);
    launch_step2<false, false>(h, st, act_tracker_dev, act_target_dev, act_dtype, obs_u8_dev, true, rew_dev, done_dev, 0u,
                               0u, 0u, h->phase, 1);
    HIP_TRY(hipGetLastError());
    if (h->s.auto_reset) return window_end(h, st);
    return T2D_OK;
}

// The rollout step's last launch (see k_act_step): policy cells + draws + env step. Same stamping / generator schedule
// as t2d_step.
extern "C" int atr_act_env_step(t2d_handle *h, const atr_act_step *args, void *obs_dev, int obs_is_u8, float *rew_dev,
                                uint8_t *done_dev, void *stream)
{
    if (!args) return fail(T2D_ERR_INVALID, "atr_act_env_step: null argument");
    const atr_act_step &a = *args;
    for (int p = 0; p < 2; p++)
        if ((!a.ig[p] && p != 0) || !a.h_out[p] || !a.actor_w[p] || !a.actor_b[p] || (a.ig[p] && (!a.c_prev[p] || !a.c_out[p])))
            return fail(T2D_ERR_INVALID, "atr_act_env_step: null policy buffer (player %d)", p);
    if (!a.ig[0] && (a.hg[0] || a.acts[0]))
        return fail(T2D_ERR_INVALID, "atr_act_env_step: ig[0] == NULL (the tracker's cell ran in atr_gate_cell) excludes hg[0] / acts[0]");
    if (!a.actions_out || !a.counter || a.R != 128 || (a.A != 4 && a.A != 8))
        return fail(T2D_ERR_INVALID, "atr_act_env_step: needs R = 128, A = 4 or 8, actions_out, counter");
    if ((a.hm_out[0] != nullptr) != (a.hm_out[1] != nullptr) || (a.hm_out[0] && (a.hm_ld < a.R || (a.hm_ld & 3) ||
                                                                                 (((uintptr_t)a.hm_out[0] | (uintptr_t)a.hm_out[1]) & 15u))))
        return fail(T2D_ERR_INVALID, "atr_act_env_step: hm_out needs both players, 16-byte aligned, hm_ld >= R and a multiple of 4");
    if (a.hm_out[0] && !h) return fail(T2D_ERR_INVALID, "atr_act_env_step: hm_out (masked hidden rows) needs the env step");
    hipStream_t st = (hipStream_t)stream;
    if (!h) {   // policy half only (the learner's bootstrap step: one more actor step, no env step)
        if (a.N <= 0) return fail(T2D_ERR_INVALID, "atr_act_env_step: N must be > 0 without an env handle");
        DevState none;
        std::memset(&none, 0, sizeof(none));
        const unsigned grid = (unsigned)(((a.N + 1) / 2 + kStep2Waves - 1) / kStep2Waves);
        if (a.A == 4)
            hipLaunchKernelGGL((k_act_step<OBS_U8, false, false, 4>), dim3(grid), dim3(64 * kStep2Waves), 0, st, none, a, nullptr,
                               nullptr, nullptr, 0u);
        else
            hipLaunchKernelGGL((k_act_step<OBS_U8, false, false, 8>), dim3(grid), dim3(64 * kStep2Waves), 0, st, none, a, nullptr,
                               nullptr, nullptr, 0u);
        HIP_TRY(hipGetLastError());
        return T2D_OK;
    }
    if (!rew_dev || !done_dev || !obs_dev) return fail(T2D_ERR_INVALID, "atr_act_env_step: null env buffer");
    if (h->np_inter)
        return fail(T2D_ERR_INVALID, "atr_act_env_step: numpy-stream Ram handles take the target's action from k_ram_np before the "
                                     "step launch (t2d_step / t2d_step_u8), not from the policy's draw inside this kernel");
    if (!use_step2(h))
        return fail(T2D_ERR_INVALID, "atr_act_env_step: exists for 'Partial' ids without the RPF target (the k_step2 family)");
    if (a.N != h->s.n) return fail(T2D_ERR_INVALID, "atr_act_env_step: policy batch %d != %d envs", a.N, h->s.n);
    if (a.A != h->s.amask + 1) return fail(T2D_ERR_INVALID, "atr_act_env_step: %d policy actions, env has %d", a.A, h->s.amask + 1);
    if (obs_is_u8 && ((uintptr_t)obs_dev & 3u) != 0u) return fail(T2D_ERR_INVALID, "atr_act_env_step: obs buffer must be 4-byte aligned");
    if (!h->reset_done) return fail(T2D_ERR_STATE, "atr_act_env_step: call t2d_reset (all envs) or t2d_inject first");
    if (h->s.auto_reset && !h->primed) return fail(T2D_ERR_STATE, "atr_act_env_step: auto_reset needs one t2d_reset before stepping");
    DeviceGuard guard(h->device);
    if (h->s.auto_reset) {
        int rc = window_begin(h, st);
        if (rc) return rc;
        h->phase++;
    }
    const int kind = obs_is_u8 ? OBS_U8 : ((((uintptr_t)obs_dev & 15u) == 0u) ? OBS_F32_VEC4 : OBS_F32_SCALAR);
#define T2D_LAUNCH_ACT2(KIND, RAMV, NAV, NAVF)                                                                          \
    hipLaunchKernelGGL((k_act_step<KIND, RAMV, true, NAV, NAVF>), pair_grid(h->s.n), dim3(64 * kStep2Waves), 0, st, h->s, a, \
                       obs_dev, rew_dev, done_dev, h->phase)
#define T2D_LAUNCH_ACT(KIND)                                                                                           \
    do {                                                                                                               \
        if (h->has_navmode) { if (a.A == 4) T2D_LAUNCH_ACT2(KIND, true, 4, true); else T2D_LAUNCH_ACT2(KIND, true, 8, true); } \
        else if (a.A == 4) { if (h->has_ram) T2D_LAUNCH_ACT2(KIND, true, 4, false); else T2D_LAUNCH_ACT2(KIND, false, 4, false); } \
        else { if (h->has_ram) T2D_LAUNCH_ACT2(KIND, true, 8, false); else T2D_LAUNCH_ACT2(KIND, false, 8, false); }   \
    } while (0)
    if (kind == OBS_U8) T2D_LAUNCH_ACT(OBS_U8);
    else if (kind == OBS_F32_VEC4) T2D_LAUNCH_ACT(OBS_F32_VEC4);
    else T2D_LAUNCH_ACT(OBS_F32_SCALAR);
#undef T2D_LAUNCH_ACT2
#undef T2D_LAUNCH_ACT
    HIP_TRY(hipGetLastError());
    if (h->s.auto_reset) return window_end(h, st);
    return T2D_OK;
}

// The small-shard rollout step after the stem as ONE launch (k_coop_step): fc + ReLU of both encoders, the LSTMCell GEMM of
// both players, then everything atr_act_env_step does. `act` as for atr_act_env_step with bias[p] = b_ih + b_hh, hg = NULL and
// hm_out set; act->ig is ignored (the gate pre-activations live in coop->gates).
extern "C" int atr_coop_env_step(t2d_handle *h, const atr_act_step *act, const atr_coop_step *coop, void *obs_dev, int obs_is_u8,
                                 float *rew_dev, uint8_t *done_dev, void *stream)
{
    if (!h || !act || !coop) return fail(T2D_ERR_INVALID, "atr_coop_env_step: null argument");
    if (h->np_inter) return fail(T2D_ERR_INVALID, "atr_coop_env_step: not for numpy-stream Ram handles (see atr_act_env_step)");
    atr_act_step a = *act;
    const atr_coop_step &k = *coop;
    const int N = h->s.n, G = k.workgroups, F = k.F, R = 128;
    for (int p = 0; p < 2; p++) {
        if (!a.c_prev[p] || !a.h_out[p] || !a.c_out[p] || !a.actor_w[p] || !a.actor_b[p] || !a.bias[p] || !a.hm_out[p])
            return fail(T2D_ERR_INVALID, "atr_coop_env_step: null policy buffer (player %d)", p);
        if (!k.y[p] || !k.fc_w[p] || !k.fc_b[p] || !k.w_cat[p])
            return fail(T2D_ERR_INVALID, "atr_coop_env_step: null layer buffer (player %d)", p);
        if (k.kfc[p] <= 0 || (k.kfc[p] & 31) || k.ldy[p] < k.kfc[p] || (k.ldy[p] & 3))
            return fail(T2D_ERR_INVALID, "atr_coop_env_step: fc input width must be a multiple of 32, row stride a multiple of 4");
        if (((uintptr_t)k.y[p] | (uintptr_t)k.fc_w[p] | (uintptr_t)k.fc_b[p] | (uintptr_t)k.w_cat[p] | (uintptr_t)a.hm_out[p]) & 15u)
            return fail(T2D_ERR_INVALID, "atr_coop_env_step: pointers must be 16-byte aligned");
        a.ig[p] = k.gates + (size_t)p * N * 4 * R;
        a.hg[p] = nullptr;
    }
    if (!a.actions_out || !a.counter || a.R != R || (a.A != 4 && a.A != 8))
        return fail(T2D_ERR_INVALID, "atr_coop_env_step: needs R = 128, A = 4 or 8, actions_out, counter");
    if (!k.fh || !k.gates || F <= 0 || (F & 31) || k.fh_ld != F + R || (k.fh_pstride & 31) ||
        (((uintptr_t)k.fh | (uintptr_t)k.gates) & 127u))
        return fail(T2D_ERR_INVALID, "atr_coop_env_step: rows must be [F | R] floats wide (F a multiple of 32), 128-byte aligned");
    if (a.hm_ld < R || (a.hm_ld & 3)) return fail(T2D_ERR_INVALID, "atr_coop_env_step: hm_ld >= R and a multiple of 4");
    if (!rew_dev || !done_dev || !obs_dev) return fail(T2D_ERR_INVALID, "atr_coop_env_step: null env buffer");
    if (!use_step2(h))
        return fail(T2D_ERR_INVALID, "atr_coop_env_step: exists for 'Partial' ids without the RPF target (the k_step2 family)");
    if (a.N != N) return fail(T2D_ERR_INVALID, "atr_coop_env_step: policy batch %d != %d envs", a.N, N);
    if (a.A != h->s.amask + 1) return fail(T2D_ERR_INVALID, "atr_coop_env_step: %d policy actions, env has %d", a.A, h->s.amask + 1);
    if (obs_is_u8 && ((uintptr_t)obs_dev & 3u) != 0u) return fail(T2D_ERR_INVALID, "atr_coop_env_step: obs buffer must be 4-byte aligned");
    // shape limits: every XCD gets N / 8 envs in whole 16-row tiles, a workgroup at most kCoopMaxUnits tiles per layer and
    // kCoopWaves env pairs
    if (G < 8 || (G & 7) || G > 1024) return fail(T2D_ERR_INVALID, "atr_coop_env_step: workgroups must be a multiple of 8");
    const int P = G / 8, Rx = N / 8;
    if ((N & 127) != 0) return fail(T2D_ERR_INVALID, "atr_coop_env_step: needs a multiple of 128 envs (got %d)", N);
    const int unitsA = 2 * (Rx / 16) * (F / 32), unitsB = 2 * (Rx / 16) * (4 * R / 32);
    if ((unitsA + P - 1) / P > atr::kCoopMaxUnits || (unitsB + P - 1) / P > atr::kCoopMaxUnits || (Rx / 2 + P - 1) / P > atr::kCoopWaves)
        return fail(T2D_ERR_INVALID, "atr_coop_env_step: %d envs are too many for %d workgroups", N, G);
    if (!h->reset_done) return fail(T2D_ERR_STATE, "atr_coop_env_step: call t2d_reset (all envs) or t2d_inject first");
    if (h->s.auto_reset && !h->primed) return fail(T2D_ERR_STATE, "atr_coop_env_step: auto_reset needs one t2d_reset before stepping");
    DeviceGuard guard(h->device);
    int set = -1;
    for (int i = 0; i < kCoopCtlSets && set < 0; i++)
        if (h->coop_grid[i] == G || h->coop_grid[i] == 0) { h->coop_grid[i] = G; set = i; }
    if (set < 0) return fail(T2D_ERR_INVALID, "atr_coop_env_step: more than %d distinct grid sizes on one handle", kCoopCtlSets);
    CoopStep c;
    for (int p = 0; p < 2; p++) {
        c.y[p] = k.y[p]; c.fc_w[p] = k.fc_w[p]; c.fc_b[p] = k.fc_b[p]; c.w_cat[p] = k.w_cat[p]; c.ldy[p] = k.ldy[p]; c.kfc[p] = k.kfc[p];
    }
    c.fh = k.fh; c.gates = k.gates; c.fh_pstride = k.fh_pstride; c.fh_ld = k.fh_ld; c.F = F;
    c.ctl = h->coop_ctl + (size_t)set * kCoopCtlWords; c.per_xcd = P;
    c.probe = (unsigned long long *)k.probe;
    hipStream_t st = (hipStream_t)stream;
    if (h->s.auto_reset) {
        int rc = window_begin(h, st);
        if (rc) return rc;
        h->phase++;
    }
    const int kind = obs_is_u8 ? OBS_U8 : ((((uintptr_t)obs_dev & 15u) == 0u) ? OBS_F32_VEC4 : OBS_F32_SCALAR);
    const size_t lds = (size_t)a.A * 4 * 128 * sizeof(float) + sizeof(atr::CoopLds);
    static_assert(sizeof(atr::CoopLds::part) + sizeof(atr::CoopLds::stage) >= atr::kCoopWaves * (kStage2Words + kTileWords) * sizeof(uint32_t), "phase C's LDS aliases the partial tiles");
#define T2D_LAUNCH_COOP2(KIND, RAMV, NAV, NAVF)                                                                        \
    do {                                                                                                               \
        static bool attr_ = false;                                                                                     \
        if (!attr_) {    /* (static + dynamic LDS beyond 64 KB needs the opt-in) */                                    \
            HIP_TRY(hipFuncSetAttribute((const void *)k_coop_step<KIND, RAMV, NAV, NAVF>,                              \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                       \
            attr_ = true;                                                                                              \
        }                                                                                                              \
        hipLaunchKernelGGL((k_coop_step<KIND, RAMV, NAV, NAVF>), dim3((unsigned)G), dim3(atr::kCoopThreads), lds, st, h->s, a, c, \
                           obs_dev, rew_dev, done_dev, h->phase);                                                      \
    } while (0)
#define T2D_LAUNCH_COOP(KIND)                                                                                          \
    do {                                                                                                               \
        if (h->has_navmode) { if (a.A == 4) T2D_LAUNCH_COOP2(KIND, true, 4, true); else T2D_LAUNCH_COOP2(KIND, true, 8, true); } \
        else if (a.A == 4) { if (h->has_ram) T2D_LAUNCH_COOP2(KIND, true, 4, false); else T2D_LAUNCH_COOP2(KIND, false, 4, false); } \
        else { if (h->has_ram) T2D_LAUNCH_COOP2(KIND, true, 8, false); else T2D_LAUNCH_COOP2(KIND, false, 8, false); } \
    } while (0)
    if (kind == OBS_U8) T2D_LAUNCH_COOP(OBS_U8);
    else if (kind == OBS_F32_VEC4) T2D_LAUNCH_COOP(OBS_F32_VEC4);
    else T2D_LAUNCH_COOP(OBS_F32_SCALAR);
#undef T2D_LAUNCH_COOP2
#undef T2D_LAUNCH_COOP
    HIP_TRY(hipGetLastError());
    if (h->s.auto_reset) return window_end(h, st);
    return T2D_OK;
}

extern "C" int t2d_step_random(t2d_handle *h, int steps, uint64_t action_seed, float *obs_dev, float *rew_dev,
                               uint8_t *done_dev, void *stream)
{
    if (!h) return fail(T2D_ERR_INVALID, "t2d_step_random: null handle");
    if (!rew_dev || !done_dev || steps < 0) return fail(T2D_ERR_INVALID, "t2d_step_random: bad argument");
    if (!h->reset_done || (h->s.auto_reset && !h->primed)) return fail(T2D_ERR_STATE, "t2d_step_random: reset first");
    DeviceGuard guard(h->device);
    for (int i = 0; i < steps; i++) {
        int rc = step_impl<true>(h, (hipStream_t)stream, nullptr, nullptr, 0, obs_dev, rew_dev, done_dev,
                                 (uint32_t)action_seed, (uint32_t)(action_seed >> 32), h->random_step++);
        if (rc) return rc;
    }
    return T2D_OK;
}

extern "C" int t2d_rollout_random(t2d_handle *h, int steps, uint64_t action_seed, float *obs_dev, float *rew_dev,
                                  uint8_t *done_dev, void *stream)
{
    if (!h) return fail(T2D_ERR_INVALID, "t2d_rollout_random: null handle");
    if (!rew_dev || !done_dev || steps < 0) return fail(T2D_ERR_INVALID, "t2d_rollout_random: bad argument");
    if (!h->reset_done || (h->s.auto_reset && !h->primed)) return fail(T2D_ERR_STATE, "t2d_rollout_random: reset first");
    if (h->np_inter) return fail(T2D_ERR_INVALID, "t2d_rollout_random: random-action rollouts are not defined for numpy-stream Ram handles");
    DeviceGuard guard(h->device);
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)h->s.n, obs_elems = h->s.obs_full ? (size_t)2 * h->s.obs_side * h->s.obs_side : (size_t)kObsPerEnv;
    int done_steps = 0;
    while (done_steps < steps) {
        float *o = obs_dev ? obs_dev + (size_t)done_steps * n * obs_elems : nullptr;
        float *r = rew_dev + (size_t)done_steps * n * 2;
        uint8_t *d = done_dev + (size_t)done_steps * n;
        if (h->has_nav) { // Nav re-planning writes a direction field other lanes re-read: one step per launch
            int rc = step_impl<true>(h, st, nullptr, nullptr, 0, o, r, d, (uint32_t)action_seed,
                                     (uint32_t)(action_seed >> 32), h->random_step++);
            if (rc) return rc;
            done_steps++;
            continue;
        }
        // up to the end of the stamp window (at most one episode switch per env in between)
        int chunk = steps - done_steps;
        if (h->s.auto_reset) {
            const int left = (int)(h->win - h->phase % h->win);
            if (chunk > left) chunk = left;
            if (chunk > 10) chunk = 10;      // (one episode switch per env and launch at most: an episode lasts >= 11 steps)
            int rc = window_begin(h, st);
            if (rc) return rc;
        }
        if (use_step2(h))
            launch_step2<true, true>(h, st, nullptr, nullptr, 0, o, false, r, d, (uint32_t)action_seed,
                                     (uint32_t)(action_seed >> 32), h->random_step, h->phase + 1u, chunk);
        else
            hipLaunchKernelGGL((k_env<OP_STEP, true, false, true>), env_grid(h->s.n), dim3(256), 0, st, h->s, nullptr,
                               nullptr, 0, nullptr, o, r, d, (uint32_t)action_seed, (uint32_t)(action_seed >> 32),
                               h->random_step, h->phase + 1u, chunk);
        HIP_TRY(hipGetLastError());
        h->random_step += (uint32_t)chunk;
        done_steps += chunk;
        if (h->s.auto_reset) {
            h->phase += (uint32_t)chunk;
            int rc = window_end(h, st);
            if (rc) return rc;
        }
    }
    return T2D_OK;
}

extern "C" int t2d_observe(t2d_handle *h, float *obs_dev, void *stream)
{
    if (!h || !obs_dev) return fail(T2D_ERR_INVALID, "t2d_observe: null argument");
    DeviceGuard guard(h->device);
    launch_env<OP_OBSERVE, false>(h, (hipStream_t)stream, nullptr, nullptr, 0, nullptr, obs_dev, nullptr, nullptr, 0u, 0u,
                                  0u, 0u);
    HIP_TRY(hipGetLastError());
    return T2D_OK;
}

// Host-side accessors first refill pending slots, then synchronise (they are test/evaluator calls).
static int quiesce(t2d_handle *h, hipStream_t st)
{
    int rc = flush_impl(h, st);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(st));
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
    if (h->s.obs_full && side != h->s.obs_side)
        return fail(T2D_ERR_INVALID, "t2d_inject: obs_type Full handle is laid out for side %d", h->s.obs_side);
    if (!maze_host || !pos_host) return fail(T2D_ERR_INVALID, "t2d_inject: null buffer");
    if (count == 0) return T2D_OK;
    DeviceGuard guard(h->device);
    if ((rc = quiesce(h, (hipStream_t)stream))) return rc;
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
        // RPF: the fixed tracker spawn may be a wall in the env's own map (track_1v1.py:233-236), so this is legal there
        if (!h->has_rpf && (m[p[0] * side + p[1]] || m[p[2] * side + p[3]]))
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
    HIP_TRY(hipMemcpyAsync(s.nav2 + first, zero.data(), nb, hipMemcpyHostToDevice, st));   // RPF: re-plan to patrol cell 1
    if (s.p_state)
        for (int k3 = 0; k3 < 3; k3++)
            HIP_TRY(hipMemcpyAsync(s.p_state + (size_t)k3 * s.n + first, zero.data(), nb, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (h->has_episode.size() != (size_t)s.n) { h->has_episode.assign((size_t)s.n, 0); h->n_has_episode = 0; }
    for (int i = 0; i < count; i++)
        if (!h->has_episode[(size_t)(first + i)]) { h->has_episode[(size_t)(first + i)] = 1; h->n_has_episode++; }
    if (h->n_has_episode == s.n) h->reset_done = true;      // (all at once or env by env)
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
    if ((rc = quiesce(h, (hipStream_t)stream))) return rc;
    HIP_TRY(hipMemcpyAsync(h->s.plan + env, &p, sizeof(p), hipMemcpyHostToDevice, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return T2D_OK;
}

extern "C" int t2d_inject_nav_goal(t2d_handle *h, int env, int goal_r, int goal_c, void *stream)
{
    int rc = check_range(h, env, 1, "t2d_inject_nav_goal");
    if (rc) return rc;
    if (!h->has_navmode) return fail(T2D_ERR_INVALID, "t2d_inject_nav_goal: the handle has no Nav target");
    if (goal_r < 0 || goal_c < 0 || goal_r >= T2D_MAX_SIDE || goal_c >= T2D_MAX_SIDE)
        return fail(T2D_ERR_INVALID, "t2d_inject_nav_goal: goal (%d, %d)", goal_r, goal_c);
    DeviceGuard guard(h->device);
    hipStream_t st = (hipStream_t)stream;
    if ((rc = quiesce(h, st))) return rc;
    hipLaunchKernelGGL(k_nav_inject_goal, dim3(1), dim3(64), 0, st, h->s, env, (uint32_t)goal_r | ((uint32_t)goal_c << 8));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
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
    if ((rc = quiesce(h, st))) return rc;
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
    if ((rc = quiesce(h, st))) return rc;
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
    if ((rc = quiesce(h, st))) return rc;
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

#if T2D_EXP == 6 || T2D_EXP == 9
extern "C" int t2d_debug_tile_words(t2d_handle *h, uint32_t *out_host /* [n][256] */)
{
    DeviceGuard guard(h->device);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out_host, T2D_EXP == 9 ? h->s.n_maps : h->s.maps, (size_t)h->s.n * kTileWords * sizeof(uint32_t),
                      hipMemcpyDeviceToHost));
    return T2D_OK;
}
#endif

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
