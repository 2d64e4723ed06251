/*
 * track2d_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * Scalar CPU restatement (plain C) of the reference's gym-track2d hot path:
 *   Track1v1Env.step/reset            envs/gym-track2d/gym_track2d/envs/track_1v1.py:71-168
 *   init_maze / generators            track_1v1.py:218-240, generators.py:21-176
 *   RamAgent / Navigator / A*         navigator.py:5-93, Astar_solver.py:42-173
 *   gym TimeLimit(max_episode_steps)  gym==0.12.5 (requirements.txt:1; not vendored), registered
 *                                     with max_episode_steps=500 at gym_track2d/__init__.py:17
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product path (active_tracking_rl_amd/) never does.
 *
 * Two random sources drive the same restated algorithms:
 *   ORC_RNG_NP     numpy legacy RandomState (MT19937) — reproduces the reference draw for draw
 *                  when np.random.seed() re-seeding is neutralised; pinned to tests/golden/.
 *   ORC_RNG_PHILOX the device specification: Philox4x32-10 counter streams keyed by
 *                  (seed, global env id, episode, stream). Same sequential algorithms; the two
 *                  O(n) shuffles (generators.py:166 and the free-list picks at :44,:61) are
 *                  replaced by a keyed permutation prefix / direct bounded draws, and heap A*
 *                  by a goal-rooted BFS direction field (equal path length, fixed tie-break).
 *                  The HIP kernels must match this mode bit for bit.
 */
#ifndef TRACK2D_ORACLE_H
#define TRACK2D_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_MAP_BLOCK = 0, ORC_MAP_MAZE = 1, ORC_MAP_EMPTY = 2 };
enum { ORC_TGT_ADV = 0, ORC_TGT_PZR = 1, ORC_TGT_FAR = 2, ORC_TGT_NAV = 3, ORC_TGT_RAM = 4, ORC_TGT_RPF = 5 };
enum { ORC_RNG_NP = 0, ORC_RNG_PHILOX = 1 };

#define ORC_MAX_SIDE 82
#define ORC_POB 6
#define ORC_WIN 13          /* 2*pob+1 */
#define ORC_OBS_CELLS 169   /* 13*13 */

typedef struct orc_env orc_env;

/* rng_mode ORC_RNG_NP: `seed` is the 32-bit np.random.seed(seed) value, env_id ignored.
 * rng_mode ORC_RNG_PHILOX: 64-bit key `seed`, global env index `env_id`. */
orc_env *orc_create(int map_type, int target_mode, int level, int max_steps,
                    int rng_mode, uint64_t seed, uint32_t env_id);
void orc_destroy(orc_env *e);

/* obs_type: 0 = 'Partial' (default; obs u8[2][13][13]), 1 = 'Full' (obs u8[2][side][side], both agents see the
 * whole map with the tracker painted 2 and the target 4 — track_1v1.py:288-290,295-307). orc_obs_size = bytes. */
void orc_set_obs_full(orc_env *e, int full);
/* action_type (track_1v1.py:17,243-249,271-285): 0 'VonNeumann' (default), 1 'Moore' (8 actions, diagonals 4..7) */
void orc_set_action_type(orc_env *e, int moore);
int orc_obs_size(const orc_env *e);

/* Re-seed the numpy-legacy stream (np.random.seed(int)). */
void orc_seed_np(orc_env *e, uint32_t seed);

/* reset(): new map, spawns, goals, scripted-target plan; writes obs[2][13][13] (values 0,1,2,4). */
void orc_reset(orc_env *e, uint8_t *obs);

/* step(actions[2]) -> obs u8[2*169], rewards f64[2], done flag (far-counter OR time limit).
 * For Ram/Nav modes actions[1] is overridden by the scripted target; the action actually
 * applied is returned in applied[2] (may be NULL). Returns 0, or -1 on an invalid action. */
int orc_step(orc_env *e, const int *actions, uint8_t *obs, double *rewards, int *done, int *applied);

/* Parity/test mode: replace map (u8 side*side, 0 free / 1 wall), positions and goals; zero the
 * counters; scripted target plans are left untouched unless reset_target != 0. */
int orc_inject(orc_env *e, int side, const uint8_t *maze, const int *pos /*[2][2]*/,
               const int *goals /*[2][2]*/);
/* Ram plan injection (parity with captured plans): len in [1,10]. */
int orc_inject_plan(orc_env *e, const int *plan, int len, int cursor);

/* State readback. */
int orc_side(const orc_env *e);
void orc_get_maze(const orc_env *e, uint8_t *maze /* side*side */);
void orc_get_state(const orc_env *e, int *pos /*[2][2]*/, int *goals /*[2][2]*/,
                   int *c_far, int *t, int64_t *d2);
void orc_get_obs(const orc_env *e, uint8_t *obs);
int orc_get_plan(const orc_env *e, int *plan /* up to 1024 */, int *cursor);
void orc_get_nav(const orc_env *e, int *nav_goal /*[2]*/, int *planb, int *plan_len);
uint32_t orc_episode(const orc_env *e);

/* n envs in lock step (same obs size each): actions int[n][2] -> obs u8[n][obs_size], rewards f64[n][2], done u8[n];
 * auto_reset != 0: a finished env is reset in the same call and obs holds the first observation of its next episode
 * (what the batched product's in-launch auto-reset returns). Returns 0, or -1 - i for an invalid action of env i. */
int orc_step_batch(orc_env **envs, int n, const int *actions, uint8_t *obs, double *rewards, uint8_t *done, int auto_reset);

/* Pure helpers exposed for unit tests. */
void orc_reward(int64_t d2, double w_p, double *r_track, double *r_target);
/* numpy-legacy primitives on a standalone MT19937 (tests compare with numpy itself). */
typedef struct orc_mt orc_mt;
orc_mt *orc_mt_new(uint32_t seed);
void orc_mt_free(orc_mt *m);
uint32_t orc_mt_u32(orc_mt *m);
double orc_mt_double(orc_mt *m);
uint32_t orc_mt_interval(orc_mt *m, uint32_t max);       /* legacy random_interval  */
void orc_mt_permutation(orc_mt *m, int n, int32_t *out); /* RandomState.permutation */
/* Philox4x32-10 block (device spec). */
void orc_philox4x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                    uint32_t out[4]);
/* Keyed permutation of [0, 6400) used by the PHILOX-mode block generator. */
uint32_t orc_perm6400(const uint32_t rk[8], uint32_t i);
/* heapq-faithful A* (Astar_solver.py:121-149): returns plan length or -1 if unsolvable. */
int orc_astar(int side, const uint8_t *maze, const int *start, const int *goal, int *actions, int cap);
/* BFS direction field (device spec for Nav): dir[side*side] in {0..3, 4=goal, 255=unreachable};
 * dist (may be NULL) int32[side*side], -1 unreachable. */
void orc_bfs_field(int side, const uint8_t *maze, const int *goal, uint8_t *dir, int32_t *dist);

#ifdef __cplusplus
}
#endif
#endif
