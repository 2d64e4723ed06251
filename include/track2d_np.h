/*
 * track2d_np.h — C ABI of the reference-exact episode source: the part of the Track2D env that draws random numbers
 * (maps, spawns, goals, the scripted Ram / Nav / RPF targets), restated on the HOST with numpy's legacy global
 * stream (MT19937, RandomState.random_sample / randint / choice / permutation) and a heapq-faithful A*, so that an env
 * seeded like the reference (np.random.seed(s)) produces the reference's episodes draw for draw. The device keeps
 * doing the per-step work (moves, rewards, done, observations): what this library generates goes in through
 * t2d_inject (include/track2d.h) and the scripted target's action goes in as the target action of t2d_step.
 * SURVEY.md section 8(f) rank 3. Host-only code (active_tracking_rl_amd/csrc/np_mode.cpp); no device memory is touched.
 *
 * Reference lines restated (G/ = envs/gym-track2d/gym_track2d/): G/envs/track_1v1.py:134-168,218-240 (reset,
 * init_maze), G/envs/generators.py:12-94,115-176 (static_goals, sample_state, sample_goal, sample_close_states,
 * get_around, the two map generators), G/envs/navigator.py:5-93 (Navigator, RamAgent), G/envs/Astar_solver.py:42-173.
 * numpy (not vendored; legacy stream frozen by NEP 19): mt19937 seeding by an integer, random_sample = (a >> 5,
 * b >> 6) / 2^53, bounded integers by masked rejection on 32-bit words, shuffle = Fisher-Yates from the top,
 * choice(n, k, replace=False) = permutation(n)[:k].
 */
#ifndef TRACK2D_NP_H
#define TRACK2D_NP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct t2d_np t2d_np;

/* One env's host-side state: the kwargs of the registry entry (G/__init__.py:3-18; codes as in track2d.h:
 * map_type 0 Block / 1 Maze / 2 Empty, target_mode 0 Adv / 1 PZR / 2 Far / 3 Nav / 4 Ram / 5 RPF) and its own
 * MT19937 stream seeded like np.random.seed(seed). Returns 0 or a negative code (t2d_np_last_error has the text). */
int t2d_np_create(int map_type, int target_mode, int level, uint32_t seed, t2d_np **out);
int t2d_np_destroy(t2d_np *e);
const char *t2d_np_last_error(void);
/* np.random.seed(seed) — restarts the stream (the reference's env.seed() has no effect on it, track_1v1.py:129-132). */
int t2d_np_seed(t2d_np *e, uint32_t seed);

/* Track1v1Env.reset() up to (not including) the observation: init_maze (new map, goals, spawns, the goal_test loop),
 * then the scripted target's reset (Ram plan / Navigator A* plan with its retries and plan B).
 * maze: u8 [82*82] row-major with row stride 82, the ENV's map (track_1v1.py:233; for RPF it keeps the walls the
 * generator cleared on the patrol cells); *side = 81 (Maze) or 82; pos = tracker r, c, target r, c (init_states);
 * goals = goal_states flattened. Feed them to t2d_inject. */
int t2d_np_reset(t2d_np *e, uint8_t *maze, int32_t *side, int32_t pos[4], int32_t goals[4]);

/* The scripted target's action for the next env step — RamAgent.step() (navigator.py:77-88) or Navigator.step(
 * old_state[1], maze_generator, None) (navigator.py:11-41), including the draws it consumes — and the host-side mirror
 * of the target's own position advanced by it (_next_state on the env's map, track_1v1.py:271-285: the target's motion
 * does not depend on the tracker). Error for the policy-driven modes (Adv / PZR / Far). */
int t2d_np_target_action(t2d_np *e, int32_t *action);

/* The MT19937 state np.random.seed(seed) leaves behind (624 words, then the read position), for t2d_np_attach. */
int t2d_np_mt_state(uint32_t seed, uint32_t out[625]);
/* DEVICE-side reference-exact episodes (csrc/track2d_hip.hip: k_gen_np). After this call the handle's generator — t2d_reset,
 * the in-launch auto-reset's pre-generated episodes — draws env i's maps, goals and spawns from ITS OWN numpy-legacy stream
 * (states_host [N][625]: t2d_np_mt_state(seed_i)), restating init_maze / RandomBlockMazeGenerator / RandomMazeGenerator /
 * sample_goal / sample_close_states / get_around (track_1v1.py:218-240; generators.py:38-94,115-176) draw for draw — whole
 * Fisher-Yates permutations included — on one wavefront per env: episode k of env i is the k-th reset() of the reference
 * env after np.random.seed(seed_i), at batch scale, with no host in the loop. For target modes that draw nothing themselves
 * (Adv, PZR, Far; Ext = host-driven), any map type, 'Partial' or 'Full' observations; call it before the first t2d_reset.
 * Returns 0 or a T2D_ERR_* code (t2d_last_error).
 *
 * T2D_TGT_RAM envs (round 6; RamAgent, G/envs/navigator.py:73-93) are accepted on handles created with auto_reset = 0. A Ram
 * target draws from the stream BETWEEN resets (the coin, the overriding action and the run length / the fresh plan, whenever
 * its plan runs out: navigator.py:80-86), so its next episode cannot be generated ahead of time. For such a handle
 *   - t2d_reset(mask) draws the episode of every env it restarts AT THAT MOMENT (init_maze, then RamAgent.reset's
 *     randint(1, 10) + choice(4, n): track_1v1.py:134-144, navigator.py:90-93) — restart finished envs with mask = the step's
 *     done flags, as the reference's worker loop does (train.py:73-74);
 *   - t2d_step / t2d_step_u8 first run RamAgent.step() for every Ram env from its stream (csrc k_ram_np) and hand the env
 *     step that action as the target's (track_1v1.py:80-82); the caller's target action is used for the other envs only;
 *   - the policy-fused step (atr_act_env_step, atr_coop_env_step) and t2d_rollout_random refuse the handle.
 * t2d_get_target reports the Ram plan as for the Philox generators.
 * T2D_TGT_NAV envs (round 6; Navigator + AstarSolver, G/envs/navigator.py:5-70, G/envs/Astar_solver.py:42-173) likewise, under
 * the same conditions: Navigator.reset's plan is made inside t2d_reset after init_maze, Navigator.step runs ahead of every
 * step launch — when the plan is used up: sample_goal(1) (a whole permutation of the free cells), then the planning loop (A*,
 * up to six failures with fresh goals, plan B = ten random actions). The search is the reference's, accident for accident:
 * heap entries [f, node] compared like Python lists (f = path cost + float64 Euclidean distance, ties by the smaller path
 * cost), heapq's exact sift order, children in action order with wall bumps skipped as explored, and the inverted replace
 * test — one lane per search, its arrays in a per-env scratch block (fault bit 4 if a search outgrows it). So the device's
 * Nav target walks the reference's own paths, not merely paths of the same length. T2D_TGT_RPF envs too: the same Navigator with
 * static goals (generators.py:12-19,48-50,68: the patrol index advances instead of a draw, the tracker spawns on the first
 * patrol cell) planning on the GENERATOR's map — the four patrol cells freed — while the handle's map, the env's own
 * (track_1v1.py:233-236), keeps whatever walls they had: a planned step into such a wall bumps, exactly as in the reference. */
struct t2d_handle;
int t2d_np_attach(struct t2d_handle *h, const uint32_t *states_host);
/* info['distance']^2 of each env's last TERMINAL step (track_1v1.py:118) on a handle with attached streams: with the in-launch
 * auto-reset the handle's d2 already belongs to the next episode when the caller sees done = 1. Entry i = env first + i; only
 * meaningful for envs that have finished an episode. */
int t2d_np_terminal_d2(struct t2d_handle *h, int first, int count, uint32_t *d2_host, void *stream);

/* The two per-step / per-episode calls for MANY envs (a batch replayed against the reference: each env keeps its own stream,
 * so the calls are independent and are spread over `threads` host threads; 0 = one per hardware thread). envs[i] -> entry i of
 * every output: mazes [count][82*82], sides [count], pos / goals [count][4]; actions [count]. Returns 0 or the first failing
 * env's code (t2d_np_last_error names it). */
int t2d_np_reset_many(t2d_np *const *envs, int count, uint8_t *mazes, int32_t *sides, int32_t *pos, int32_t *goals, int threads);
int t2d_np_target_actions(t2d_np *const *envs, int count, int32_t *actions, int threads);

/* The target's current plan (plan_actions, a_i) and, for Nav / RPF, its goal: for tests against the reference's
 * plan0 / navgoal0. plan: caller buffer of max_len ints; *len = full length (may exceed max_len). */
int t2d_np_get_plan(const t2d_np *e, int32_t *plan, int32_t max_len, int32_t *len, int32_t *cursor, int32_t navgoal[2]);

/* AstarSolver(start, [0,1,2,3], maze, goal) (Astar_solver.py:86-173) alone: maze u8 [side*side] row-major (non-zero =
 * wall). *solvable = solution_node is not None; actions = get_actions() (caller buffer of max_len ints, *n = length). */
int t2d_np_astar(const uint8_t *maze, int32_t side, const int32_t start[2], const int32_t goal[2], int32_t *actions,
                 int32_t max_len, int32_t *n, int32_t *solvable);

/* t2d_np_astar's search run by the DEVICE code (csrc/track2d_hip.hip astar_np, one wavefront on GPU `device`): same arguments and
 * results; for the known-answer tests that hold the HIP restatement to the reference's recorded searches. T2D_ERR_* codes. */
int t2d_np_astar_device(int device, const uint8_t *maze, int32_t side, const int32_t start[2], const int32_t goal[2],
                        int32_t *actions, int32_t max_len, int32_t *n, int32_t *solvable);

/* Primitives of the stream, exposed for the known-answer tests against the installed numpy. */
int t2d_np_draw(t2d_np *e, int kind, uint32_t arg, uint32_t count, double *out);
/* kind 0: random_sample() x count; 1: randint(0, arg) x count; 2: permutation(arg) (count ignored, arg values out). */

#ifdef __cplusplus
}
#endif
#endif
