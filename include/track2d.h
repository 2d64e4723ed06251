/*
 * track2d.h — C ABI of libtrack2d_hip.so: the MI355X-native, batched replacement for the reference's
 * gym-track2d environment step/reset loop.
 *
 * The reference has no FFI: its seam is the Python gym protocol. Each entry point below names the
 * reference interface it replaces (paths relative to the reference repo root,
 * G/ = envs/gym-track2d/gym_track2d/). The reference-side binding a maintainer would add is the ctypes
 * stub shown in INTEGRATION.md (and shipped as active_tracking_rl_amd/vec_env.py).
 *
 * Conventions
 *   - every function returns T2D_OK (0) or a negative t2d_status; nothing throws across the ABI;
 *     t2d_last_error() returns a thread-local message for the last failure;
 *   - "dev" pointers are HIP device pointers owned by the caller (PyTorch tensors in practice);
 *     "host" pointers are ordinary host memory; `stream` is a hipStream_t passed as void* (NULL = the
 *     default stream). No entry point synchronises the host except where stated;
 *   - the library owns the batched env state (SoA arrays + 1 KiB bit-packed map tiles in HBM);
 *   - one handle per GPU/process; calls on one handle must not overlap in time.
 */
#ifndef TRACK2D_H
#define TRACK2D_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define T2D_ABI_VERSION 1

typedef enum {
    T2D_OK = 0,
    T2D_ERR_INVALID = -1,   /* bad argument */
    T2D_ERR_HIP = -2,       /* HIP runtime error (message has hipGetErrorString) */
    T2D_ERR_NO_DEVICE = -3, /* no usable gfx950 device */
    T2D_ERR_STATE = -4      /* call not valid in this state (e.g. step before reset) */
} t2d_status;

/* map_type / target_mode values: the kwargs of the registry, G/__init__.py:3-18 */
enum { T2D_MAP_BLOCK = 0, T2D_MAP_MAZE = 1, T2D_MAP_EMPTY = 2 };
enum { T2D_TGT_ADV = 0, T2D_TGT_PZR = 1, T2D_TGT_FAR = 2, T2D_TGT_NAV = 3, T2D_TGT_RAM = 4, T2D_TGT_RPF = 5,
       /* not a registry mode: the target's action always comes from the caller (w_p = 0 like Adv/Nav/Ram/RPF) and the
        * agents may be injected on wall cells, as the reference's RPF spawn can be (track_1v1.py:233-236). Used by the
        * reference-exact mode, whose scripted targets run on the host (include/track2d_np.h). */
       T2D_TGT_EXT = 6 };
/* dtype codes for action arrays */
enum { T2D_ACT_U8 = 0, T2D_ACT_I32 = 1, T2D_ACT_I64 = 2 };
/* obs_type of the registry kwargs (G/__init__.py:11), define_observation at G/envs/track_1v1.py:251-262 */
enum { T2D_OBS_PARTIAL = 0, T2D_OBS_FULL = 1 };
enum { T2D_ACTIONS_VONNEUMANN = 0, T2D_ACTIONS_MOORE = 1 };   /* t2d_config.action_type */

#define T2D_NUM_AGENTS 2
#define T2D_POB 6          /* pob_size, G/envs/track_1v1.py:16 */
#define T2D_WIN 13         /* 2*pob+1 */
#define T2D_OBS_PER_ENV (T2D_NUM_AGENTS * T2D_WIN * T2D_WIN) /* 338 values */
#define T2D_MAX_SIDE 82

typedef struct t2d_handle t2d_handle;

/* Replaces the constructor arguments of Track1v1Env (G/envs/track_1v1.py:14-22) + the registry kwargs
 * + gym's TimeLimit(max_episode_steps=500) (G/__init__.py:17). */
typedef struct {
    uint32_t abi_version;      /* T2D_ABI_VERSION */
    int32_t device;            /* HIP device ordinal */
    int32_t num_envs;          /* envs held by THIS handle (local shard) */
    uint32_t env_id_base;      /* global index of local env 0: random streams are keyed by the global
                                  id, so a sharded batch reproduces the unsharded one */
    uint64_t seed;             /* Philox key */
    int32_t max_episode_steps; /* TimeLimit; 500 in the registry, 0 disables */
    int32_t auto_reset;        /* 1: step() regenerates finished envs in the same launch and returns the
                                     first observation of the next episode (vector-env convention);
                                  0: gym protocol, caller resets (t2d_reset with a mask) */
    uint8_t map_type, target_mode, level;
    uint8_t obs_type;          /* 0 'Partial': obs [N,2,13,13]; 1 'Full': obs [N,2,S,S], S = 82 (Block/Empty) or 81
                                  (Maze), every env of the handle must then have the same S */
    uint8_t action_type;       /* Track1v1Env(action_type=...) (G/envs/track_1v1.py:17,243-249,275-279): 0 'VonNeumann'
                                  (actions 0..3, every registered id), 1 'Moore' (actions 0..7: 4..7 are the diagonals; only
                                  the destination cell is tested). Moore with a scripted target (Ram/Nav/RPF) is refused.
                                  (Occupies former padding: the layout is unchanged and 0 is the old behaviour.) */
    uint8_t reserved_[3];
    /* optional per-env overrides (host pointers, num_envs bytes each, NULL = uniform): BASELINE config 5
     * mixes Block and Maze maps in one batch */
    const uint8_t *map_type_per_env;
    const uint8_t *target_mode_per_env;
    const uint8_t *level_per_env;
} t2d_config;

const char *t2d_last_error(void);
int t2d_abi_version(void);
int t2d_config_size(void);   /* sizeof(t2d_config) as the library was built: a binding checks its struct against it */

/* create_env(env_id, args) / Track1v1Env.__init__ — environment.py:11-32, G/envs/track_1v1.py:14-69 */
int t2d_create(const t2d_config *cfg, t2d_handle **out);
int t2d_destroy(t2d_handle *h);
int t2d_num_envs(const t2d_handle *h);

/* Track1v1Env.reset() (G/envs/track_1v1.py:134-168) for every env whose mask byte is non-zero
 * (mask_dev == NULL: all). Generates map, spawns, goals and the scripted-target plan on the device,
 * zeroes the counters and, if obs_dev != NULL, writes all N observations: f32 [N,2,13,13] (values
 * 0,1,2,4 — what frame_stack's np.float32(obs) yields, environment.py:138); [N,2,S,S] for obs_type Full. */
int t2d_reset(t2d_handle *h, const uint8_t *mask_dev, float *obs_dev, void *stream);

/* Track1v1Env.step(action) wrapped by TimeLimit.step (G/envs/track_1v1.py:71-127) for all N envs in one
 * launch. act_tracker_dev / act_target_dev: [N] arrays of `act_dtype` (values 0..3; the target action is
 * ignored for Ram/Nav modes, as at track_1v1.py:80-84; may be NULL if every env is Ram/Nav).
 * Outputs: obs f32 [N,2,13,13], rewards f32 [N,2] (float64 arithmetic rounded once to f32, i.e.
 * torch.tensor(r).float() at player_util.py:58), done u8 [N]. */
int t2d_step(t2d_handle *h, const void *act_tracker_dev, const void *act_target_dev, int act_dtype,
             float *obs_dev, float *rew_dev, uint8_t *done_dev, void *stream);

/* The same step with the observations left as bytes: obs u8 [N,2,13,13] (values 0,1,2,4), i.e. the step before
 * frame_stack's np.float32(obs) cast (environment.py:138,146); the consumer (the policy's conv stem, perception.py:86-92)
 * decodes them in its first layer. SURVEY 8(d): B_step = 709 B for this variant. Only for 'Partial' observations and
 * handles without Nav/RPF targets; obs_u8_dev must be 4-byte aligned. */
int t2d_step_u8(t2d_handle *h, const void *act_tracker_dev, const void *act_target_dev, int act_dtype,
                uint8_t *obs_u8_dev, float *rew_dev, uint8_t *done_dev, void *stream);

/* With auto_reset, finished envs switch to a pre-generated "next episode" slot inside the step launch; the
 * library refills consumed slots with a generator launch every <= 10 steps on the caller's stream (a slot
 * cannot be needed again sooner: done needs 11 consecutive far steps). t2d_flush runs the generator now for
 * whatever is pending (used before state readback; t2d_reset / t2d_inject / t2d_get_* flush implicitly). */
int t2d_flush(t2d_handle *h, void *stream);

/* Asynchronous generator (off by default). The reference builds the next map inside reset(), on the worker's
 * critical path (G/envs/track_1v1.py:134-168); here the refill of consumed slots can leave the caller's stream
 * altogether: with enable != 0 the step stamps run in two windows of 5 steps, the slots consumed in one window are
 * regenerated by a launch forked onto a library-owned stream after the window's last step, and the caller's stream
 * waits for it (event) one window later, before the first step that could need such a slot again. Results are
 * identical to the in-order generator; only where the generator kernel runs changes. Inside a stream capture
 * (hipGraph) the fork/join becomes a parallel branch of the graph; the capture must end with no fork open:
 * call t2d_generator_join(h, stream) (or t2d_flush) before ending it, and capture a whole number of stamp cycles
 * (t2d_generator_cycle steps) so that replays line up. Switching modes flushes first. */
int t2d_generator_async(t2d_handle *h, int enable, void *stream);
/* Make `stream` wait for every forked generator launch (no generator work is added, the stamps keep running). */
int t2d_generator_join(t2d_handle *h, void *stream);
/* Steps per stamp cycle (10, or min(10, max_episode_steps) rounded to an even number when asynchronous). */
int t2d_generator_cycle(const t2d_handle *h);

/* _get_obs() of the current state without stepping (G/envs/track_1v1.py:287-293). */
int t2d_observe(t2d_handle *h, float *obs_dev, void *stream);

/* Parity/test mode: overwrite envs [first, first+count) with host-supplied maps (u8 side*side each, 0 free
 * / 1 wall, side 81 or 82), positions pos[count][4] = {tracker r,c, target r,c} and goals (may be NULL);
 * counters are zeroed; the scripted-target plan is cleared (a Nav target re-plans at its next step, a Ram target
 * needs t2d_inject_plan). Synchronises the stream. */
int t2d_inject(t2d_handle *h, int first, int count, int side, const uint8_t *maze_host,
               const int32_t *pos_host, const int32_t *goals_host, void *stream);
/* Ram target plan injection for one env: len in [1,10], actions 0..3. Synchronises. */
int t2d_inject_plan(t2d_handle *h, int env, const int32_t *plan_host, int len, int cursor, void *stream);
/* Nav target goal injection for one env (after t2d_inject): the target's next plan leads to (goal_r, goal_c) instead of a
 * cell it would draw itself — Navigator.reset's plan to goal_states[1] (G/envs/navigator.py:43-63) with the reference's
 * goal. Unreachable goals fall back to the target's own re-plan. Synchronises. */
int t2d_inject_nav_goal(t2d_handle *h, int env, int goal_r, int goal_c, void *stream);

/* State readback for tests/evaluators (synchronises). Any output pointer may be NULL.
 * pos/goals: [count][4]; maps: [count][82*82] u8 with row stride 82 (cells outside `side` are 0). */
int t2d_get_state(t2d_handle *h, int first, int count, int32_t *pos_host, int32_t *goals_host,
                  int32_t *c_far_host, int32_t *t_host, uint32_t *episode_host, int32_t *side_host,
                  uint32_t *d2_host, void *stream);
int t2d_get_maps(t2d_handle *h, int first, int count, uint8_t *maps_host, void *stream);
/* Scripted-target state: plan[count][10], len, cursor, navgoal[count][2] (synchronises). */
int t2d_get_target(t2d_handle *h, int first, int count, int32_t *plan_host, int32_t *len_host,
                   int32_t *cursor_host, int32_t *navgoal_host, void *stream);

/* Maze maps grown ahead of the generator pass (csrc/track2d_hip.hip: k_pregrow). RandomMazeGenerator._generate_maze
 * (generators.py:115-145) is one long serial chain per map and, in this library, a pure function of (seed, global env id,
 * episode number, level): it need not run inside the pass that builds a new episode (reset(), track_1v1.py:134-168) around
 * it. t2d_pregrow grows, for every Maze env, the mazes of the episodes the coming passes will build (current + 3, + 4) into a
 * per-env ring the pass then copies from; a pass that does not find its maze there grows it itself: results are the same
 * bit for bit either way. No-op for handles without Maze envs.
 *   T2D_PREGROW_INLINE    launch on `stream`, in order (e.g. a stream that runs beside the one that steps the envs: safe
 *                         without any ordering between the two — a ring entry is only re-used once its episode has started)
 *   T2D_PREGROW_FORK      launch on the handle's own stream, forked from `stream` here and joined into the caller's
 *                         stream by the next generator pass, t2d_generator_join or t2d_flush (capturable: join before the
 *                         capture ends)
 *   T2D_PREGROW_AUTO_ON / _OFF   every generator pass forks one behind itself (plain stepping loops) */
#define T2D_PREGROW_INLINE 0
#define T2D_PREGROW_FORK 1
#define T2D_PREGROW_AUTO_ON 2
#define T2D_PREGROW_AUTO_OFF 3
int t2d_pregrow(t2d_handle *h, int mode, void *stream);
/* {mazes the passes took from the ring, mazes the passes grew themselves, mazes k_pregrow grew, entries it left}. Synchronises. */
int t2d_pregrow_stats(t2d_handle *h, uint32_t stats_host[4], void *stream);

/* Sticky device-side fault word (bit 0: an action outside 0..3 was seen and masked). Synchronises. */
int t2d_get_faults(t2d_handle *h, uint32_t *faults_host, void *stream);

/* Benchmark support: T random-policy steps (on-device Philox actions, auto-reset) in ONE launch per step
 * with no host work in between; writes the last step's outputs. */
int t2d_step_random(t2d_handle *h, int steps, uint64_t action_seed, float *obs_dev, float *rew_dev,
                    uint8_t *done_dev, void *stream);

/* The same random-policy steps as t2d_step_random, but up to 10 consecutive steps per LAUNCH (the env's state stays in
 * registers and its map tile in LDS between steps; a generator pass runs between launches exactly where the
 * per-step path runs it), and EVERY step's outputs are kept: obs_dev [steps,N,2,13,13] (or NULL), rew_dev
 * [steps,N,2], done_dev [steps,N]. Bit-identical to `steps` calls of t2d_step_random(h, 1, ...). This is the
 * "persistent T-step" env-only mode of SURVEY.md 8(d)(ii); handles with Nav targets fall back to one step per launch. */
int t2d_rollout_random(t2d_handle *h, int steps, uint64_t action_seed, float *obs_dev, float *rew_dev,
                       uint8_t *done_dev, void *stream);

/* Pure helper exposed for the exhaustive reward parity test: rewards for n squared distances. */
int t2d_reward_table(t2d_handle *h, const uint32_t *d2_dev, int n, double w_p, float *r_track_dev,
                     float *r_target_dev, void *stream);

#ifdef __cplusplus
}
#endif
#endif
