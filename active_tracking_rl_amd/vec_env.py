"""VecTrack2D — batched Track2D environment on one MI355X, a thin ctypes binding of the C ABI in
include/track2d.h (libtrack2d_hip.so, hand-written HIP for gfx950).

This is the reference-side binding shown in INTEGRATION.md: PyTorch only provides device memory and the
current HIP stream. There is NO CPU fallback: if the extension is missing or no GPU is visible this module
raises (a silent fallback would void every parity claim).

Replaces, for N envs at once: Track1v1Env.reset/step (envs/gym-track2d/gym_track2d/envs/track_1v1.py:71-168),
gym's TimeLimit(500) (gym_track2d/__init__.py:17) and frame_stack's float32 cast (environment.py:128-156).
"""
import ctypes as C
import os

import numpy as np
import torch

from . import registry

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("T2D_LIB_PATH") or os.path.join(_HERE, "libtrack2d_hip.so")   # override: kernel experiments
ABI_VERSION = 1

ACT_U8, ACT_I32, ACT_I64 = 0, 1, 2
_ACT_DTYPE = {torch.uint8: ACT_U8, torch.int32: ACT_I32, torch.int64: ACT_I64}


class T2DError(RuntimeError):
    pass


class _Config(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32), ("device", C.c_int32), ("num_envs", C.c_int32), ("env_id_base", C.c_uint32),
        ("seed", C.c_uint64), ("max_episode_steps", C.c_int32), ("auto_reset", C.c_int32),
        ("map_type", C.c_uint8), ("target_mode", C.c_uint8), ("level", C.c_uint8), ("obs_type", C.c_uint8),
        ("action_type", C.c_uint8), ("reserved_", C.c_uint8 * 3),
        ("map_type_per_env", C.c_void_p), ("target_mode_per_env", C.c_void_p), ("level_per_env", C.c_void_p),
    ]


_lib = None

# every symbol include/track2d.h declares (tests check the library exports all of them)
ABI_SYMBOLS = (
    "t2d_last_error", "t2d_abi_version", "t2d_config_size", "t2d_create", "t2d_destroy", "t2d_num_envs", "t2d_reset", "t2d_step",
    "t2d_observe", "t2d_inject", "t2d_inject_plan", "t2d_inject_nav_goal", "t2d_get_state", "t2d_get_maps", "t2d_get_target",
    "t2d_get_faults", "t2d_step_u8", "t2d_step_random", "t2d_rollout_random", "t2d_reward_table", "t2d_flush",
    "t2d_generator_async", "t2d_generator_join", "t2d_generator_cycle", "t2d_pregrow", "t2d_pregrow_stats",
)


def load_library():
    """dlopen libtrack2d_hip.so and declare the prototypes. Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise T2DError("HIP extension %s is missing — run `python -m active_tracking_rl_amd.build` "
                       "(there is no CPU fallback)" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i32, u64 = C.c_void_p, C.c_int, C.c_uint64
    L.t2d_last_error.restype = C.c_char_p
    L.t2d_abi_version.restype = i32
    L.t2d_create.restype = i32
    L.t2d_create.argtypes = [C.POINTER(_Config), C.POINTER(vp)]
    L.t2d_destroy.restype = i32
    L.t2d_destroy.argtypes = [vp]
    L.t2d_num_envs.restype = i32
    L.t2d_num_envs.argtypes = [vp]
    L.t2d_reset.restype = i32
    L.t2d_reset.argtypes = [vp, vp, vp, vp]
    L.t2d_step.restype = i32
    L.t2d_step.argtypes = [vp, vp, vp, i32, vp, vp, vp, vp]
    L.t2d_step_u8.restype = i32
    L.t2d_step_u8.argtypes = [vp, vp, vp, i32, vp, vp, vp, vp]
    L.t2d_observe.restype = i32
    L.t2d_observe.argtypes = [vp, vp, vp]
    L.t2d_flush.restype = i32
    L.t2d_flush.argtypes = [vp, vp]
    L.t2d_generator_async.restype = i32
    L.t2d_generator_async.argtypes = [vp, i32, vp]
    L.t2d_generator_join.restype = i32
    L.t2d_generator_join.argtypes = [vp, vp]
    L.t2d_generator_cycle.restype = i32
    L.t2d_generator_cycle.argtypes = [vp]
    L.t2d_config_size.restype = i32
    if L.t2d_config_size() != C.sizeof(_Config):
        raise T2DError("t2d_config is %d bytes in the library, %d in this binding" % (L.t2d_config_size(), C.sizeof(_Config)))
    L.t2d_inject.restype = i32
    L.t2d_inject.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp]
    L.t2d_inject_plan.restype = i32
    L.t2d_inject_plan.argtypes = [vp, i32, vp, i32, i32, vp]
    L.t2d_get_state.restype = i32
    L.t2d_get_state.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    L.t2d_get_maps.restype = i32
    L.t2d_get_maps.argtypes = [vp, i32, i32, vp, vp]
    L.t2d_get_target.restype = i32
    L.t2d_get_target.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp]
    L.t2d_get_faults.restype = i32
    L.t2d_get_faults.argtypes = [vp, vp, vp]
    L.t2d_pregrow.restype = i32
    L.t2d_pregrow.argtypes = [vp, i32, vp]
    L.t2d_pregrow_stats.restype = i32
    L.t2d_pregrow_stats.argtypes = [vp, vp, vp]
    L.t2d_np_terminal_d2.restype = i32
    L.t2d_np_terminal_d2.argtypes = [vp, i32, i32, vp, vp]
    L.t2d_step_random.restype = i32
    L.t2d_step_random.argtypes = [vp, i32, u64, vp, vp, vp, vp]
    L.t2d_rollout_random.restype = i32
    L.t2d_rollout_random.argtypes = [vp, i32, u64, vp, vp, vp, vp]
    L.t2d_reward_table.restype = i32
    L.t2d_reward_table.argtypes = [vp, vp, i32, C.c_double, vp, vp, vp]
    if L.t2d_abi_version() != ABI_VERSION:
        raise T2DError("libtrack2d_hip.so ABI %d != binding ABI %d" % (L.t2d_abi_version(), ABI_VERSION))
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise T2DError("track2d error %d: %s" % (rc, load_library().t2d_last_error().decode()))


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class VecTrack2D(object):
    """N independent Track2D envs stepped by one kernel launch.

    reset() -> obs f32 [N,2,13,13]           (agent 0 = tracker, agent 1 = target)
    step(a_tracker[N], a_target[N]) -> (obs f32 [N,2,13,13], rewards f32 [N,2], done uint8 [N])
    With auto_reset=True finished envs restart inside the same launch and `obs` holds the first observation
    of their next episode (the reference worker discards the terminal observation too: train.py:73-74).
    """

    def __init__(self, env_id=None, num_envs=1, device="cuda:0", seed=1, env_id_base=0, auto_reset=True,
                 map_type=None, target_mode=None, level=0, max_episode_steps=None,
                 map_type_per_env=None, target_mode_per_env=None, level_per_env=None, obs_type="Partial",
                 async_gen=False, action_type="VonNeumann"):
        if not torch.cuda.is_available():
            raise T2DError("VecTrack2D needs an MI355X visible to PyTorch-ROCm (no CPU fallback)")
        self.L = load_library()
        if env_id is not None:
            sp = registry.spec(env_id)
            map_type, target_mode, level = sp["map_type"], sp["target_mode"], sp["level"]
            obs_type = sp["obs_type"]
            if max_episode_steps is None:
                max_episode_steps = sp["max_episode_steps"]
        if max_episode_steps is None:
            max_episode_steps = registry.MAX_EPISODE_STEPS
        self.env_id = env_id
        self.device = torch.device(device)
        self.num_envs = int(num_envs)
        self.map_type, self.target_mode, self.level = map_type, target_mode, int(level)
        cfg = _Config()
        cfg.abi_version = ABI_VERSION
        cfg.device = self.device.index if self.device.index is not None else torch.cuda.current_device()
        cfg.num_envs = self.num_envs
        cfg.env_id_base = int(env_id_base)
        cfg.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        cfg.max_episode_steps = int(max_episode_steps)
        cfg.auto_reset = 1 if auto_reset else 0
        cfg.map_type = registry.MAP_CODE[map_type]
        cfg.target_mode = registry.TARGET_CODE[target_mode]
        cfg.level = int(level)
        cfg.obs_type = 1 if obs_type == "Full" else 0
        if action_type not in ("VonNeumann", "Moore"):      # track_1v1.py:243-249
            raise TypeError("Action type must be either 'VonNeumann' or 'Moore'")
        cfg.action_type = 1 if action_type == "Moore" else 0
        self.action_type, self.num_actions = action_type, 8 if action_type == "Moore" else 4
        self.obs_type = obs_type
        if obs_type == "Full":   # track_1v1.py:254-256: Box(shape=(1, S, S)); S = 81 for Maze maps, else 82
            all_maze = (map_type == "Maze") if map_type_per_env is None else bool((np.asarray(map_type_per_env) == 1).all())
            self.obs_hw = (81, 81) if all_maze else (82, 82)
        else:
            self.obs_hw = (13, 13)
        self._keep = []
        for name, arr in (("map_type_per_env", map_type_per_env), ("target_mode_per_env", target_mode_per_env),
                          ("level_per_env", level_per_env)):
            if arr is not None:
                a = np.ascontiguousarray(arr, np.uint8)
                assert a.shape == (self.num_envs,)
                self._keep.append(a)
                setattr(cfg, name, a.ctypes.data)
        self.scripted_target = target_mode in ("Ram", "Nav", "RPF") and target_mode_per_env is None
        tm = [registry.TARGET_CODE[target_mode]] if target_mode_per_env is None else list(np.asarray(target_mode_per_env))
        rpf = any(int(m) == registry.TARGET_CODE["RPF"] for m in tm)
        self.supports_u8 = obs_type == "Partial" and not rpf     # t2d_step_u8 / atr_act_env_step (the k_step2 kernel family)
        h = C.c_void_p()
        _check(self.L.t2d_create(C.byref(cfg), C.byref(h)))
        self.h = h
        self.auto_reset = bool(auto_reset)
        self.async_gen = False
        if async_gen and auto_reset and min(10, int(max_episode_steps) or 10) >= 2:
            self.generator_async(True)

    # -- lifecycle ------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "h", None):
            self.L.t2d_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _new_obs(self):
        return torch.empty((self.num_envs, 2) + self.obs_hw, dtype=torch.float32, device=self.device)

    # -- gym-protocol-shaped batched calls ----------------------------------------------------------
    def reset(self, mask=None, out=None):
        obs = out if out is not None else self._new_obs()
        mp = None
        if mask is not None:
            mask = mask.to(device=self.device, dtype=torch.uint8).contiguous()
            assert mask.shape == (self.num_envs,)
            mp = C.c_void_p(mask.data_ptr())
        _check(self.L.t2d_reset(self.h, mp, C.c_void_p(obs.data_ptr()), self._stream()))
        return obs

    def step(self, act_tracker, act_target=None, out=None):
        if out is None:
            obs = self._new_obs()
            rew = torch.empty((self.num_envs, 2), dtype=torch.float32, device=self.device)
            done = torch.empty((self.num_envs,), dtype=torch.uint8, device=self.device)
        else:
            obs, rew, done = out
        a0 = act_tracker.reshape(-1)
        assert a0.is_cuda and a0.is_contiguous() and a0.numel() == self.num_envs and a0.dtype in _ACT_DTYPE
        a1p = None
        if act_target is not None:
            a1 = act_target.reshape(-1)
            assert a1.is_cuda and a1.is_contiguous() and a1.numel() == self.num_envs and a1.dtype == a0.dtype
            a1p = C.c_void_p(a1.data_ptr())
        elif not self.scripted_target:
            raise T2DError("act_target is required unless every env has a scripted (Ram/Nav/RPF) target")
        _check(self.L.t2d_step(self.h, C.c_void_p(a0.data_ptr()), a1p, _ACT_DTYPE[a0.dtype],
                               C.c_void_p(obs.data_ptr()), C.c_void_p(rew.data_ptr()),
                               C.c_void_p(done.data_ptr()), self._stream()))
        return obs, rew, done

    def step_u8(self, act_tracker, act_target=None, out=None):
        """step() with the observations left as bytes: obs u8 [N,2,13,13] (t2d_step_u8)."""
        if out is None:
            out = (torch.empty((self.num_envs, 2) + self.obs_hw, dtype=torch.uint8, device=self.device),
                   torch.empty((self.num_envs, 2), dtype=torch.float32, device=self.device),
                   torch.empty((self.num_envs,), dtype=torch.uint8, device=self.device))
        obs, rew, done = out
        a0 = act_tracker.reshape(-1)
        assert obs.dtype == torch.uint8 and obs.is_contiguous()
        assert a0.is_cuda and a0.is_contiguous() and a0.numel() == self.num_envs and a0.dtype in _ACT_DTYPE
        a1p = None
        if act_target is not None:
            a1 = act_target.reshape(-1)
            assert a1.is_cuda and a1.is_contiguous() and a1.numel() == self.num_envs and a1.dtype == a0.dtype
            a1p = C.c_void_p(a1.data_ptr())
        elif not self.scripted_target:
            raise T2DError("act_target is required unless every env has a scripted (Ram/Nav/RPF) target")
        _check(self.L.t2d_step_u8(self.h, C.c_void_p(a0.data_ptr()), a1p, _ACT_DTYPE[a0.dtype],
                                  C.c_void_p(obs.data_ptr()), C.c_void_p(rew.data_ptr()),
                                  C.c_void_p(done.data_ptr()), self._stream()))
        return obs, rew, done

    def step_random(self, steps, action_seed=1, out=None):
        """`steps` launches with on-device Philox actions (benchmark / soak mode)."""
        if out is None:
            out = (self._new_obs(), torch.empty((self.num_envs, 2), dtype=torch.float32, device=self.device),
                   torch.empty((self.num_envs,), dtype=torch.uint8, device=self.device))
        obs, rew, done = out
        _check(self.L.t2d_step_random(self.h, int(steps), int(action_seed), C.c_void_p(obs.data_ptr()),
                                      C.c_void_p(rew.data_ptr()), C.c_void_p(done.data_ptr()), self._stream()))
        return obs, rew, done

    def rollout_random(self, steps, action_seed=1, out=None, keep_obs=True):
        """`steps` random-policy steps, up to 10 per launch (t2d_rollout_random), keeping every step's outputs:
        obs [steps,N,2,h,w] (None with keep_obs=False), rewards [steps,N,2], done [steps,N]."""
        if out is None:
            h, w = self.obs_hw
            obs = torch.empty((steps, self.num_envs, 2, h, w), dtype=torch.float32, device=self.device) if keep_obs else None
            out = (obs, torch.empty((steps, self.num_envs, 2), dtype=torch.float32, device=self.device),
                   torch.empty((steps, self.num_envs), dtype=torch.uint8, device=self.device))
        obs, rew, done = out
        _check(self.L.t2d_rollout_random(self.h, int(steps), int(action_seed),
                                         C.c_void_p(obs.data_ptr()) if obs is not None else None,
                                         C.c_void_p(rew.data_ptr()), C.c_void_p(done.data_ptr()), self._stream()))
        return obs, rew, done

    def flush(self):
        """Run the generator now for every consumed slot, in order on the current stream, and restart the step stamps
        (see t2d_flush in include/track2d.h)."""
        _check(self.L.t2d_flush(self.h, self._stream()))

    def generator_async(self, enable=True):
        """Refill consumed next-episode slots on a library-owned side stream, overlapped with the following steps
        (t2d_generator_async). Same results as the in-order generator."""
        _check(self.L.t2d_generator_async(self.h, 1 if enable else 0, self._stream()))
        self.async_gen = bool(enable)

    def generator_join(self):
        """Make the current stream wait for every forked generator launch (required before a hipGraph capture ends)."""
        _check(self.L.t2d_generator_join(self.h, self._stream()))

    @property
    def generator_cycle(self):
        return int(self.L.t2d_generator_cycle(self.h))

    def observe(self, out=None):
        obs = out if out is not None else self._new_obs()
        _check(self.L.t2d_observe(self.h, C.c_void_p(obs.data_ptr()), self._stream()))
        return obs

    # -- parity / test hooks -------------------------------------------------------------------------
    def inject(self, mazes, pos, goals=None, first=0):
        mazes = np.ascontiguousarray(mazes, np.uint8)
        if mazes.ndim == 2:
            mazes = mazes[None]
        count, side = mazes.shape[0], mazes.shape[1]
        pos = np.ascontiguousarray(np.asarray(pos, np.int32).reshape(count, 4))
        gp = None
        if goals is not None:
            goals = np.ascontiguousarray(np.asarray(goals, np.int32).reshape(count, 4))
            gp = _np_ptr(goals)
        _check(self.L.t2d_inject(self.h, int(first), count, side, _np_ptr(mazes), _np_ptr(pos), gp, self._stream()))

    def inject_plan(self, env, plan, cursor=0):
        plan = np.ascontiguousarray(plan, np.int32)
        _check(self.L.t2d_inject_plan(self.h, int(env), _np_ptr(plan), len(plan), int(cursor), self._stream()))

    def inject_nav_goal(self, env, goal):
        """Nav target of `env`: plan to `goal` (r, c) at the next step instead of to a self-drawn cell (after inject)."""
        self.L.t2d_inject_nav_goal.restype = C.c_int
        self.L.t2d_inject_nav_goal.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        _check(self.L.t2d_inject_nav_goal(self.h, int(env), int(goal[0]), int(goal[1]), self._stream()))

    def get_state(self, first=0, count=None):
        count = self.num_envs - first if count is None else count
        pos = np.zeros((count, 2, 2), np.int32)
        goals = np.zeros((count, 2, 2), np.int32)
        c_far, t, side = (np.zeros(count, np.int32) for _ in range(3))
        ep, d2 = np.zeros(count, np.uint32), np.zeros(count, np.uint32)
        _check(self.L.t2d_get_state(self.h, first, count, _np_ptr(pos), _np_ptr(goals), _np_ptr(c_far), _np_ptr(t),
                                    _np_ptr(ep), _np_ptr(side), _np_ptr(d2), self._stream()))
        return dict(pos=pos, goals=goals, c_far=c_far, t=t, episode=ep, side=side, d2=d2)

    def terminal_d2(self, first=0, count=None):
        """uint32 [count]: squared distance of each env's last TERMINAL step on a handle with numpy streams attached
        (t2d_np_terminal_d2: after an in-launch auto-reset get_state()['d2'] is already the next episode's)."""
        count = self.num_envs - first if count is None else count
        d2 = np.zeros(count, np.uint32)
        _check(self.L.t2d_np_terminal_d2(self.h, first, count, _np_ptr(d2), self._stream()))
        return d2

    def get_maps(self, first=0, count=None):
        """u8 [count, 82, 82]; cells outside an env's side x side square are 0."""
        count = self.num_envs - first if count is None else count
        maps = np.zeros((count, 82, 82), np.uint8)
        _check(self.L.t2d_get_maps(self.h, first, count, _np_ptr(maps), self._stream()))
        return maps

    def get_target(self, first=0, count=None):
        count = self.num_envs - first if count is None else count
        plan = np.zeros((count, 10), np.int32)
        ln, cur = np.zeros(count, np.int32), np.zeros(count, np.int32)
        ng = np.zeros((count, 2), np.int32)
        _check(self.L.t2d_get_target(self.h, first, count, _np_ptr(plan), _np_ptr(ln), _np_ptr(cur), _np_ptr(ng),
                                     self._stream()))
        return dict(plan=plan, len=ln, cursor=cur, navgoal=ng)

    PREGROW_INLINE, PREGROW_FORK, PREGROW_AUTO_ON, PREGROW_AUTO_OFF = 0, 1, 2, 3

    def pregrow(self, fork=True):
        """Grow the Maze maps the coming generator passes will need ahead of them (t2d_pregrow, csrc k_pregrow: a maze is a pure
        function of seed, env id and episode number, and its growth is the long serial chain of a generated episode). fork: on
        the handle's own stream, joined by the next pass / generator_join(); else in order on the current stream (meant for a
        stream that runs BESIDE the one stepping the envs). No-op without Maze envs; same episodes bit for bit either way.
        (T2D_PREGROW=0 in the environment turns the call into a no-op: A/B runs.)"""
        if os.environ.get("T2D_PREGROW", "1") == "0":
            return
        _check(self.L.t2d_pregrow(self.h, self.PREGROW_FORK if fork else self.PREGROW_INLINE, self._stream()))

    def pregrow_auto(self, enable=True):
        """Every generator pass forks a pregrow() behind itself (plain stepping loops: the growth runs under the next steps)."""
        _check(self.L.t2d_pregrow(self.h, self.PREGROW_AUTO_ON if enable else self.PREGROW_AUTO_OFF, self._stream()))

    def pregrow_stats(self):
        """dict(taken, grown_in_pass, pregrown, left): mazes the passes copied from the ring / grew themselves, mazes k_pregrow
        grew, ring entries it found in order."""
        f = np.zeros(4, np.uint32)
        _check(self.L.t2d_pregrow_stats(self.h, _np_ptr(f), self._stream()))
        return dict(taken=int(f[0]), grown_in_pass=int(f[1]), pregrown=int(f[2]), left=int(f[3]))

    def faults(self):
        f = np.zeros(1, np.uint32)
        _check(self.L.t2d_get_faults(self.h, _np_ptr(f), self._stream()))
        return int(f[0])

    def reward_table(self, d2, w_p):
        d2 = d2.to(device=self.device, dtype=torch.int32).contiguous()
        rt = torch.empty(d2.numel(), dtype=torch.float32, device=self.device)
        rg = torch.empty_like(rt)
        _check(self.L.t2d_reward_table(self.h, C.c_void_p(d2.data_ptr()), d2.numel(), float(w_p),
                                       C.c_void_p(rt.data_ptr()), C.c_void_p(rg.data_ptr()), self._stream()))
        return rt, rg
