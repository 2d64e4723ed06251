"""ctypes binding of oracle/libtrack2d_oracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (active_tracking_rl_amd) never does. See oracle/track2d_oracle.h for what the
library restates (reference file:line per function).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libtrack2d_oracle.so")

MAP = {"Block": 0, "Maze": 1, "Empty": 2}
TGT = {"Adv": 0, "PZR": 1, "Far": 2, "Nav": 3, "Ram": 4, "RPF": 5}
RNG_NP, RNG_PHILOX = 0, 1


def build(force=False):
    src = os.path.join(_HERE, "track2d_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libtrack2d_oracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        vp, i32, u32, u64, f64 = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64, C.c_double
        pi = C.POINTER(C.c_int)
        pu8 = C.POINTER(C.c_uint8)
        L.orc_create.restype = vp
        L.orc_create.argtypes = [i32, i32, i32, i32, i32, u64, u32]
        L.orc_destroy.argtypes = [vp]
        L.orc_seed_np.argtypes = [vp, u32]
        L.orc_set_obs_full.argtypes = [vp, i32]
        L.orc_set_action_type.argtypes = [vp, i32]
        L.orc_obs_size.restype = i32
        L.orc_obs_size.argtypes = [vp]
        L.orc_reset.argtypes = [vp, pu8]
        L.orc_step.restype = i32
        L.orc_step.argtypes = [vp, pi, pu8, C.POINTER(f64), pi, pi]
        L.orc_inject.restype = i32
        L.orc_inject.argtypes = [vp, i32, pu8, pi, pi]
        L.orc_inject_plan.restype = i32
        L.orc_inject_plan.argtypes = [vp, pi, i32, i32]
        L.orc_side.restype = i32
        L.orc_side.argtypes = [vp]
        L.orc_get_maze.argtypes = [vp, pu8]
        L.orc_get_state.argtypes = [vp, pi, pi, pi, pi, C.POINTER(C.c_int64)]
        L.orc_get_obs.argtypes = [vp, pu8]
        L.orc_get_plan.restype = i32
        L.orc_get_plan.argtypes = [vp, pi, pi]
        L.orc_get_nav.argtypes = [vp, pi, pi, pi]
        L.orc_episode.restype = u32
        L.orc_episode.argtypes = [vp]
        L.orc_step_batch.restype = i32
        L.orc_step_batch.argtypes = [C.POINTER(vp), i32, pi, pu8, C.POINTER(f64), pu8, i32]
        L.orc_reward.argtypes = [C.c_int64, f64, C.POINTER(f64), C.POINTER(f64)]
        L.orc_mt_new.restype = vp
        L.orc_mt_new.argtypes = [u32]
        L.orc_mt_free.argtypes = [vp]
        L.orc_mt_u32.restype = u32
        L.orc_mt_u32.argtypes = [vp]
        L.orc_mt_double.restype = f64
        L.orc_mt_double.argtypes = [vp]
        L.orc_mt_interval.restype = u32
        L.orc_mt_interval.argtypes = [vp, u32]
        L.orc_mt_permutation.argtypes = [vp, i32, C.POINTER(C.c_int32)]
        L.orc_philox4x32.argtypes = [u32] * 6 + [C.POINTER(u32)]
        L.orc_perm6400.restype = u32
        L.orc_perm6400.argtypes = [C.POINTER(u32), u32]
        L.orc_astar.restype = i32
        L.orc_astar.argtypes = [i32, pu8, pi, pi, pi, i32]
        L.orc_bfs_field.argtypes = [i32, pu8, pi, pu8, C.POINTER(C.c_int32)]
        _lib = L
    return _lib


def _p(arr, ct):
    return arr.ctypes.data_as(C.POINTER(ct))


class OracleEnv(object):
    """One scalar Track2D env (obs u8[2,13,13], rewards f64[2], done bool)."""

    def __init__(self, map_type="Block", target_mode="PZR", level=0, max_steps=500,
                 rng_mode=RNG_NP, seed=0, env_id=0, obs_type="Partial", action_type="VonNeumann"):
        self.L = lib()
        self.map_type, self.target_mode = map_type, target_mode
        self.h = self.L.orc_create(MAP[map_type], TGT[target_mode], level, max_steps, rng_mode,
                                   int(seed), int(env_id))
        self.full = obs_type == "Full"
        self.L.orc_set_obs_full(self.h, 1 if self.full else 0)
        self.L.orc_set_action_type(self.h, 1 if action_type == "Moore" else 0)
        s = self.L.orc_side(self.h)
        self._obs = np.zeros((2, s, s) if self.full else (2, 13, 13), np.uint8)
        self._rew = np.zeros(2, np.float64)
        self._done = C.c_int(0)
        self._applied = np.zeros(2, np.int32)

    def __del__(self):
        try:
            if self.h:
                self.L.orc_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def seed_np(self, seed):
        self.L.orc_seed_np(self.h, int(seed))

    def reset(self):
        self.L.orc_reset(self.h, _p(self._obs, C.c_uint8))
        return self._obs.copy()

    def step(self, actions):
        act = np.asarray(actions, np.int32).copy()
        rc = self.L.orc_step(self.h, _p(act, C.c_int), _p(self._obs, C.c_uint8), _p(self._rew, C.c_double),
                             C.byref(self._done), _p(self._applied, C.c_int))
        if rc != 0:
            raise ValueError("invalid action %r" % (actions,))
        return self._obs.copy(), self._rew.copy(), bool(self._done.value), self._applied.copy()

    def inject(self, maze, pos, goals=None):
        maze = np.ascontiguousarray(maze, np.uint8)
        pos = np.ascontiguousarray(pos, np.int32)
        g = np.ascontiguousarray(goals if goals is not None else np.zeros((2, 2)), np.int32)
        rc = self.L.orc_inject(self.h, maze.shape[0], _p(maze, C.c_uint8), _p(pos, C.c_int), _p(g, C.c_int))
        if rc != 0:
            raise ValueError("inject failed")

    def inject_plan(self, plan, cursor=0):
        plan = np.ascontiguousarray(plan, np.int32)
        if self.L.orc_inject_plan(self.h, _p(plan, C.c_int), len(plan), cursor) != 0:
            raise ValueError("bad plan")

    @property
    def side(self):
        return self.L.orc_side(self.h)

    @property
    def maze(self):
        s = self.side
        m = np.zeros((s, s), np.uint8)
        self.L.orc_get_maze(self.h, _p(m, C.c_uint8))
        return m

    def state(self):
        pos = np.zeros((2, 2), np.int32)
        goals = np.zeros((2, 2), np.int32)
        c_far, t, d2 = C.c_int(0), C.c_int(0), C.c_int64(0)
        self.L.orc_get_state(self.h, _p(pos, C.c_int), _p(goals, C.c_int), C.byref(c_far), C.byref(t), C.byref(d2))
        return dict(pos=pos, goals=goals, c_far=c_far.value, t=t.value, d2=d2.value)

    def obs(self):
        o = np.zeros_like(self._obs)
        self.L.orc_get_obs(self.h, _p(o, C.c_uint8))
        return o

    def nav(self):
        """(goal [2], plan B?, plan length) of a Nav / RPF target."""
        g = np.zeros(2, np.int32)
        pb, ln = C.c_int(0), C.c_int(0)
        self.L.orc_get_nav(self.h, _p(g, C.c_int), C.byref(pb), C.byref(ln))
        return g, bool(pb.value), ln.value

    def plan(self):
        buf = np.zeros(1024, np.int32)
        cur = C.c_int(0)
        n = self.L.orc_get_plan(self.h, _p(buf, C.c_int), C.byref(cur))
        return buf[:min(n, 1024)].copy(), cur.value


class OracleBatch(object):
    """n scalar oracle envs stepped in lock step by one C call (orc_step_batch) — the checker of the full-size parity
    tests: reset() -> obs u8 [n,2,13,13]; step(actions [n,2]) -> (obs, rewards f64 [n,2], done u8 [n]); with auto_reset a
    finished env restarts inside the call and reports its next episode's first observation, as the product does."""

    def __init__(self, envs, auto_reset=True):
        self.envs, self.n, self.auto_reset = list(envs), len(envs), bool(auto_reset)
        self.L = lib()
        self._h = (C.c_void_p * self.n)(*[e.h for e in self.envs])
        shape = self.envs[0]._obs.shape
        assert all(e._obs.shape == shape for e in self.envs)
        self._obs = np.zeros((self.n,) + shape, np.uint8)
        self._rew = np.zeros((self.n, 2), np.float64)
        self._done = np.zeros(self.n, np.uint8)

    def reset(self):
        for i, e in enumerate(self.envs):
            self._obs[i] = e.reset()
        return self._obs.copy()

    def step(self, actions):
        act = np.ascontiguousarray(actions, np.int32).reshape(self.n, 2)
        rc = self.L.orc_step_batch(self._h, self.n, _p(act, C.c_int), _p(self._obs, C.c_uint8), _p(self._rew, C.c_double),
                                   _p(self._done, C.c_uint8), 1 if self.auto_reset else 0)
        if rc != 0:
            raise ValueError("invalid action for env %d" % (-1 - rc))
        return self._obs.copy(), self._rew.copy(), self._done.copy()


def reward(d2, w_p):
    a, b = C.c_double(0), C.c_double(0)
    lib().orc_reward(int(d2), float(w_p), C.byref(a), C.byref(b))
    return a.value, b.value


def philox4x32(k0, k1, c0, c1, c2, c3):
    out = (C.c_uint32 * 4)()
    lib().orc_philox4x32(k0, k1, c0, c1, c2, c3, out)
    return [int(x) for x in out]


def perm6400(rk, i):
    arr = (C.c_uint32 * 8)(*[int(x) for x in rk])
    return int(lib().orc_perm6400(arr, int(i)))


def astar(maze, start, goal, cap=8192):
    maze = np.ascontiguousarray(maze, np.uint8)
    s = np.asarray(start, np.int32).copy()
    g = np.asarray(goal, np.int32).copy()
    out = np.zeros(cap, np.int32)
    n = lib().orc_astar(maze.shape[0], _p(maze, C.c_uint8), _p(s, C.c_int), _p(g, C.c_int), _p(out, C.c_int), cap)
    return None if n < 0 else out[:n].copy()


def bfs_field(maze, goal):
    maze = np.ascontiguousarray(maze, np.uint8)
    S = maze.shape[0]
    g = np.asarray(goal, np.int32).copy()
    d = np.zeros((S, S), np.uint8)
    dist = np.zeros((S, S), np.int32)
    lib().orc_bfs_field(S, _p(maze, C.c_uint8), _p(g, C.c_int), _p(d, C.c_uint8), _p(dist, C.c_int32))
    return d, dist


class MT(object):
    def __init__(self, seed):
        self.L = lib()
        self.h = self.L.orc_mt_new(int(seed))

    def __del__(self):
        try:
            self.L.orc_mt_free(self.h)
        except Exception:
            pass

    def u32(self):
        return int(self.L.orc_mt_u32(self.h))

    def double(self):
        return float(self.L.orc_mt_double(self.h))

    def interval(self, mx):
        return int(self.L.orc_mt_interval(self.h, int(mx)))

    def permutation(self, n):
        out = np.zeros(n, np.int32)
        self.L.orc_mt_permutation(self.h, n, _p(out, C.c_int32))
        return out
