"""ctypes binding of include/track2d_np.h: the reference-exact episode source (numpy-legacy MT19937 stream +
heapq-faithful A*, host side of libtrack2d_hip.so). Used by environment.Track2DEnv(rng="numpy"): what it generates
goes to the device through t2d_inject, the scripted target's action through t2d_step."""
import ctypes as C

import numpy as np

from . import registry, vec_env

NP_SYMBOLS = ("t2d_np_create", "t2d_np_destroy", "t2d_np_last_error", "t2d_np_seed", "t2d_np_reset",
              "t2d_np_target_action", "t2d_np_get_plan", "t2d_np_astar", "t2d_np_draw", "t2d_np_reset_many",
              "t2d_np_target_actions", "t2d_np_mt_state", "t2d_np_attach", "t2d_np_terminal_d2", "t2d_np_astar_device")
_ready = False


def _lib():
    global _ready
    L = vec_env.load_library()
    if not _ready:
        vp, i32, u32 = C.c_void_p, C.c_int32, C.c_uint32
        L.t2d_np_last_error.restype = C.c_char_p
        L.t2d_np_create.restype = i32
        L.t2d_np_create.argtypes = [i32, i32, i32, u32, C.POINTER(vp)]
        L.t2d_np_destroy.restype = i32
        L.t2d_np_destroy.argtypes = [vp]
        L.t2d_np_seed.restype = i32
        L.t2d_np_seed.argtypes = [vp, u32]
        L.t2d_np_reset.restype = i32
        L.t2d_np_reset.argtypes = [vp, vp, vp, vp, vp]
        L.t2d_np_target_action.restype = i32
        L.t2d_np_target_action.argtypes = [vp, vp]
        L.t2d_np_get_plan.restype = i32
        L.t2d_np_get_plan.argtypes = [vp, vp, i32, vp, vp, vp]
        L.t2d_np_astar.restype = i32
        L.t2d_np_astar.argtypes = [vp, i32, vp, vp, vp, i32, vp, vp]
        L.t2d_np_draw.restype = i32
        L.t2d_np_draw.argtypes = [vp, i32, u32, u32, vp]
        L.t2d_np_mt_state.restype = i32
        L.t2d_np_mt_state.argtypes = [u32, vp]
        L.t2d_np_astar_device.restype = i32
        L.t2d_np_astar_device.argtypes = [i32, vp, i32, vp, vp, vp, i32, vp, vp]
        L.t2d_np_attach.restype = i32
        L.t2d_np_attach.argtypes = [vp, vp]
        L.t2d_np_reset_many.restype = i32
        L.t2d_np_reset_many.argtypes = [vp, i32, vp, vp, vp, vp, i32]
        L.t2d_np_target_actions.restype = i32
        L.t2d_np_target_actions.argtypes = [vp, i32, vp, i32]
        _ready = True
    return L


class NpError(RuntimeError):
    pass


def _check(rc):
    if rc != 0:
        raise NpError("track2d_np error %d: %s" % (rc, _lib().t2d_np_last_error().decode()))


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class NpEpisodeSource(object):
    """One env's random side: np.random.seed(seed), then reset() / target_action() in the reference's draw order."""

    def __init__(self, map_type, target_mode, level=0, seed=0):
        self.L = _lib()
        self.map_type, self.target_mode, self.level = map_type, target_mode, int(level)
        h = C.c_void_p()
        _check(self.L.t2d_np_create(registry.MAP_CODE[map_type], registry.TARGET_CODE[target_mode], int(level),
                                    int(seed) & 0xFFFFFFFF, C.byref(h)))
        self.h = h
        self.scripted = target_mode in ("Ram", "Nav", "RPF")

    def close(self):
        if getattr(self, "h", None):
            self.L.t2d_np_destroy(self.h)
            self.h = None

    __del__ = close

    def seed(self, seed):
        _check(self.L.t2d_np_seed(self.h, int(seed) & 0xFFFFFFFF))

    def reset(self):
        """-> (maze u8 [side, side], pos int32 [2, 2] (tracker, target), goals int32 [2, 2])."""
        maze = np.zeros((82, 82), np.uint8)
        side = np.zeros(1, np.int32)
        pos, goals = np.zeros(4, np.int32), np.zeros(4, np.int32)
        _check(self.L.t2d_np_reset(self.h, _ptr(maze), _ptr(side), _ptr(pos), _ptr(goals)))
        s = int(side[0])
        return np.ascontiguousarray(maze[:s, :s]), pos.reshape(2, 2), goals.reshape(2, 2)

    def target_action(self):
        a = np.zeros(1, np.int32)
        _check(self.L.t2d_np_target_action(self.h, _ptr(a)))
        return int(a[0])

    def plan(self):
        """-> (plan_actions int32 [len], a_i, navgoal int32 [2])."""
        ln, cur, ng = np.zeros(1, np.int32), np.zeros(1, np.int32), np.zeros(2, np.int32)
        _check(self.L.t2d_np_get_plan(self.h, None, 0, _ptr(ln), _ptr(cur), _ptr(ng)))
        plan = np.zeros(max(int(ln[0]), 1), np.int32)
        _check(self.L.t2d_np_get_plan(self.h, _ptr(plan), int(ln[0]), _ptr(ln), _ptr(cur), _ptr(ng)))
        return plan[:int(ln[0])], int(cur[0]), ng

    def draw(self, kind, arg=0, count=1):
        out = np.zeros(max(int(arg) if kind == 2 else int(count), 1), np.float64)
        _check(self.L.t2d_np_draw(self.h, int(kind), int(arg), int(count), _ptr(out)))
        return out[:int(arg) if kind == 2 else int(count)]


class NpBatchSource(object):
    """The random side of N envs, each with its own numpy-legacy stream (np.random.seed(seeds[i]) semantics), advanced
    together on host threads (t2d_np_reset_many / t2d_np_target_actions). map_types / target_modes / levels: one value or
    one per env."""

    def __init__(self, map_types, target_modes, levels, seeds, threads=0):
        n = len(seeds)
        expand = lambda v: list(v) if isinstance(v, (list, tuple, np.ndarray)) else [v] * n
        self.map_types, self.target_modes, self.levels = expand(map_types), expand(target_modes), [int(x) for x in expand(levels)]
        assert len(self.map_types) == n and len(self.target_modes) == n and len(self.levels) == n
        self.n, self.threads, self.L = n, int(threads), _lib()
        self.src = [NpEpisodeSource(self.map_types[i], self.target_modes[i], self.levels[i], int(seeds[i])) for i in range(n)]
        self.scripted = np.array([s.scripted for s in self.src], bool)

    def close(self):
        for s in self.src:
            s.close()

    def _handles(self, idx):
        return (C.c_void_p * len(idx))(*[self.src[i].h for i in idx])

    def reset(self, idx=None):
        """Episodes for the envs in idx (all when None) -> (mazes u8 [k, 82, 82], sides int32 [k], pos int32 [k, 2, 2],
        goals int32 [k, 2, 2])."""
        idx = list(range(self.n)) if idx is None else [int(i) for i in idx]
        k = len(idx)
        mazes = np.zeros((k, 82, 82), np.uint8)
        sides, pos, goals = np.zeros(k, np.int32), np.zeros((k, 4), np.int32), np.zeros((k, 4), np.int32)
        if k:
            _check(self.L.t2d_np_reset_many(self._handles(idx), k, _ptr(mazes), _ptr(sides), _ptr(pos), _ptr(goals), self.threads))
        return mazes, sides, pos.reshape(k, 2, 2), goals.reshape(k, 2, 2)

    def target_actions(self, idx):
        """The scripted targets' actions for the next step of the envs in idx (each must have a Ram / Nav / RPF target)."""
        idx = [int(i) for i in idx]
        a = np.zeros(len(idx), np.int32)
        if idx:
            _check(self.L.t2d_np_target_actions(self._handles(idx), len(idx), _ptr(a), self.threads))
        return a


def mt_states(seeds):
    """uint32 [len(seeds), 625]: the MT19937 state np.random.seed(seed) leaves behind, per seed (624 words + the read position)."""
    L = _lib()
    out = np.zeros((len(seeds), 625), np.uint32)
    for i, sd in enumerate(seeds):
        _check(L.t2d_np_mt_state(int(sd) & 0xFFFFFFFF, _ptr(out[i])))
    return out


def attach_device_streams(core, seeds):
    """Hand a vec_env.VecTrack2D one numpy-legacy stream per env (np.random.seed(seeds[i])): from now on its generator — reset()
    and the pre-generated episodes of the in-launch auto-reset — restates the reference's draws on the device (t2d_np_attach,
    csrc/track2d_hip.hip k_gen_np). Before the first reset. Adv / PZR / Far (and host-driven Ext) targets on any handle; Ram
    targets (whose draws interleave with the resets: RamAgent, navigator.py:73-93) on handles created with auto_reset=False — the
    episode of a restarted env is then drawn inside reset(mask) and RamAgent.step() runs on the device ahead of every step
    (k_ram_np), as does Navigator.step() for Nav and RPF targets, with the reference's heap A* restated on the device."""
    assert len(seeds) == core.num_envs
    st = np.ascontiguousarray(mt_states(seeds))
    rc = _lib().t2d_np_attach(core.h, _ptr(st))
    if rc != 0:
        raise NpError("t2d_np_attach failed (%d): %s" % (rc, core.L.t2d_last_error().decode()))


def astar_device(maze, start, goal, max_len=8192, device=0):
    """astar() run by the device restatement (csrc/track2d_hip.hip astar_np): -> (solvable, actions int32 [n])."""
    maze = np.ascontiguousarray(maze, np.uint8)
    side = maze.shape[0]
    assert maze.shape == (side, side)
    st, gl = np.asarray(start, np.int32).reshape(2).copy(), np.asarray(goal, np.int32).reshape(2).copy()
    acts = np.zeros(max_len, np.int32)
    n, ok = np.zeros(1, np.int32), np.zeros(1, np.int32)
    rc = _lib().t2d_np_astar_device(int(device), _ptr(maze), side, _ptr(st), _ptr(gl), _ptr(acts), max_len, _ptr(n), _ptr(ok))
    if rc != 0:
        from . import vec_env
        raise NpError("t2d_np_astar_device failed (%d): %s" % (rc, vec_env.load_library().t2d_last_error().decode()))
    return bool(ok[0]), acts[:int(n[0])].copy()


def astar(maze, start, goal, max_len=8192):
    """AstarSolver(start, [0, 1, 2, 3], maze, goal): -> (solvable, actions int32 [n])."""
    maze = np.ascontiguousarray(maze, np.uint8)
    side = maze.shape[0]
    assert maze.shape == (side, side)
    st, gl = np.asarray(start, np.int32).reshape(2).copy(), np.asarray(goal, np.int32).reshape(2).copy()
    acts = np.zeros(max_len, np.int32)
    n, ok = np.zeros(1, np.int32), np.zeros(1, np.int32)
    _check(_lib().t2d_np_astar(_ptr(maze), side, _ptr(st), _ptr(gl), _ptr(acts), max_len, _ptr(n), _ptr(ok)))
    assert int(n[0]) <= max_len
    return bool(ok[0]), acts[:int(n[0])].copy()
