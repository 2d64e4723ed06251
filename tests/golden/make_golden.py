#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE ENV in the build container.

Run once, here (needs /root/reference; the GPU box never sees it):

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_golden.py

What it does
  * puts tests/golden/_refstubs (a ~40-line stand-in for the absent gym / skimage packages) and
    /root/reference/envs/gym-track2d on sys.path and imports gym_track2d.envs.track_1v1.Track1v1Env;
  * neutralises the argument-less np.random.seed() calls inside generators.py:41,56 (they make the
    reference non-deterministic) so that one np.random.seed(s) fixes the whole trajectory;
  * records, per case, the map, spawns, goals, the action streams as consumed by _next_state, and
    per-step obs / rewards / done / far counter / positions.

The outputs are DATA (inputs + expected outputs) — no reference source text is stored.
gym's TimeLimit (gym==0.12.5, not vendored) is restated as `done |= elapsed >= 500` when
`time_limit` is set for a case.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")
sys.path.insert(0, os.path.join(HERE, "_refstubs"))
sys.path.insert(0, os.path.join(REF, "envs", "gym-track2d"))

import numpy as np  # noqa: E402

from gym_track2d.envs.track_1v1 import Track1v1Env  # noqa: E402
from gym_track2d.envs.Astar_solver import AstarSolver  # noqa: E402
import gym_track2d  # noqa: E402,F401  (fills the registry stub)
from gym.envs.registration import REGISTRY  # noqa: E402

_real_seed = np.random.seed


def patched_seed(*a, **k):
    if a or k:
        return _real_seed(*a, **k)
    return None  # argument-less re-seed neutralised


np.random.seed = patched_seed


def pack_maze(m):
    m = np.asarray(m)
    return np.packbits((m != 0).astype(np.uint8).reshape(-1)), m.shape[0]


def chase_action(env, rs):
    (r0, c0), (r1, c1) = [list(map(int, s)) for s in env.state]
    cands = []
    if r1 < r0: cands.append(0)
    if r1 > r0: cands.append(1)
    if c1 < c0: cands.append(2)
    if c1 > c0: cands.append(3)
    if not cands or rs.rand() < 0.15:
        return int(rs.randint(0, 4))
    return int(cands[rs.randint(0, len(cands))])


def chase_action_moore(env, rs):
    """Chase with the Moore table (track_1v1.py:277-279): diagonals 4 (-1,+1), 5 (+1,+1), 6 (-1,-1), 7 (+1,-1)."""
    (r0, c0), (r1, c1) = [list(map(int, s)) for s in env.state]
    dr, dc = (r1 > r0) - (r1 < r0), (c1 > c0) - (c1 < c0)
    if rs.rand() < 0.15 or (dr == 0 and dc == 0):
        return int(rs.randint(0, 8))
    table = {(-1, 0): 0, (1, 0): 1, (0, -1): 2, (0, 1): 3, (-1, 1): 4, (1, 1): 5, (-1, -1): 6, (1, -1): 7}
    return table[(dr, dc)]


def run_case(map_type, mode, level, seed, n_episodes, max_steps, policy="random", time_limit=500, obs_type="Partial",
             stop_on_done=True, action_type="VonNeumann"):
    env = Track1v1Env(map_type=map_type, target_mode=mode, level=level, obs_type=obs_type, action_type=action_type)
    n_act = 8 if action_type == "Moore" else 4
    assert env.action_space[0].n == n_act
    emitted = []
    if env.Target:
        tgt = env.Target[0]
        orig_step = tgt.step

        def rec_step(*a, **k):
            out = orig_step(*a, **k)
            act = out[0] if isinstance(out, tuple) else out
            emitted.append(int(np.asarray(act).reshape(-1)[0]))
            return out

        tgt.step = rec_step
    rs = np.random.RandomState(10_000 + seed)  # private action stream (not the global RNG)
    np.random.seed(seed)                       # seeds the reference's global stream
    eps = []
    for ep in range(n_episodes):
        obs0 = env.reset()
        rec = dict(maze=np.array(env.maze), init=np.array(env.init_states, np.int32).copy(),
                   goals=np.array(env.goal_states, np.int32).copy(), obs0=np.asarray(obs0).astype(np.uint8),
                   act_in=[], act_applied=[], obs=[], rew=[], done=[], cfar=[], pos=[])
        if mode == "Ram":
            rec["plan0"] = np.array(env.Target[0].plan_actions, np.int32).copy()
        if mode in ("Nav", "RPF"):
            rec["plan0"] = np.array(env.Target[0].plan_actions, np.int32).copy()
            rec["navgoal0"] = np.array(env.Target[0].goal_states, np.int32).copy()
        t = 0
        while True:
            if policy == "chase":
                a0 = chase_action_moore(env, rs) if n_act == 8 else chase_action(env, rs)
            else:
                a0 = int(rs.randint(0, n_act))
            a1 = int(rs.randint(0, n_act))
            n_em = len(emitted)
            obs, rew, done, info = env.step([a0, a1])
            t += 1
            applied1 = emitted[n_em] if len(emitted) > n_em else a1
            if time_limit and t >= time_limit:
                done = True
            rec["act_in"].append([a0, a1]); rec["act_applied"].append([a0, applied1])
            rec["obs"].append(np.asarray(obs).astype(np.uint8).reshape(2, *np.asarray(obs).shape[-2:]))
            rec["rew"].append(np.asarray(rew, np.float64).copy()); rec["done"].append(bool(done))
            rec["cfar"].append(int(env.C_far))
            rec["pos"].append(np.array(env.state, np.int32).copy())
            assert set(np.unique(rec["obs"][-1])) <= {0, 1, 2, 4}
            if (done and stop_on_done) or t >= max_steps:
                break
        eps.append(rec)
    return eps


def flatten(prefix, eps, out):
    out[prefix + "n_eps"] = np.int32(len(eps))
    for i, r in enumerate(eps):
        p = "%sep%d_" % (prefix, i)
        bits, side = pack_maze(r["maze"])
        out[p + "maze"] = bits; out[p + "side"] = np.int32(side)
        out[p + "init"] = r["init"]; out[p + "goals"] = r["goals"]; out[p + "obs0"] = r["obs0"].reshape(2, *r["obs0"].shape[-2:])
        out[p + "act_in"] = np.array(r["act_in"], np.uint8); out[p + "act_applied"] = np.array(r["act_applied"], np.uint8)
        out[p + "obs"] = np.array(r["obs"], np.uint8); out[p + "rew"] = np.array(r["rew"], np.float64)
        out[p + "done"] = np.array(r["done"], np.uint8); out[p + "cfar"] = np.array(r["cfar"], np.int32)
        out[p + "pos"] = np.array(r["pos"], np.int32)
        for k in ("plan0", "navgoal0"):
            if k in r:
                out[p + k] = r[k]


def episodes():
    out = {}
    cases = []
    # (map, mode, level, seed, n_episodes, max_steps, policy)
    for mode in ("PZR", "Adv", "Far", "Ram"):
        cases.append(("Block", mode, 0, 11, 3, 120, "random"))
        cases.append(("Maze", mode, 0, 12, 3, 120, "random"))
    cases.append(("Block", "PZR", 1, 13, 2, 80, "random"))
    cases.append(("Maze", "PZR", 1, 14, 2, 80, "random"))
    cases.append(("Empty", "PZR", 0, 15, 2, 60, "random"))
    cases.append(("Block", "PZR", 0, 16, 1, 520, "chase"))   # runs into the 500-step TimeLimit
    cases.append(("Block", "Ram", 0, 17, 2, 300, "chase"))
    cases.append(("Block", "Nav", 0, 18, 2, 150, "chase"))
    cases.append(("Maze", "Nav", 0, 19, 2, 150, "chase"))
    cases.append(("Maze", "Ram", 1, 20, 2, 100, "chase"))
    names = []
    for (mp, mode, lvl, seed, n_ep, mx, pol) in cases:
        name = "%s_%s_l%d_s%d" % (mp, mode, lvl, seed)
        eps = run_case(mp, mode, lvl, seed, n_ep, mx, pol)
        flatten(name + "/", eps, out)
        out[name + "/meta"] = np.array([mp, mode, str(lvl), str(seed), pol])
        names.append(name)
        print(name, [len(e["obs"]) for e in eps], [bool(e["done"][-1]) for e in eps])
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "episodes.npz"), **out)


def rpf_episodes():
    """target_mode='RPF' ids (static goals, generators.py:12-19,48-50,68): the target patrols four fixed cells; the
    env keeps walls the generator cleared at those cells (track_1v1.py:233-236). Seeds are searched so that at least
    one Block case has a wall on a candidate cell in the env's own map."""
    out = {}
    names = []
    cases = [("Block", 0, 41, 2, 260, "chase"), ("Maze", 0, 42, 2, 200, "chase"), ("Empty", 0, 43, 1, 150, "random"),
             ("Block", 1, None, 2, 300, "chase")]
    for (mp, lvl, seed, n_ep, mx, pol) in cases:
        if seed is None:                    # find a level-1 Block seed whose env map has a wall on a patrol cell
            for seed in range(50, 400):
                eps = run_case(mp, "RPF", lvl, seed, n_ep, 3, pol)
                cand = [(13, 13), (68, 13), (68, 68), (13, 68)]
                if any(e["maze"][r][c] == 1 for e in eps for (r, c) in cand[1:]):
                    break
        name = "%s_RPF_l%d_s%d" % (mp, lvl, seed)
        eps = run_case(mp, "RPF", lvl, seed, n_ep, mx, pol, stop_on_done=False)   # keep stepping: patrol re-plans
        flatten(name + "/", eps, out)
        out[name + "/meta"] = np.array([mp, "RPF", str(lvl), str(seed), pol])
        names.append(name)
        print("rpf", name, [len(e["obs"]) for e in eps], [[int(e["maze"][r][c]) for (r, c) in
              ((13, 13), (e["maze"].shape[0] * 5 // 6, 13))] for e in eps])
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "episodes_rpf.npz"), **out)


def moore_episodes():
    """action_type='Moore' (track_1v1.py:17,243-249,277-279): 8 actions, diagonals cut corners (only the destination cell
    is tested). No registered id selects it; the constructor argument exists, so the table is pinned too. (With a Nav /
    RPF target the reference itself fails under Moore: the Navigator's A* receives the 8-action list and indexes its
    4-entry transition table — Astar_solver.py:138,163-169 -> KeyError — so those combinations do not exist.)"""
    out = {}
    names = []
    for (mp, mode, lvl, seed, n_ep, mx, pol) in (("Block", "PZR", 0, 61, 2, 150, "random"), ("Maze", "Adv", 0, 62, 2, 150, "random"),
                                                 ("Block", "Far", 1, 63, 2, 200, "chase"), ("Maze", "PZR", 1, 64, 2, 200, "chase"),
                                                 ("Block", "Ram", 0, 65, 2, 200, "chase"), ("Empty", "PZR", 0, 66, 1, 120, "chase")):
        name = "%s_%s_l%d_s%d" % (mp, mode, lvl, seed)
        eps = run_case(mp, mode, lvl, seed, n_ep, mx, pol, action_type="Moore")
        flatten(name + "/", eps, out)
        out[name + "/meta"] = np.array([mp, mode, str(lvl), str(seed), pol])
        names.append(name)
        acts = np.concatenate([np.array(e["act_applied"]).reshape(-1) for e in eps])
        print("moore", name, [len(e["obs"]) for e in eps], "diagonal share %.2f" % float((acts >= 4).mean()))
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "episodes_moore.npz"), **out)


def full_obs_episodes():
    """obs_type='Full' ids (Track2D-*Full*-v*): short episodes, obs [2, S, S] per step."""
    out = {}
    names = []
    for (mp, mode, lvl, seed) in (("Block", "PZR", 0, 31), ("Maze", "Ram", 0, 32), ("Block", "Nav", 1, 33)):
        name = "%s_%s_l%d_s%d" % (mp, mode, lvl, seed)
        eps = run_case(mp, mode, lvl, seed, 2, 25, "random", obs_type="Full")
        flatten(name + "/", eps, out)
        out[name + "/meta"] = np.array([mp, mode, str(lvl), str(seed), "random"])
        names.append(name)
        print("full", name, [len(e["obs"]) for e in eps], eps[0]["obs"][0].shape)
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "episodes_full.npz"), **out)


def edge_cases():
    """Hand-placed situations driven through the reference env's own step()."""
    out = {}
    names = []

    def drive(name, maze, pos, actions, mode="PZR"):
        env = Track1v1Env(map_type="Block", target_mode="Adv" if mode == "Adv" else mode, level=1)
        np.random.seed(5)
        env.reset()
        env.maze = np.array(maze, dtype=float)
        env.state = [list(pos[0]), list(pos[1])]
        env.init_states = env.state
        env.C_far = 0
        obs0 = np.asarray(env._get_obs()).astype(np.uint8).reshape(2, 13, 13)
        obs, rew, done, cfar, posl = [], [], [], [], []
        for a in actions:
            o, r, d, _ = env.step(list(a))
            obs.append(np.asarray(o).astype(np.uint8).reshape(2, 13, 13)); rew.append(np.asarray(r, np.float64).copy())
            done.append(bool(d)); cfar.append(int(env.C_far)); posl.append(np.array(env.state, np.int32).copy())
        bits, side = pack_maze(maze)
        p = name + "/"
        out[p + "maze"] = bits; out[p + "side"] = np.int32(side); out[p + "pos0"] = np.array(pos, np.int32)
        out[p + "obs0"] = obs0; out[p + "actions"] = np.array(actions, np.uint8)
        out[p + "obs"] = np.array(obs, np.uint8); out[p + "rew"] = np.array(rew, np.float64)
        out[p + "done"] = np.array(done, np.uint8); out[p + "cfar"] = np.array(cfar, np.int32)
        out[p + "pos"] = np.array(posl, np.int32); out[p + "mode"] = np.array(mode)
        names.append(name)

    def empty(S=82):
        m = np.zeros((S, S), np.uint8)
        m[0, :] = m[-1, :] = 1; m[:, 0] = m[:, -1] = 1
        return m

    rs = np.random.RandomState(77)
    m = empty()
    # co-located agents, then separate
    drive("colocated", m, [[10, 10], [10, 10]], [[0, 0], [0, 1], [2, 3], [3, 2], [1, 1]])
    # corners: maximum padding (6 out-of-bounds rows/cols) and wall bumps
    drive("corner_tl", m, [[1, 1], [1, 2]], [[0, 0], [2, 2], [0, 2], [3, 3], [1, 1]])
    drive("corner_br", m, [[80, 80], [80, 79]], [[1, 1], [3, 3], [1, 3], [2, 2], [0, 0]])
    drive("corner_tr_bl", m, [[1, 80], [80, 1]], [[0, 1], [3, 2], [2, 3], [1, 0]], mode="Far")
    # other agent exactly at Chebyshev distance 6 / 7 (window edge) and d2 = 36 vs 37
    drive("cheb6", m, [[40, 40], [34, 46]], [[0, 0], [1, 1], [1, 1], [2, 3], [3, 2]])
    drive("d2_36_37", m, [[40, 40], [40, 46]], [[2, 3], [3, 2], [0, 1], [1, 0], [2, 2], [2, 3]])
    drive("d2_37_diag", m, [[40, 40], [41, 46]], [[0, 0], [0, 1], [1, 1], [2, 3]])
    # 11-step far run -> done on the 11th consecutive far step; re-entry resets the counter
    drive("far_run", m, [[10, 10], [10, 30]], [[2, 3]] * 5 + [[3, 2]] * 30 + [[2, 3]] * 40, mode="Adv")
    drive("far_run_pzr", m, [[10, 10], [30, 30]], [[0, 1]] * 14)
    drive("far_run_farmode", m, [[10, 10], [10, 20]], [[2, 3]] * 14, mode="Far")
    # dense random obstacles: K = 959 (max level-0 density) and K = 0
    dm = empty()
    idx = rs.permutation(6400)[:959]
    dm[1 + idx // 80, 1 + idx % 80] = 1
    free = np.argwhere(dm == 0)
    p0 = free[rs.randint(len(free))]
    acts = rs.randint(0, 4, size=(60, 2)).tolist()
    drive("dense959", dm, [list(map(int, p0)), list(map(int, p0))], acts)
    drive("dense959_far", dm, [list(map(int, free[5])), list(map(int, free[-5]))], acts[:20], mode="Far")
    # maze-sized (81x81) empty map = Maze with density 0
    drive("maze81_empty", empty(81), [[79, 79], [78, 79]], rs.randint(0, 4, size=(30, 2)).tolist())
    drive("maze81_tl", empty(81), [[1, 1], [2, 1]], rs.randint(0, 4, size=(30, 2)).tolist(), mode="Adv")
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "edges.npz"), **out)
    print("edges:", names)


def _dist_worker(job):
    """One worker of distributions(): `count` resets of the REFERENCE env on one np.random.seed, statistics only."""
    map_type, mode, seed, count = job
    np.random.seed(seed)
    env = Track1v1Env(map_type=map_type, target_mode=mode, level=0)
    S = 82 if map_type == "Block" else 81
    st = dict(walls=np.zeros(64, np.int64), wall_rows=np.zeros(S, np.int64), wall_cols=np.zeros(S, np.int64),
              offs=np.zeros(4, np.int64), tr_rows=np.zeros(S, np.int64), tr_cols=np.zeros(S, np.int64),
              g0_rows=np.zeros(S, np.int64), g1_cols=np.zeros(S, np.int64), plan_len=np.zeros(32, np.int64),
              first_act=np.zeros(4, np.int64), flags=np.zeros(4, np.int64))
    for _ in range(count):
        env.reset()
        m = (np.array(env.maze) != 0)
        inner = m[1:-1, 1:-1]
        k = int(inner.sum())
        # Block: K = int(0.15 U * 6400) in [0, 959] -> 64 bins of 15; Maze: interior wall cells, 64 bins of 16 (clipped)
        st["walls"][min(63, k // (15 if map_type == "Block" else 16))] += 1
        st["wall_rows"] += m.sum(1); st["wall_cols"] += m.sum(0)
        (r0, c0), (r1, c1) = [list(map(int, x)) for x in env.init_states]
        st["offs"][(r1 - r0 + 1) * 2 + (c1 - c0 + 1)] += 1          # (-1,-1) (-1,0) (0,-1) (0,0)
        st["tr_rows"][r0] += 1; st["tr_cols"][c0] += 1
        st["g0_rows"][int(env.goal_states[0][0])] += 1; st["g1_cols"][int(env.goal_states[1][1])] += 1
        if mode == "Ram":
            plan = np.asarray(env.Target[0].plan_actions)
            st["plan_len"][len(plan)] += 1
            st["first_act"][int(plan[0])] += 1
        if mode == "Nav":
            tgt = env.Target[0]
            plan_b = isinstance(tgt.plan_actions, np.ndarray)        # np.random.choice(..., 10): plan B (navigator.py:58-59)
            st["flags"][0] += int(plan_b)
            st["flags"][1] += int(list(map(int, tgt.goal_states)) != list(map(int, env.goal_states[1])))   # goal re-drawn
            if not plan_b:
                st["plan_len"][min(31, len(tgt.plan_actions) // 8)] += 1     # A* path length, 32 bins of 8 cells
                st["first_act"][int(tgt.plan_actions[0])] += 1
    st["flags"][3] = count
    return (map_type, mode, seed), st


def distributions():
    """What the reference's generators DRAW, as histograms over >= 20 000 resets per case: the device's
    Philox generators are a different bit source by design, so 'same episodes' can only mean 'same distributions' — these
    counts are what tests/test_generator_distributions*.py hold the device (and the oracle's PHILOX mode) to. Two halves
    (different seeds) are stored separately so that a test can scale its bound by the reference's own half-vs-half noise.
    Reference: generators.py:38-94,115-176; navigator.py:43-63,73-93; track_1v1.py:218-240."""
    import multiprocessing as mp
    cases = [("Block", "Ram"), ("Maze", "Ram"), ("Block", "Nav"), ("Maze", "Nav")]
    per_job, jobs_per_half = 1250, 8                                # 2 halves x 8 jobs x 1250 = 20 000 resets per case
    jobs = []
    for ci, (mt, md) in enumerate(cases):
        for h in range(2):
            for j in range(jobs_per_half):
                jobs.append((mt, md, 900000 + ci * 1000 + h * 100 + j, per_job))
    with mp.Pool(min(8, os.cpu_count() or 1)) as pool:
        res = pool.map(_dist_worker, jobs, chunksize=1)
    out = {}
    for (mt, md, seed), st in res:
        h = (seed // 100) % 10
        for k, v in st.items():
            key = "%s_%s/h%d/%s" % (mt, md, h, k)
            out[key] = out.get(key, 0) + v
    out["cases"] = np.array(["%s_%s" % c for c in cases])
    np.savez_compressed(os.path.join(HERE, "distributions.npz"), **out)
    for mt, md in cases:
        a = out["%s_%s/h0/flags" % (mt, md)]
        print("distributions", mt, md, "resets per half", int(a[3]), "offsets", out["%s_%s/h0/offs" % (mt, md)].tolist(),
              "planB / re-drawn", a[:2].tolist())


def astar_cases():
    out = {}
    rs = np.random.RandomState(123)
    n = 0
    for map_type in ("Block", "Maze"):
        for level in (0, 1):
            for k in range(4):
                env = Track1v1Env(map_type=map_type, target_mode="PZR", level=level)
                np.random.seed(200 + n)
                env.reset()
                maze = np.array(env.maze)
                free = np.argwhere(maze == 0)
                for q in range(3):
                    s = free[rs.randint(len(free))]; g = free[rs.randint(len(free))]
                    if q == 2:
                        g = s.copy()  # start == goal -> empty plan
                    sol = AstarSolver([int(s[0]), int(s[1])], [0, 1, 2, 3], maze, [int(g[0]), int(g[1])])
                    p = "a%d/" % n
                    bits, side = pack_maze(maze)
                    out[p + "maze"] = bits; out[p + "side"] = np.int32(side)
                    out[p + "start"] = np.array(s, np.int32); out[p + "goal"] = np.array(g, np.int32)
                    out[p + "solvable"] = np.bool_(sol.solvable())
                    out[p + "actions"] = np.array(sol.get_actions() if sol.solvable() else [], np.int32)
                    n += 1
    # walled-off goal -> unsolvable
    m = np.zeros((82, 82), np.uint8); m[0, :] = m[-1, :] = 1; m[:, 0] = m[:, -1] = 1
    m[9, 9:12] = 1; m[11, 9:12] = 1; m[10, 9] = 1; m[10, 11] = 1
    sol = AstarSolver([3, 3], [0, 1, 2, 3], m.astype(float), [10, 10])
    p = "a%d/" % n
    bits, side = pack_maze(m)
    out[p + "maze"] = bits; out[p + "side"] = np.int32(side); out[p + "start"] = np.array([3, 3], np.int32)
    out[p + "goal"] = np.array([10, 10], np.int32); out[p + "solvable"] = np.bool_(sol.solvable())
    out[p + "actions"] = np.array([], np.int32)
    n += 1
    out["count"] = np.int32(n)
    np.savez_compressed(os.path.join(HERE, "astar.npz"), **out)
    print("astar cases:", n)


def registry():
    ids = sorted(REGISTRY)
    rows = [[i, REGISTRY[i]["kwargs"]["map_type"], REGISTRY[i]["kwargs"]["obs_type"], str(REGISTRY[i]["kwargs"]["level"]),
             REGISTRY[i]["kwargs"]["target_mode"], str(REGISTRY[i]["max_episode_steps"])] for i in ids]
    np.savez_compressed(os.path.join(HERE, "registry.npz"), rows=np.array(rows))
    print("registry ids:", len(ids))


def det_weights(shape, k):
    """Deterministic pseudo-weights shared with tests/test_model.py (no checkpoint needs to be stored)."""
    n = int(np.prod(shape))
    fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else int(shape[0])
    w = np.sin(np.arange(n, dtype=np.float64) * 0.7391 + 0.1 * k) / np.sqrt(max(fan_in, 1))
    return w.astype(np.float32).reshape(shape)


def model_fixture():
    """Reference A3C_Dueling.forward(test=True) (model.py:238-265) on deterministic weights and inputs."""
    import argparse
    import torch
    sys.path.insert(0, REF)
    import model as ref_model  # the reference's model.py (needs the cv2 stub for utils.py)
    from gym import spaces
    out = {}
    rs = np.random.RandomState(5)
    for net in ("tat-maze-lstm", "maze-lstm"):
        args = argparse.Namespace(stack_frames=1, rnn_out=128, network=net, single=False)
        obs_space = [spaces.Box(0, 6, (1, 13, 13), np.float32) for _ in range(2)]
        act_space = [spaces.Discrete(4) for _ in range(2)]
        torch.manual_seed(0)
        m = ref_model.build_model(obs_space, act_space, args, torch.device("cpu"))
        sd = m.state_dict()
        keys = sorted(sd.keys())
        for k, name in enumerate(keys):
            sd[name].copy_(torch.from_numpy(det_weights(tuple(sd[name].shape), k)))
        m.eval()
        B = 6
        states = rs.choice([0, 1, 2, 4], size=(B, 2, 1, 1, 13, 13)).astype(np.float32)
        hx = rs.randn(B, 2, 128).astype(np.float32) * 0.3
        cx = rs.randn(B, 2, 128).astype(np.float32) * 0.3
        vals, acts, ents, lps, hxo, cxo, rp = [], [], [], [], [], [], []
        for b in range(B):
            with torch.no_grad():
                v, a, e, lp, (h, c), r = m((torch.from_numpy(states[b]), (torch.from_numpy(hx[b]), torch.from_numpy(cx[b]))), True)
            vals.append(v.numpy()); acts.append([int(x) for x in a]); ents.append(e.numpy()); lps.append(lp.numpy())
            hxo.append(h.numpy()); cxo.append(c.numpy()); rp.append(np.asarray(r.numpy() if hasattr(r, "numpy") else r, np.float32).reshape(-1))
        p = net + "/"
        out[p + "keys"] = np.array(keys); out[p + "shapes"] = np.array([str(tuple(sd[k].shape)) for k in keys])
        out[p + "states"] = states; out[p + "hx"] = hx; out[p + "cx"] = cx
        out[p + "values"] = np.array(vals); out[p + "actions"] = np.array(acts); out[p + "entropies"] = np.array(ents)
        out[p + "log_probs"] = np.array(lps); out[p + "hx_out"] = np.array(hxo); out[p + "cx_out"] = np.array(cxo)
        out[p + "r_pred"] = np.array(rp)
        out[p + "n_params"] = np.int64(sum(v.numel() for v in sd.values()))
        print(net, "params", int(out[p + "n_params"]), "values", np.array(vals).shape)
    np.savez_compressed(os.path.join(HERE, "model.npz"), **out)


def loss_fixture():
    """Reference Agent.optimize (player_util.py:108-161) on synthetic rollout buffers: pins the n-step return /
    GAE / entropy / aux-reward loss arithmetic for done and not-done rollouts and the three training modes."""
    import argparse
    import torch
    sys.path.insert(0, REF)
    import player_util as ref_pu
    from gym import spaces
    out = {}
    rs = np.random.RandomState(9)

    class FakeEnv(object):
        observation_space = [spaces.Box(0, 6, (1, 13, 13), np.float32)] * 2
        action_space = [spaces.Discrete(4)] * 2

    case = 0
    for aux in ("reward", "none"):
        for done in (True, False):
            for mode in (-1, 0, 1):
                T = int(rs.randint(3, 9))
                args = argparse.Namespace(network="tat-maze-lstm", rnn_out=128, gamma=0.9, tau=1.0, entropy=0.01, aux=aux)
                ag = ref_pu.Agent(None, FakeEnv(), args, None, torch.device("cpu"))
                ag.w_entropy_target = 0.2
                boot = torch.tensor(rs.randn(2, 1), dtype=torch.float32)
                lin = torch.nn.Linear(1, 1)  # stands in for shared_model (only .parameters() is used)

                class StubModel(torch.nn.Module):   # returns the bootstrap value; no parameters of its own
                    def forward(self, inp, test=False, b=boot):
                        return b.clone(), None, None, None, None, None

                ag.model = StubModel()
                ag.state = torch.zeros(2, 1, 1, 13, 13)
                ag.done = done
                vals = rs.randn(T, 2, 1).astype(np.float32); lps = -np.abs(rs.randn(T, 2, 1)).astype(np.float32)
                ents = np.abs(rs.randn(T, 2, 1)).astype(np.float32); rews = rs.uniform(-1, 1, (T, 2, 1)).astype(np.float32)
                preds = rs.randn(T, 1, 1).astype(np.float32)
                leaf = torch.ones(1, requires_grad=True)
                ag.values = [torch.from_numpy(v) * leaf for v in vals]
                ag.log_probs = [torch.from_numpy(v) * leaf for v in lps]
                ag.entropies = [torch.from_numpy(v) * leaf for v in ents]
                ag.rewards = [torch.from_numpy(v) for v in rews]
                ag.preds = [torch.from_numpy(v) * leaf for v in preds]
                opt = torch.optim.SGD([leaf], lr=0.0)
                pl, vl, en, pr = ag.optimize([leaf], opt, lin, mode, torch.device("cpu"))
                p = "c%d/" % case
                out[p + "aux"] = np.array(aux); out[p + "done"] = np.bool_(done); out[p + "mode"] = np.int32(mode)
                out[p + "boot"] = boot.numpy(); out[p + "values"] = vals; out[p + "log_probs"] = lps
                out[p + "entropies"] = ents; out[p + "rewards"] = rews; out[p + "preds"] = preds
                out[p + "policy_loss"] = pl.detach().numpy(); out[p + "value_loss"] = vl.detach().numpy()
                out[p + "entropy_sum"] = en.detach().numpy(); out[p + "pred_loss"] = pr.detach().numpy()
                out[p + "dloss_dleaf"] = leaf.grad.numpy().copy()   # scalar summary of the total loss gradient
                case += 1
    out["count"] = np.int32(case)
    np.savez_compressed(os.path.join(HERE, "loss.npz"), **out)
    print("loss cases:", case)


if __name__ == "__main__":
    if "--full-only" in sys.argv:
        full_obs_episodes()
        sys.exit(0)
    if "--rpf-only" in sys.argv:
        rpf_episodes()
        sys.exit(0)
    if "--moore-only" in sys.argv:
        moore_episodes()
        sys.exit(0)
    if "--model-only" in sys.argv:
        model_fixture()
        loss_fixture()
        sys.exit(0)
    if "--distributions-only" in sys.argv:
        distributions()
        sys.exit(0)
    episodes()
    full_obs_episodes()
    rpf_episodes()
    moore_episodes()
    edge_cases()
    astar_cases()
    registry()
    model_fixture()
    loss_fixture()
    distributions()
