"""Property tests (hypothesis) of the invariants listed in SURVEY.md §8c, on the CPU oracle in BOTH random modes —
no reference needed at run time. They guard the restated generators/step against regressions and document the
reference's quirks (2x2 spawn window, exact obstacle count, done only on the 11th far step or the time limit)."""
import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import oracle as orc

MODES = ["Adv", "PZR", "Far", "Ram", "Nav"]


@settings(max_examples=40, deadline=None)
@given(seed=st.integers(0, 2**31 - 1), map_type=st.sampled_from(["Block", "Maze", "Empty"]),
       mode=st.sampled_from(MODES), level=st.integers(0, 1), rng=st.sampled_from([orc.RNG_NP, orc.RNG_PHILOX]))
def test_reset_invariants(seed, map_type, mode, level, rng):
    env = orc.OracleEnv(map_type, mode, level, 500, rng, seed, env_id=seed % 977)
    if rng == orc.RNG_NP:
        env.seed_np(seed)
    obs = env.reset()
    m, S = env.maze, env.side
    assert S == (81 if map_type == "Maze" else 82)
    assert m[0].all() and m[-1].all() and m[:, 0].all() and m[:, -1].all()           # border walls
    if map_type == "Block":
        k = int(m[1:-1, 1:-1].sum())
        assert k == 320 if level == 1 else 0 <= k <= 959                              # exactly int(r * 6400)
    if map_type == "Empty":
        assert int(m[1:-1, 1:-1].sum()) == 0
    s = env.state()
    (r0, c0), (r1, c1) = s["pos"]
    assert m[r0, c0] == 0 and m[r1, c1] == 0                                          # spawns on free cells
    assert 0 <= r0 - r1 <= 1 and 0 <= c0 - c1 <= 1                                    # 2x2 window up-left (get_around)
    for g in s["goals"]:
        assert m[g[0], g[1]] == 0
    assert [r0, c0] not in [list(g) for g in s["goals"]]                              # goal_test loop
    assert set(np.unique(obs)) <= {0, 1, 2, 4}
    assert obs[0, 6, 6] == 2 and obs[1, 6, 6] == 4                                    # centre = own colour


@settings(max_examples=25, deadline=None)
@given(seed=st.integers(0, 2**31 - 1), mode=st.sampled_from(MODES), rng=st.sampled_from([orc.RNG_NP, orc.RNG_PHILOX]),
       map_type=st.sampled_from(["Block", "Maze"]))
def test_step_invariants(seed, mode, rng, map_type):
    env = orc.OracleEnv(map_type, mode, 0, 60, rng, seed, env_id=3)
    if rng == orc.RNG_NP:
        env.seed_np(seed)
    env.reset()
    rs = np.random.RandomState(seed % 1000)
    far_run = 0
    for t in range(1, 61):
        before = env.state()["pos"].copy()
        obs, rew, done, applied = env.step(rs.randint(0, 4, 2))
        s = env.state()
        m = env.maze
        for i in range(2):                                   # one 4-connected step or a wall bump
            d = np.abs(s["pos"][i] - before[i]).sum()
            assert d in (0, 1) and m[s["pos"][i][0], s["pos"][i][1]] == 0
        d2 = s["d2"]
        assert d2 == int(((s["pos"][0] - s["pos"][1]) ** 2).sum())
        far_run = 0 if d2 <= 36 else far_run + 1
        assert s["c_far"] == far_run
        assert done == (far_run > 10 or t >= 60)             # 11th consecutive far step, or the TimeLimit
        w_p = {"PZR": 1.0, "Far": -0.5}.get(mode, 0.0)
        assert tuple(rew) == orc.reward(d2, w_p)
        assert -1 <= rew[0] <= 1 and rew[1] >= -1
        assert set(np.unique(obs)) <= {0, 1, 2, 4} and obs[0, 6, 6] == 2 and obs[1, 6, 6] == 4
        # the other agent shows up in the window iff it is within Chebyshev distance 6
        dr, dc = s["pos"][1] - s["pos"][0]
        if max(abs(dr), abs(dc)) <= 6 and (dr, dc) != (0, 0):
            assert obs[0, 6 + dr, 6 + dc] == 4 and obs[1, 6 - dr, 6 - dc] == 2
        else:
            assert (obs[0] == 4).sum() == 0 and (obs[1] == 2).sum() == 0
        if done:
            break


@settings(max_examples=15, deadline=None)
@given(seed=st.integers(0, 2**31 - 1), map_type=st.sampled_from(["Block", "Maze"]))
def test_bfs_direction_field_descends_along_shortest_paths(seed, map_type):
    """Device-spec Nav planner: following dir from any reachable cell reaches the goal in exactly dist steps, and
    that length equals the heapq-faithful A* plan length (the reference planner)."""
    env = orc.OracleEnv(map_type, "PZR", 0, 500, orc.RNG_PHILOX, seed, 1)
    env.reset()
    m = env.maze
    free = np.argwhere(m == 0)
    rs = np.random.RandomState(seed % 997)
    goal = free[rs.randint(len(free))]
    d, dist = orc.bfs_field(m, goal)
    DR, DC = (-1, 1, 0, 0), (0, 0, -1, 1)
    for _ in range(5):
        s = free[rs.randint(len(free))]
        if dist[s[0], s[1]] < 0:
            assert d[s[0], s[1]] == 255
            assert orc.astar(m, s, goal) is None
            continue
        r, c, n = int(s[0]), int(s[1]), 0
        while (r, c) != (int(goal[0]), int(goal[1])):
            a = int(d[r, c])
            assert a < 4
            r, c, n = r + DR[a], c + DC[a], n + 1
            assert m[r, c] == 0
        assert n == dist[s[0], s[1]]
        plan = orc.astar(m, s, goal)
        assert plan is not None and len(plan) == n


@settings(max_examples=30, deadline=None)
@given(seed=st.integers(0, 2**31 - 1), map_type=st.sampled_from(["Block", "Maze", "Empty"]), level=st.integers(0, 1),
       rng=st.sampled_from([orc.RNG_NP, orc.RNG_PHILOX]))
def test_rpf_reset_and_patrol_invariants(seed, map_type, level, rng):
    """RPF ids (static goals): tracker spawn fixed at patrol cell 0 = (S/6, S/6) — even when the env's own copy of the
    map has a wall there —, target in the 2x2 window up-left of it, both goals = patrol cell 1, the scripted target
    only ever emits legal actions, and the env never moves an agent INTO a wall."""
    env = orc.OracleEnv(map_type, "RPF", level, 500, rng, seed, env_id=seed % 331)
    if rng == orc.RNG_NP:
        env.seed_np(seed)
    env.reset()
    S, m = env.side, env.maze
    s = env.state()
    lo, hi = S // 6, S * 5 // 6
    assert tuple(s["pos"][0]) == (lo, lo)
    assert 0 <= s["pos"][0][0] - s["pos"][1][0] <= 1 and 0 <= s["pos"][0][1] - s["pos"][1][1] <= 1
    assert [tuple(g) for g in s["goals"]] == [(hi, lo), (hi, lo)]
    rs = np.random.RandomState(seed % 911)
    for t in range(40):
        before = env.state()["pos"].copy()
        obs, rew, done, applied = env.step(rs.randint(0, 4, 2))
        assert 0 <= applied[1] <= 3
        after = env.state()["pos"]
        for i in range(2):
            d = np.abs(after[i] - before[i]).sum()
            assert d in (0, 1)
            if d == 1:
                assert m[after[i][0], after[i][1]] == 0
        assert set(np.unique(obs)) <= {0, 1, 2, 4}
