"""Pins the CPU oracle (oracle/track2d_oracle.c, numpy-legacy RNG mode) to the golden vectors captured
from the reference env (tests/golden/make_golden.py): maps, spawns, goals, scripted-target actions,
per-step obs / rewards / done — all bit-exact, over several consecutive episodes of one RNG stream."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, unpack_maze
from oracle import oracle as orc


def _names(npz):
    return [str(n) for n in npz["names"]]


def test_episode_cases_bit_exact(golden_episodes):
    assert _replay_from_seed(golden_episodes) > 2500


def test_moore_action_cases_bit_exact():
    """action_type='Moore' (track_1v1.py:243-249,277-279): the reference's episodes with the 8-action table — diagonal
    moves that cut corners, bumps, the Ram target drawing its plans from 8 actions — replayed from the seed alone."""
    g = np.load(os.path.join(GOLDEN, "episodes_moore.npz"))
    assert _replay_from_seed(g, action_type="Moore") > 1400
    assert max(int(g[n + "/ep0_act_applied"].max()) for n in _names(g)) == 7


def _replay_from_seed(g, action_type="VonNeumann"):
    checked_steps = 0
    for name in _names(g):
        mp, mode, lvl, seed, _pol = [str(x) for x in g[name + "/meta"]]
        env = orc.OracleEnv(mp, mode, int(lvl), 500, orc.RNG_NP, int(seed), action_type=action_type)
        env.seed_np(int(seed))
        for ep in range(int(g[name + "/n_eps"])):
            p = "%s/ep%d_" % (name, ep)
            obs0 = env.reset()
            maze = unpack_maze(g[p + "maze"], g[p + "side"])
            assert env.side == int(g[p + "side"]), name
            np.testing.assert_array_equal(env.maze, maze, err_msg=name + " maze")
            st = env.state()
            np.testing.assert_array_equal(st["pos"], g[p + "init"], err_msg=name + " init")
            np.testing.assert_array_equal(st["goals"], g[p + "goals"], err_msg=name + " goals")
            np.testing.assert_array_equal(obs0, g[p + "obs0"], err_msg=name + " obs0")
            if p + "plan0" in g.files:
                plan, cur = env.plan()
                np.testing.assert_array_equal(plan, g[p + "plan0"], err_msg=name + " plan0")
                assert cur == 0
            acts = g[p + "act_in"]
            for t in range(len(acts)):
                obs, rew, done, applied = env.step(acts[t])
                assert np.array_equal(obs, g[p + "obs"][t]), (name, ep, t)
                assert rew[0] == g[p + "rew"][t][0] and rew[1] == g[p + "rew"][t][1], (name, ep, t, rew)
                assert done == bool(g[p + "done"][t]), (name, ep, t)
                assert np.array_equal(applied, g[p + "act_applied"][t]), (name, ep, t)
                s = env.state()
                assert s["c_far"] == g[p + "cfar"][t]
                assert np.array_equal(s["pos"], g[p + "pos"][t])
                checked_steps += 1
    return checked_steps


def test_edge_cases_bit_exact(golden_edges):
    g = golden_edges
    for name in _names(g):
        mode = str(g[name + "/mode"])
        maze = unpack_maze(g[name + "/maze"], g[name + "/side"])
        env = orc.OracleEnv("Block", mode, 1, 0, orc.RNG_NP, 0)
        env.inject(maze, g[name + "/pos0"])
        np.testing.assert_array_equal(env.obs(), g[name + "/obs0"], err_msg=name)
        for t, a in enumerate(g[name + "/actions"]):
            obs, rew, done, _ = env.step(a)
            assert np.array_equal(obs, g[name + "/obs"][t]), (name, t)
            assert tuple(rew) == tuple(g[name + "/rew"][t]), (name, t)
            assert done == bool(g[name + "/done"][t]), (name, t)
            s = env.state()
            assert s["c_far"] == g[name + "/cfar"][t], (name, t)
            assert np.array_equal(s["pos"], g[name + "/pos"][t]), (name, t)


def test_far_run_terminates_on_eleventh_far_step(golden_edges):
    g = golden_edges
    d = g["far_run_pzr/done"]
    assert d.argmax() == 10 and d[10] == 1  # done exactly on the 11th consecutive far step


def test_astar_matches_reference(golden_astar):
    g = golden_astar
    n = int(g["count"])
    solv = 0
    for i in range(n):
        p = "a%d/" % i
        maze = unpack_maze(g[p + "maze"], g[p + "side"])
        got = orc.astar(maze, g[p + "start"], g[p + "goal"])
        if bool(g[p + "solvable"]):
            assert got is not None
            np.testing.assert_array_equal(got, g[p + "actions"], err_msg=p)
            solv += 1
        else:
            assert got is None
    assert solv >= 40


def test_time_limit_episode(golden_episodes):
    g = golden_episodes
    d = g["Block_PZR_l0_s16/ep0_done"]
    assert len(d) == 500 and d[-1] == 1 and d[:-1].sum() == 0


def test_full_observation_cases_bit_exact():
    """obs_type='Full' ids: both agents observe the whole map (tracker 2, target 4), reference fixtures."""
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "episodes_full.npz"))
    for name in _names(g):
        mp, mode, lvl, seed, _pol = [str(x) for x in g[name + "/meta"]]
        env = orc.OracleEnv(mp, mode, int(lvl), 500, orc.RNG_NP, int(seed), obs_type="Full")
        env.seed_np(int(seed))
        for ep in range(int(g[name + "/n_eps"])):
            p = "%s/ep%d_" % (name, ep)
            obs0 = env.reset()
            S = int(g[p + "side"])
            assert obs0.shape == (2, S, S)
            np.testing.assert_array_equal(obs0, g[p + "obs0"], err_msg=name)
            for t, a in enumerate(g[p + "act_in"]):
                obs, rew, done, applied = env.step(a)
                assert np.array_equal(obs, g[p + "obs"][t]), (name, ep, t)
                assert tuple(rew) == tuple(g[p + "rew"][t]) and done == bool(g[p + "done"][t])
                assert np.array_equal(obs[0], obs[1])            # Full: identical for both agents


def test_rpf_static_goal_cases_bit_exact():
    """target_mode='RPF' (static patrol goals, generators.py:12-19,48-50,68): maps, the fixed tracker spawn, goals,
    the Navigator's first plan and every step (the target's emitted actions included) against the reference,
    including an episode whose first patrol cell is a wall in the env's own copy of the map (track_1v1.py:233-236)."""
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "episodes_rpf.npz"))
    steps, walled = 0, 0
    for name in _names(g):
        mp, mode, lvl, seed, _pol = [str(x) for x in g[name + "/meta"]]
        env = orc.OracleEnv(mp, mode, int(lvl), 500, orc.RNG_NP, int(seed))
        env.seed_np(int(seed))
        for ep in range(int(g[name + "/n_eps"])):
            p = "%s/ep%d_" % (name, ep)
            obs0 = env.reset()
            maze = unpack_maze(g[p + "maze"], g[p + "side"])
            np.testing.assert_array_equal(env.maze, maze, err_msg=name + " maze")
            S = maze.shape[0]
            walled += int(maze[S * 5 // 6, S // 6] == 1)
            st = env.state()
            np.testing.assert_array_equal(st["pos"], g[p + "init"], err_msg=name + " init")
            assert tuple(st["pos"][0]) == (S // 6, S // 6)
            np.testing.assert_array_equal(st["goals"], g[p + "goals"], err_msg=name + " goals")
            np.testing.assert_array_equal(obs0, g[p + "obs0"], err_msg=name + " obs0")
            plan, cur = env.plan()
            np.testing.assert_array_equal(plan, g[p + "plan0"], err_msg=name + " plan0")
            for t, a in enumerate(g[p + "act_in"]):
                obs, rew, done, applied = env.step(a)
                assert np.array_equal(applied, g[p + "act_applied"][t]), (name, ep, t)
                assert np.array_equal(obs, g[p + "obs"][t]), (name, ep, t)
                assert tuple(rew) == tuple(g[p + "rew"][t]), (name, ep, t)
                assert done == bool(g[p + "done"][t]), (name, ep, t)
                assert np.array_equal(env.state()["pos"], g[p + "pos"][t]), (name, ep, t)
                assert env.state()["c_far"] == g[p + "cfar"][t]
                steps += 1
    assert steps > 1500 and walled >= 1
