"""-m gpu parity tests proper: the HIP path (through the C ABI, via active_tracking_rl_amd.vec_env) against
 (a) the golden vectors captured from the reference env — injected maps/spawns/actions, bit-exact obs, done,
     far counter, positions; rewards exactly float32(reference float64);
 (b) the CPU oracle in its PHILOX (device-spec) mode — generated maps, spawns, goals, Ram plans and whole
     auto-reset trajectories, bit-exact;
 (c) size-independent invariants at the BASELINE sizes (4096 / 16384 envs).
Tolerance: none — integer/byte work is bit-exact; rewards are float64 arithmetic rounded once to f32 and are
compared for equality (north_star allows 1e-6)."""
import numpy as np
import pytest
import torch

from conftest import unpack_maze
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

MODE_CODE = {"Adv": 0, "PZR": 1, "Far": 2, "Nav": 3, "Ram": 4}


@pytest.fixture(scope="module")
def vec():
    from active_tracking_rl_amd import vec_env
    assert torch.cuda.is_available(), "gpu tests need the MI355X"
    return vec_env


def _pad82(m):
    out = np.zeros((82, 82), np.uint8)
    out[: m.shape[0], : m.shape[1]] = m
    return out


def _run_injected(vec, items, action_type="VonNeumann"):
    """items: list of dict(maze, pos0, actions[T,2], obs[T,2,13,13], rew[T,2] f64, done[T], cfar[T], pos[T,2,2],
    mode, obs0). Each item becomes one env of a batch (81- and 82-sided maps in separate batches)."""
    for side in (81, 82):
        sel = [it for it in items if it["maze"].shape[0] == side]
        if not sel:
            continue
        n = len(sel)
        modes = np.array([MODE_CODE[it["mode"]] for it in sel], np.uint8)
        env = vec.VecTrack2D(num_envs=n, map_type="Block", target_mode="Adv", level=1, auto_reset=False,
                             max_episode_steps=500, target_mode_per_env=modes, action_type=action_type)
        env.inject(np.stack([it["maze"] for it in sel]), np.stack([it["pos0"].reshape(4) for it in sel]))
        obs0 = env.observe().cpu().numpy()
        for i, it in enumerate(sel):
            assert np.array_equal(obs0[i].astype(np.uint8), it["obs0"]), ("obs0", it["name"])
            assert np.array_equal(obs0[i], it["obs0"].astype(np.float32))
        T = max(len(it["actions"]) for it in sel)
        acts = np.zeros((T, n, 2), np.int64)
        for i, it in enumerate(sel):
            acts[: len(it["actions"]), i] = it["actions"]
        acts_d = torch.from_numpy(acts).cuda()
        for t in range(T):
            obs, rew, done = env.step(acts_d[t, :, 0].contiguous(), acts_d[t, :, 1].contiguous())
            obs, rew, done = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
            st = env.get_state()
            for i, it in enumerate(sel):
                if t >= len(it["actions"]):
                    continue
                tag = (it["name"], t)
                assert np.array_equal(obs[i], it["obs"][t].astype(np.float32)), tag
                assert np.array_equal(rew[i], it["rew"][t].astype(np.float32)), (tag, rew[i], it["rew"][t])
                assert bool(done[i]) == bool(it["done"][t]), tag
                assert st["c_far"][i] == min(int(it["cfar"][t]), 255), tag      # the device counter saturates at 255
                assert np.array_equal(st["pos"][i], it["pos"][t]), tag
        assert env.faults() == 0
        env.close()


def test_golden_episodes_injected(vec, golden_episodes):
    g = golden_episodes
    items = []
    for name in [str(n) for n in g["names"]]:
        mp, mode, lvl, seed, _ = [str(x) for x in g[name + "/meta"]]
        for ep in range(int(g[name + "/n_eps"])):
            p = "%s/ep%d_" % (name, ep)
            items.append(dict(name=p, maze=unpack_maze(g[p + "maze"], g[p + "side"]), pos0=g[p + "init"],
                              actions=g[p + "act_applied"], obs=g[p + "obs"], rew=g[p + "rew"], done=g[p + "done"],
                              cfar=g[p + "cfar"], pos=g[p + "pos"], obs0=g[p + "obs0"],
                              mode=mode if mode in ("PZR", "Far") else "Adv"))
    assert len(items) >= 30
    _run_injected(vec, items)


def test_golden_rpf_episodes_injected(vec):
    """The reference's RPF episodes (incl. a patrol cell that is a wall in the env's own map) replayed on the device
    with the actions the reference applied: observations, rewards, done, far counter, positions."""
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "episodes_rpf.npz"))
    items = []
    for name in [str(n) for n in g["names"]]:
        for ep in range(int(g[name + "/n_eps"])):
            p = "%s/ep%d_" % (name, ep)
            items.append(dict(name=p, maze=unpack_maze(g[p + "maze"], g[p + "side"]), pos0=g[p + "init"],
                              actions=g[p + "act_applied"], obs=g[p + "obs"], rew=g[p + "rew"], done=g[p + "done"],
                              cfar=g[p + "cfar"], pos=g[p + "pos"], obs0=g[p + "obs0"], mode="Adv"))
    assert len(items) >= 6
    _run_injected(vec, items)


def test_golden_moore_episodes_injected(vec):
    """action_type='Moore' (track_1v1.py:243-249,275-279): the reference's 8-action episodes (diagonals that cut corners,
    bumps into walls and map borders) replayed on the device; and the refusals that go with the table."""
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "episodes_moore.npz"))
    items = []
    for name in [str(n) for n in g["names"]]:
        mode = str(g[name + "/meta"][1])
        for ep in range(int(g[name + "/n_eps"])):
            p = "%s/ep%d_" % (name, ep)
            items.append(dict(name=p, maze=unpack_maze(g[p + "maze"], g[p + "side"]), pos0=g[p + "init"],
                              actions=g[p + "act_applied"], obs=g[p + "obs"], rew=g[p + "rew"], done=g[p + "done"],
                              cfar=g[p + "cfar"], pos=g[p + "pos"], obs0=g[p + "obs0"],
                              mode=mode if mode in ("PZR", "Far") else "Adv"))
    assert len(items) >= 10 and max(int(it["actions"].max()) for it in items) == 7
    _run_injected(vec, items, action_type="Moore")
    with pytest.raises(vec.T2DError):                      # scripted targets plan in the four-move table
        vec.VecTrack2D("Track2D-BlockPartialRam-v0", num_envs=4, action_type="Moore")
    with pytest.raises(TypeError):
        vec.VecTrack2D("Track2D-BlockPartialPZR-v0", num_envs=4, action_type="Hex")
    # action 4 is a fault in the four-move table, a diagonal in the eight-move one
    for at, want in (("VonNeumann", 1), ("Moore", 0)):
        env = vec.VecTrack2D("Track2D-BlockPartialPZR-v0", num_envs=8, seed=2, action_type=at)
        env.reset()
        a = torch.full((8,), 4, dtype=torch.int64, device="cuda")
        env.step(a, a)
        assert env.faults() == want, at
        env.close()
    # random-action stepping draws from all eight moves and stays on free cells
    env = vec.VecTrack2D("Track2D-MazePartialAdv-v0", num_envs=512, seed=3, action_type="Moore")
    env.reset()
    p0 = env.get_state()["pos"].copy()
    env.rollout_random(40, 7, keep_obs=False)
    st, maps = env.get_state(), env.get_maps()
    moved = st["pos"].astype(np.int64) - p0
    assert (np.abs(moved[:, 0]).min(axis=1) > 0).any()     # some tracker has moved on both axes
    for i in range(512):
        for ag in range(2):
            assert maps[i][st["pos"][i, ag, 0], st["pos"][i, ag, 1]] == 0
    assert env.faults() == 0
    env.close()


def test_golden_edges_injected(vec, golden_edges):
    g = golden_edges
    items = []
    for name in [str(n) for n in g["names"]]:
        items.append(dict(name=name, maze=unpack_maze(g[name + "/maze"], g[name + "/side"]), pos0=g[name + "/pos0"],
                          actions=g[name + "/actions"], obs=g[name + "/obs"], rew=g[name + "/rew"],
                          done=g[name + "/done"], cfar=g[name + "/cfar"], pos=g[name + "/pos"],
                          obs0=g[name + "/obs0"], mode=str(g[name + "/mode"])))
    _run_injected(vec, items)


def test_reward_table_exhaustive_bit_exact(vec):
    env = vec.VecTrack2D(num_envs=1, map_type="Block", target_mode="PZR")
    d2 = torch.arange(0, 2 * 81 * 81 + 1, dtype=torch.int32)
    for w_p in (1.0, -0.5, 0.0):
        rt, rg = env.reward_table(d2, w_p)
        rt, rg = rt.cpu().numpy(), rg.cpu().numpy()
        want = np.array([orc.reward(int(v), w_p) for v in d2.numpy()], np.float64).astype(np.float32)
        assert np.array_equal(rt, want[:, 0]) and np.array_equal(rg, want[:, 1]), w_p
    env.close()


def _oracle_batch(n, map_types, modes, levels, seed, base=0):
    return [orc.OracleEnv(map_types[i], modes[i], int(levels[i]), 500, orc.RNG_PHILOX, seed, base + i)
            for i in range(n)]


def _check_generated(vec, n, map_type, mode, level, seed, steps, per_env=None, base=0):
    if per_env is None:
        mts, mds, lvs = [map_type] * n, [mode] * n, [level] * n
        env = vec.VecTrack2D(num_envs=n, map_type=map_type, target_mode=mode, level=level, seed=seed,
                             env_id_base=base, auto_reset=True)
    else:
        mts, mds, lvs = per_env
        from active_tracking_rl_amd import registry
        env = vec.VecTrack2D(num_envs=n, map_type="Block", target_mode="Adv", level=0, seed=seed, env_id_base=base,
                             auto_reset=True,
                             map_type_per_env=np.array([registry.MAP_CODE[m] for m in mts], np.uint8),
                             target_mode_per_env=np.array([registry.TARGET_CODE[m] for m in mds], np.uint8),
                             level_per_env=np.array(lvs, np.uint8))
    oracles = _oracle_batch(n, mts, mds, lvs, seed, base)
    obs = env.reset().cpu().numpy()
    want0 = np.stack([o.reset() for o in oracles])
    maps = env.get_maps()
    st = env.get_state()
    for i, o in enumerate(oracles):
        assert st["side"][i] == o.side
        assert np.array_equal(maps[i], _pad82(o.maze)), ("map", i)
        s = o.state()
        assert np.array_equal(st["pos"][i], s["pos"]), ("spawn", i, st["pos"][i], s["pos"])
        assert np.array_equal(st["goals"][i], s["goals"]), ("goals", i)
    assert np.array_equal(obs, want0.astype(np.float32))
    rs = np.random.RandomState(seed)
    n_done = 0
    for t in range(steps):
        acts = rs.randint(0, 4, size=(n, 2))
        a = torch.from_numpy(acts).cuda()
        obs, rew, done = env.step(a[:, 0].contiguous(), a[:, 1].contiguous())
        obs, rew, done = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
        for i, o in enumerate(oracles):
            wo, wr, wd, _ = o.step(acts[i])
            if wd:
                wo = o.reset()  # auto-reset: first observation of the next episode
                n_done += 1
            assert bool(done[i]) == wd, (t, i)
            assert np.array_equal(rew[i], wr.astype(np.float32)), (t, i, rew[i], wr)
            assert np.array_equal(obs[i], wo.astype(np.float32)), (t, i)
    st = env.get_state()
    for i, o in enumerate(oracles):
        s = o.state()
        assert np.array_equal(st["pos"][i], s["pos"]) and st["c_far"][i] == s["c_far"] and st["t"][i] == s["t"]
        assert st["episode"][i] == o.L.orc_episode(o.h)
    assert env.faults() == 0
    env.close()
    return n_done


def test_generated_block_pzr_matches_oracle(vec):
    assert _check_generated(vec, 96, "Block", "PZR", 0, seed=7, steps=90) > 50


def test_generated_block_ram_matches_oracle(vec):
    assert _check_generated(vec, 64, "Block", "Ram", 0, seed=11, steps=120) > 30


def test_generated_maze_far_and_levels_match_oracle(vec):
    _check_generated(vec, 48, "Maze", "Far", 0, seed=3, steps=60)
    _check_generated(vec, 16, "Maze", "Adv", 1, seed=4, steps=40)
    _check_generated(vec, 16, "Block", "Adv", 1, seed=5, steps=40)
    _check_generated(vec, 16, "Empty", "PZR", 0, seed=6, steps=40)


def test_generated_nav_matches_oracle(vec):
    """Nav target: goal sampling, BFS direction field, closed-loop descent, re-planning at the goal."""
    _check_generated(vec, 40, "Block", "Nav", 0, seed=13, steps=150)
    _check_generated(vec, 24, "Maze", "Nav", 0, seed=14, steps=120)
    _check_generated(vec, 8, "Block", "Nav", 1, seed=15, steps=60)


def test_nav_plan_b_when_target_is_walled_in(vec):
    """Unreachable goals: six failed plans -> plan B = 10 random actions (navigator.py:22-36), repeatedly."""
    m = np.zeros((82, 82), np.uint8)
    m[0, :] = m[-1, :] = 1; m[:, 0] = m[:, -1] = 1
    m[9, 9:12] = 1; m[11, 9:12] = 1; m[10, 9] = 1; m[10, 11] = 1     # the target sits in a 1-cell pocket
    pos = np.array([[[20, 20], [10, 10]], [[12, 10], [10, 10]]], np.int32)
    n = 2
    env = vec.VecTrack2D(num_envs=n, map_type="Block", target_mode="Nav", level=1, seed=5, auto_reset=False)
    oracles = _oracle_batch(n, ["Block"] * n, ["Nav"] * n, [1] * n, 5)
    env.reset()
    for o in oracles:
        o.reset()
    env.inject(np.stack([m, m]), pos.reshape(n, 4))
    for i, o in enumerate(oracles):
        o.inject(m, pos[i])
    rs = np.random.RandomState(3)
    for t in range(35):
        acts = rs.randint(0, 4, size=(n, 2))
        a = torch.from_numpy(acts).cuda()
        obs, rew, done = env.step(a[:, 0].contiguous())
        tg = env.get_target()
        for i, o in enumerate(oracles):
            wo, wr, wd, applied = o.step(acts[i])
            assert np.array_equal(obs[i].cpu().numpy(), wo.astype(np.float32)), (t, i)
            assert np.array_equal(rew[i].cpu().numpy(), wr.astype(np.float32))
            assert o.state()["pos"][1].tolist() == [10, 10]          # never leaves the pocket
            plan, cur = o.plan()
            assert tg["len"][i] == 10 and tg["cursor"][i] == cur and np.array_equal(tg["plan"][i], plan)
    env.close()


def test_full_observation_ids(vec):
    """obs_type='Full' (Track2D-*Full*-v*): (a) reference golden episodes injected, (b) generated vs the oracle."""
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "episodes_full.npz"))
    for name in [str(n) for n in g["names"]]:
        mp, mode, lvl, seed, _ = [str(x) for x in g[name + "/meta"]]
        for ep in range(int(g[name + "/n_eps"])):
            p = "%s/ep%d_" % (name, ep)
            S = int(g[p + "side"])
            env = vec.VecTrack2D(num_envs=1, map_type=mp, target_mode="PZR" if mode == "PZR" else "Adv", level=1,
                                 auto_reset=False, obs_type="Full")
            env.inject(unpack_maze(g[p + "maze"], S)[None], g[p + "init"].reshape(1, 4))
            obs0 = env.observe().cpu().numpy()
            assert obs0.shape == (1, 2, S, S) and np.array_equal(obs0[0], g[p + "obs0"].astype(np.float32))
            for t, a in enumerate(g[p + "act_applied"]):
                at = torch.tensor([[int(a[0])], [int(a[1])]], device="cuda")
                obs, rew, done = env.step(at[0], at[1])
                assert np.array_equal(obs.cpu().numpy()[0], g[p + "obs"][t].astype(np.float32)), (name, ep, t)
                assert np.array_equal(rew.cpu().numpy()[0], g[p + "rew"][t].astype(np.float32))
                assert bool(done.item()) == bool(g[p + "done"][t])
            env.close()
    for env_id, mp, mode in (("Track2D-BlockFullPZR-v0", "Block", "PZR"), ("Track2D-MazeFullNav-v1", "Maze", "Nav")):
        n = 6
        lvl = int(env_id[-1])
        env = vec.VecTrack2D(env_id, num_envs=n, seed=4)
        oracles = [orc.OracleEnv(mp, mode, lvl, 500, orc.RNG_PHILOX, 4, i, obs_type="Full") for i in range(n)]
        obs = env.reset().cpu().numpy()
        assert np.array_equal(obs, np.stack([o.reset() for o in oracles]).astype(np.float32))
        rs = np.random.RandomState(1)
        for t in range(40):
            acts = rs.randint(0, 4, size=(n, 2))
            a = torch.from_numpy(acts).cuda()
            obs, rew, done = env.step(a[:, 0].contiguous(), a[:, 1].contiguous())
            for i, o in enumerate(oracles):
                wo, wr, wd, _ = o.step(acts[i])
                if wd:
                    wo = o.reset()
                assert np.array_equal(obs[i].cpu().numpy(), wo.astype(np.float32)), (env_id, t, i)
                assert bool(done[i].item()) == wd
        env.close()


def test_generated_mixed_batch_and_sharding(vec):
    n = 40
    rs = np.random.RandomState(0)
    mts = [("Block", "Maze")[k] for k in rs.randint(0, 2, n)]
    mds = [("Adv", "PZR", "Far", "Ram", "Nav")[k] for k in rs.randint(0, 5, n)]
    lvs = rs.randint(0, 2, n).tolist()
    _check_generated(vec, n, None, None, None, seed=21, steps=50, per_env=(mts, mds, lvs))
    # a shard [base, base+n) of a larger job is keyed by GLOBAL env ids
    _check_generated(vec, 24, "Block", "PZR", 0, seed=7, steps=30, base=1000)


def test_long_soak_with_chasing_tracker_reaches_time_limit(vec):
    """1200-step lock-step soak against the oracle with a tracker that chases the target (episodes run into the
    500-step TimeLimit, generator launches every 10 steps, Ram/Nav plans roll over many times)."""
    for mode, n in (("Ram", 96), ("Nav", 48)):
        env = vec.VecTrack2D(num_envs=n, map_type="Block", target_mode=mode, level=0, seed=77, auto_reset=True)
        oracles = _oracle_batch(n, ["Block"] * n, [mode] * n, [0] * n, 77)
        env.reset()
        for o in oracles:
            o.reset()
        rs = np.random.RandomState(5)
        limit_hits = far_hits = 0
        pos = np.stack([o.state()["pos"] for o in oracles])
        for t in range(1200):
            d = pos[:, 1] - pos[:, 0]
            vert = np.abs(d[:, 0]) >= np.abs(d[:, 1])
            a0 = np.where(vert, np.where(d[:, 0] < 0, 0, 1), np.where(d[:, 1] < 0, 2, 3))
            rnd = rs.rand(n) < 0.1
            a0 = np.where(rnd, rs.randint(0, 4, n), a0)
            acts = np.stack([a0, rs.randint(0, 4, n)], 1)
            a = torch.from_numpy(acts).cuda()
            obs, rew, done = env.step(a[:, 0].contiguous(), a[:, 1].contiguous())
            obs, rew, done = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
            for i, o in enumerate(oracles):
                wo, wr, wd, _ = o.step(acts[i])
                if wd:
                    s = o.state()
                    limit_hits += s["t"] >= 500
                    far_hits += s["c_far"] > 10
                    wo = o.reset()
                assert bool(done[i]) == wd, (mode, t, i)
                assert np.array_equal(rew[i], wr.astype(np.float32)), (mode, t, i)
                assert np.array_equal(obs[i], wo.astype(np.float32)), (mode, t, i)
                pos[i] = o.state()["pos"]
        assert limit_hits + far_hits > 50 and (mode != "Ram" or limit_hits > 20), (mode, limit_hits, far_hits)
        st = env.get_state()
        assert np.array_equal(st["pos"], pos) and env.faults() == 0
        env.close()


def test_odd_batch_sizes_and_tail_blocks(vec):
    for n in (1, 2, 3, 5, 7):
        _check_generated(vec, n, "Block", "PZR", 0, seed=100 + n, steps=25)


def test_gym_protocol_masked_reset(vec):
    n = 12
    env = vec.VecTrack2D(num_envs=n, map_type="Block", target_mode="PZR", seed=9, auto_reset=False)
    oracles = _oracle_batch(n, ["Block"] * n, ["PZR"] * n, [0] * n, 9)
    env.reset()
    for o in oracles:
        o.reset()
    rs = np.random.RandomState(1)
    for t in range(60):
        acts = rs.randint(0, 4, size=(n, 2))
        a = torch.from_numpy(acts).cuda()
        obs, rew, done = env.step(a[:, 0].contiguous(), a[:, 1].contiguous())
        want = []
        dn = []
        for i, o in enumerate(oracles):
            wo, _, wd, _ = o.step(acts[i])
            want.append(wo); dn.append(wd)
        assert np.array_equal(done.cpu().numpy().astype(bool), np.array(dn))
        assert np.array_equal(obs.cpu().numpy(), np.stack(want).astype(np.float32))  # terminal obs kept
        if any(dn):
            obs = env.reset(mask=done)
            for i, o in enumerate(oracles):
                if dn[i]:
                    want[i] = o.reset()
            assert np.array_equal(obs.cpu().numpy(), np.stack(want).astype(np.float32))
    env.close()


def test_invalid_action_sets_fault(vec):
    env = vec.VecTrack2D(num_envs=4, map_type="Block", target_mode="PZR", seed=2)
    env.reset()
    a = torch.tensor([0, 1, 7, 3], dtype=torch.int64, device="cuda")
    env.step(a, a.clone())
    assert env.faults() & 1
    env.close()


@pytest.mark.parametrize("n,map_type,mode", [(4096, "Block", "PZR"), (16384, "Block", "Adv"), (8192, "Maze", "Ram"),
                                             (8192, "Maze", "Nav")])
def test_invariants_at_baseline_sizes(vec, n, map_type, mode):
    """SURVEY.md §8c property list, checked on the full batch without the oracle."""
    env = vec.VecTrack2D(num_envs=n, map_type=map_type, target_mode=mode, seed=1, auto_reset=True)
    obs = env.reset()
    S = 81 if map_type == "Maze" else 82
    maps = env.get_maps()
    st = env.get_state()
    assert (maps[:, 0, :S] == 1).all() and (maps[:, S - 1, :S] == 1).all()
    assert (maps[:, :S, 0] == 1).all() and (maps[:, :S, S - 1] == 1).all()
    if map_type == "Block":
        k = maps[:, 1:81, 1:81].reshape(n, -1).sum(1)
        assert k.min() >= 0 and k.max() <= 959
        assert len(np.unique(k)) > 300          # K = int(0.15 * u * 6400) varies per episode
    idx = np.arange(n)
    p = st["pos"]
    assert (maps[idx, p[:, 0, 0], p[:, 0, 1]] == 0).all() and (maps[idx, p[:, 1, 0], p[:, 1, 1]] == 0).all()
    d = p[:, 0] - p[:, 1]                        # target spawns in the 2x2 block up-left of the tracker
    assert ((d >= 0) & (d <= 1)).all()
    o = obs.cpu().numpy()
    assert set(np.unique(o)) <= {0.0, 1.0, 2.0, 4.0}
    assert (o[:, 0, 6, 6] == 2).all() and (o[:, 1, 6, 6] == 4).all()
    ep_before = st["episode"].copy()
    steps_done = np.zeros(n, np.int64)
    total_done = 0
    for t in range(40):
        obs, rew, done = env.step_random(1, action_seed=5)
        dn = done.cpu().numpy().astype(bool)
        total_done += int(dn.sum())
        r = rew.cpu().numpy()
        assert (r >= -1).all() and (r[:, 0] <= 1).all()
        steps_done += 1
        assert not dn[steps_done < 11].any()     # done needs 11 consecutive far steps
        steps_done[dn] = 0
    st2 = env.get_state()
    assert (st2["episode"] - ep_before).sum() == total_done
    o = obs.cpu().numpy()
    assert (o[:, 0, 6, 6] == 2).all() and (o[:, 1, 6, 6] == 4).all()
    assert set(np.unique(o)) <= {0.0, 1.0, 2.0, 4.0}
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("env_id", ["Track2D-BlockPartialPZR-v0", "Track2D-BlockPartialRam-v0", "Track2D-MazePartialFar-v0",
                                    "Track2D-MazePartialNav-v0"])
def test_fused_random_rollout_is_bit_identical_to_single_step_launches(env_id):
    """t2d_rollout_random (up to 10 env steps per launch, state in registers / tile in LDS) against the same number of
    one-step launches from the same seed: every step's observations, rewards and done flags, and the final state."""
    import torch
    from active_tracking_rl_amd.vec_env import VecTrack2D
    n, steps = 777, 47          # odd sizes: chunks of 10 + a tail, several episode switches (max_episode_steps=25)
    a = VecTrack2D(env_id, num_envs=n, seed=5, max_episode_steps=25)
    b = VecTrack2D(env_id, num_envs=n, seed=5, max_episode_steps=25)
    oa, ob = a.reset(), b.reset()
    assert torch.equal(oa, ob)
    a.step_random(3, 9)                                  # un-aligned start: 3 steps into the generator window
    b.step_random(3, 9)
    obs, rew, done = a.rollout_random(steps, 9)
    for t in range(steps):
        o1, r1, d1 = b.step_random(1, 9)
        assert torch.equal(obs[t], o1), (env_id, t)
        assert torch.equal(rew[t], r1) and torch.equal(done[t], d1), (env_id, t)
    assert int(done.sum()) > n                           # the time limit alone ends every episode once
    sa, sb = a.get_state(), b.get_state()
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k
    assert np.array_equal(a.get_maps(), b.get_maps())
    a.close(); b.close()


@pytest.mark.gpu
def test_generated_rpf_matches_oracle(vec):
    """RPF ids: fixed tracker spawn, patrol goals without random draws, plans on the generator's map (patrol cells
    free) executed open-loop on the env's own map — generated episodes against the oracle's device-spec mode."""
    assert _check_generated(vec, 32, "Block", "RPF", 0, seed=21, steps=100) > 20
    _check_generated(vec, 16, "Maze", "RPF", 0, seed=22, steps=80)
    _check_generated(vec, 24, "Block", "RPF", 1, seed=23, steps=80)
    _check_generated(vec, 8, "Empty", "RPF", 0, seed=24, steps=60)


@pytest.mark.gpu
def test_rpf_patrol_replans_and_bumps_into_walled_patrol_cells(vec):
    """Long patrols without resets: the target walks the four patrol cells in order; two of them are walls in the
    env's own map (the generator cleared them only in its copy, track_1v1.py:233-236), so the open-loop plan bumps."""
    rs = np.random.RandomState(3)
    n = 6
    mazes = []
    for i in range(n):
        m = (rs.rand(82, 82) < 0.04).astype(np.uint8)
        m[0, :] = m[-1, :] = 1; m[:, 0] = m[:, -1] = 1
        m[13, 13] = 0; m[12:14, 12:14] = 0
        m[68, 13] = 1 if i % 2 == 0 else 0                  # patrol cell 1 walled in every other env
        m[68, 68] = 1 if i % 3 == 0 else 0
        mazes.append(m)
    mazes = np.stack(mazes)
    pos = np.tile(np.array([[13, 13], [12, 13]], np.int32), (n, 1, 1))
    env = vec.VecTrack2D(num_envs=n, map_type="Block", target_mode="RPF", level=1, seed=9, auto_reset=False,
                         max_episode_steps=0)
    oracles = [orc.OracleEnv("Block", "RPF", 1, 0, orc.RNG_PHILOX, 9, i) for i in range(n)]
    env.reset()
    for o in oracles:
        o.reset()
    env.inject(mazes, pos.reshape(n, 4))
    for i, o in enumerate(oracles):
        o.inject(mazes[i], pos[i])
    visited = set()
    for t in range(420):
        acts = rs.randint(0, 4, size=(n, 2))
        a = torch.from_numpy(acts).cuda()
        obs, rew, done = env.step(a[:, 0].contiguous(), a[:, 1].contiguous())
        obs, rew = obs.cpu().numpy(), rew.cpu().numpy()
        for i, o in enumerate(oracles):
            wo, wr, wd, applied = o.step(acts[i])
            assert np.array_equal(obs[i], wo.astype(np.float32)), (t, i)
            assert np.array_equal(rew[i], wr.astype(np.float32)), (t, i)
            visited.add((i, tuple(o.state()["pos"][1])))
    st = env.get_state()
    for i, o in enumerate(oracles):
        assert np.array_equal(st["pos"][i], o.state()["pos"])
    # the patrol really went round: env 1 (nothing walled) stood on at least three patrol cells
    assert sum(((1, c) in visited) for c in ((13, 13), (68, 13), (68, 68), (13, 68))) >= 3
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("env_id,n", [("Track2D-BlockPartialPZR-v0", 4096), ("Track2D-BlockPartialRam-v0", 333),
                                      ("Track2D-MazePartialAdv-v0", 2)])
def test_u8_observations_equal_the_f32_ones(env_id, n):
    """t2d_step_u8 (obs left as bytes, SURVEY 8(d) B_step = 709 variant) against t2d_step from the same seed with the
    same actions: identical rewards / done flags and obs_u8 == obs_f32 cell for cell, across episode switches."""
    import torch
    from active_tracking_rl_amd.vec_env import VecTrack2D
    a = VecTrack2D(env_id, num_envs=n, seed=11, max_episode_steps=30)
    b = VecTrack2D(env_id, num_envs=n, seed=11, max_episode_steps=30)
    assert torch.equal(a.reset(), b.reset())
    g = torch.Generator(device="cuda").manual_seed(3)
    for t in range(70):
        acts = torch.randint(0, 4, (2, n), device="cuda", generator=g, dtype=torch.uint8 if t % 2 else torch.int64)
        of, rf, df = a.step(acts[0], acts[1])
        ou, ru, du = b.step_u8(acts[0], acts[1])
        assert ou.dtype == torch.uint8 and torch.equal(of, ou.float()), (env_id, t)
        assert torch.equal(rf, ru) and torch.equal(df, du), (env_id, t)
    sa, sb = a.get_state(), b.get_state()
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k
    a.close(); b.close()


@pytest.mark.gpu
def test_unaligned_observation_buffer_takes_the_scalar_store_path():
    """An obs pointer that is only 4-byte aligned (a view at an odd float offset) must give the same observations."""
    import torch
    from active_tracking_rl_amd.vec_env import VecTrack2D
    n = 65
    a = VecTrack2D("Track2D-BlockPartialPZR-v0", num_envs=n, seed=2)
    b = VecTrack2D("Track2D-BlockPartialPZR-v0", num_envs=n, seed=2)
    a.reset(); b.reset()
    raw = torch.zeros(n * 338 + 8, device="cuda")
    view = raw[1:1 + n * 338].view(n, 2, 13, 13)
    assert view.data_ptr() % 16 != 0
    rew, done = torch.empty((n, 2), device="cuda"), torch.empty((n,), dtype=torch.uint8, device="cuda")
    for t in range(15):
        acts = torch.randint(0, 4, (2, n), device="cuda")
        o1, r1, d1 = a.step(acts[0], acts[1])
        o2, r2, d2 = b.step(acts[0], acts[1], out=(view, rew, done))
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2), t
    assert float(raw[0]) == 0.0 and float(raw[1 + n * 338:].abs().sum()) == 0.0      # nothing written outside the view
    a.close(); b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("env_id,max_steps", [("Track2D-BlockPartialPZR-v0", 25), ("Track2D-MazePartialNav-v0", 25),
                                              ("Track2D-BlockPartialRam-v0", 11), ("Track2D-MazePartialFar-v0", 7),
                                              ("Track2D-BlockPartialRPF-v0", 500)])
def test_async_generator_is_bit_identical_to_the_in_order_one(env_id, max_steps):
    """t2d_generator_async: consumed next-episode slots refilled on the library's side stream, one stamp window (10 steps
    with the two next-episode slots per env) behind the steps, against the in-order generator from the same seed — every
    observation, reward and done flag of single-step launches and of the fused multi-step launches, the final state and
    maps. Episode lengths down to 7 (windows of 6 steps) and the far rule's 11-step minimum are both exercised."""
    import torch
    from active_tracking_rl_amd.vec_env import VecTrack2D
    n = 515
    a = VecTrack2D(env_id, num_envs=n, seed=11, max_episode_steps=max_steps, async_gen=True)
    b = VecTrack2D(env_id, num_envs=n, seed=11, max_episode_steps=max_steps)
    assert a.async_gen and not b.async_gen
    g_every = 2 * min(11, max_steps) - 2          # two pre-generated episodes per env: one pass per 2 L - 2 steps
    assert a.generator_cycle == 2 * (g_every // 2) and b.generator_cycle == g_every
    assert torch.equal(a.reset(), b.reset())
    ndone = 0
    for t in range(83):
        oa, ra, da = a.step_random(1, 3)
        ob, rb, db = b.step_random(1, 3)
        assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(da, db), (env_id, t)
        ndone += int(da.sum())
    obs, rew, done = a.rollout_random(38, 4)
    obs2, rew2, done2 = b.rollout_random(38, 4)
    assert torch.equal(obs, obs2) and torch.equal(rew, rew2) and torch.equal(done, done2)
    assert ndone + int(done.sum()) > (2 * n if max_steps <= 25 else 0)
    # masked reset and a mode switch in mid-window flush the forked launches first
    mask = torch.zeros(n, dtype=torch.uint8, device="cuda")
    mask[::3] = 1
    assert torch.equal(a.reset(mask), b.reset(mask))
    a.generator_async(False)
    b.generator_async(True)
    for t in range(27):
        oa, ra, da = a.step_random(1, 5)
        ob, rb, db = b.step_random(1, 5)
        assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(da, db), (env_id, t)
    sa, sb = a.get_state(), b.get_state()
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k
    assert np.array_equal(a.get_maps(), b.get_maps())
    a.close(); b.close()


@pytest.mark.gpu
def test_device_ram_and_nav_targets_follow_reference_plans(vec, golden_episodes):
    """The device's OWN scripted-target code on reference data (not just the injected `act_applied` stream):
    Ram — the reference's first plan (`plan0`, RamAgent.reset, navigator.py:90-93) injected with t2d_inject_plan: until its
    last entry (which RamAgent.step may overwrite, :81-83) the device's target must move exactly as the reference's did;
    Nav — the reference's first goal (`navgoal0`, Navigator.reset, navigator.py:43-63) injected with t2d_inject_nav_goal:
    the device's target, following its BFS direction field on the reference's map, reaches it in exactly len(plan0)
    steps — the length of the reference's A* plan — moving one cell every step."""
    g = golden_episodes
    checked_ram = checked_nav = 0
    for name in [str(n) for n in g["names"]]:
        mp, mode, lvl, seed, _ = [str(x) for x in g[name + "/meta"]]
        if mode not in ("Ram", "Nav"):
            continue
        for ep in range(int(g[name + "/n_eps"])):
            p = "%s/ep%d_" % (name, ep)
            maze = unpack_maze(g[p + "maze"], g[p + "side"])
            side = int(g[p + "side"])
            plan0 = g[p + "plan0"]
            env = vec.VecTrack2D(num_envs=1, map_type=mp, target_mode=mode, level=int(lvl), auto_reset=False,
                                 max_episode_steps=500)
            env.inject(maze[None], g[p + "init"].reshape(1, 4))
            if mode == "Ram":
                env.inject_plan(0, plan0)
                n = min(len(plan0) - 1, len(g[p + "act_applied"]))
                for t in range(n):
                    a0 = torch.tensor([int(g[p + "act_applied"][t, 0])], device="cuda")
                    obs, rew, done = env.step(a0, a0)        # the target's action argument is ignored: the plan decides
                    assert int(g[p + "act_applied"][t, 1]) == int(plan0[t])
                    assert np.array_equal(env.get_state()["pos"][0], g[p + "pos"][t]), (p, t)
                    assert np.array_equal(obs.cpu().numpy()[0], g[p + "obs"][t].astype(np.float32)), (p, t)
                    assert np.array_equal(rew.cpu().numpy()[0], g[p + "rew"][t].astype(np.float32)), (p, t)
                    checked_ram += 1
            else:
                goal = g[p + "navgoal0"].reshape(-1)[:2]
                env.inject_nav_goal(0, goal)
                a0 = torch.zeros(1, dtype=torch.int64, device="cuda")
                prev = env.get_state()["pos"][0, 1].copy()
                for t in range(len(plan0)):
                    env.step(a0, a0)
                    cur = env.get_state()["pos"][0, 1]
                    assert abs(int(cur[0]) - int(prev[0])) + abs(int(cur[1]) - int(prev[1])) == 1, (p, t)   # never bumps
                    assert maze[cur[0], cur[1]] == 0
                    assert (t == len(plan0) - 1) == bool(np.array_equal(cur, goal)), (p, t, cur, goal)
                    prev = cur.copy()
                    checked_nav += 1
            assert env.faults() == 0
            env.close()
    assert checked_ram >= 40 and checked_nav >= 200


@pytest.mark.parametrize("env_id,mixed", [("Track2D-MazePartialNav-v0", False), ("Track2D-MazePartialPZR-v1", False),
                                          ("Track2D-BlockPartialAdv-v0", True)])
def test_pregrown_mazes_give_the_same_episodes(env_id, mixed):
    """t2d_pregrow (k_pregrow): Maze maps grown AHEAD of the generator pass, on another stream, into a per-env ring the pass
    copies from. Two handles with the same seed — one that grows every maze inside the pass, one with pregrow after every pass
    (auto mode: forked behind the pass, under the following steps) — stepped with the same actions through many short episodes:
    every observation, reward and done flag equal, final states equal, and the ring is really used (most mazes taken from it).
    RandomMazeGenerator._generate_maze, generators.py:115-145; reset(), track_1v1.py:134-168."""
    from active_tracking_rl_amd.vec_env import VecTrack2D
    n, steps = 192, 260
    kw = dict(num_envs=n, seed=21, env_id_base=640, max_episode_steps=13)
    if mixed:
        kw["map_type_per_env"] = np.array([i % 2 for i in range(n)], np.uint8)      # Block / Maze alternating
    a, b = VecTrack2D(env_id, **kw), VecTrack2D(env_id, **kw)
    b.pregrow_auto(True)
    oa, ob = a.reset(), b.reset()
    assert torch.equal(oa, ob)
    rs = np.random.RandomState(5)
    dones = 0
    for t in range(steps):
        acts = torch.from_numpy(rs.randint(0, 4, size=(2, n))).cuda()
        ra = a.step(acts[0], acts[1])
        rb = b.step(acts[0], acts[1])
        if t % 37 == 5:
            b.pregrow(fork=False)                                   # (and the in-order form now and then: same ring)
        for x, y, what in zip(ra, rb, ("obs", "rew", "done")):
            assert torch.equal(x, y), (what, t)
        dones += int(ra[2].sum())
        if t == 59:
            st0 = b.pregrow_stats()                                 # (the first passes found an empty ring)
    sa, sb = a.get_state(), b.get_state()
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k
    assert np.array_equal(a.get_maps(), b.get_maps())
    st = b.pregrow_stats()
    assert dones > 10 * n
    taken, inline = st["taken"] - st0["taken"], st["grown_in_pass"] - st0["grown_in_pass"]
    assert st["pregrown"] > 0 and taken > 100 and taken >= 9 * inline, (st0, st)
    assert a.pregrow_stats()["taken"] == 0
    assert a.faults() == 0 and b.faults() == 0
    a.close(); b.close()
