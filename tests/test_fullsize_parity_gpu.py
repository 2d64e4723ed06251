"""-m gpu: full-size lock-step parity of the STAND-ALONE step kernel (`k_step2`, the C ABI's `t2d_step` / `t2d_step_u8`)
against the C oracle (PHILOX mode), driven with externally chosen actions.

What this covers: `environment.VecEnv.step(actions, out=slot of rollout_buffers)` — byte observations written into a
rollout store, int64 action tensors, in-launch auto-reset, the generator pass — at the full per-GPU shard size of every
BASELINE.json GPU configuration with the LAST rank's `env_id_base`; >= 30 steps, every env compared every step
(observations, rewards == float32(oracle float64), done flags) and the final positions / far counters / step counters /
episode numbers. This is the path of `Agent.action_test`, the evaluator, `Track2DEnv` and any caller that brings its own
actions; it is NOT the kernel bench.py's timed region runs: since round 3 the rollout ends each env step inside
`k_act_step` (fused.act_env_step) in replayed hipGraphs — that path has its own lock-step test,
tests/test_timed_region_parity_gpu.py.
The oracle side is `oracle.OracleBatch` (orc_step_batch: a plain loop over the scalar oracle, test infrastructure).
Reference semantics: envs/gym-track2d/gym_track2d/envs/track_1v1.py:71-168."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def _lockstep(env_id, n, base, steps, seed=1, map_types=None, policy="random", min_done=0):
    from active_tracking_rl_amd import registry
    from active_tracking_rl_amd.environment import VecEnv
    sp = registry.spec(env_id)
    over = {}
    mts = [sp["map_type"]] * n
    if map_types is not None:
        mts = list(map_types)
        over["map_type_per_env"] = np.array([registry.MAP_CODE[m] for m in mts], np.uint8)
    env = VecEnv(env_id, n, device="cuda:0", seed=seed, env_id_base=base, obs_u8=True, **over)
    # the byte path is the production path wherever the kernels offer it (every 'Partial' id since round 3)
    assert env.obs_u8 == env.core.supports_u8
    T = 20
    buf = env.rollout_buffers(T)
    oracle = orc.OracleBatch([orc.OracleEnv(mts[i], sp["target_mode"], sp["level"], 500, orc.RNG_PHILOX, seed, base + i)
                              for i in range(n)])
    obs = env.reset()
    want = oracle.reset()
    assert obs.dtype == (torch.uint8 if env.obs_u8 else torch.float32)
    assert np.array_equal(obs.reshape(n, 2, 13, 13).cpu().numpy().astype(np.uint8), want)
    rs = np.random.RandomState(seed + 17)
    pos = np.stack([o.state()["pos"] for o in oracle.envs])
    n_done = 0
    for t in range(steps):
        if policy == "chase":    # a tracker that runs after the target: long episodes, Ram / Nav plans roll over
            d = pos[:, 1] - pos[:, 0]
            vert = np.abs(d[:, 0]) >= np.abs(d[:, 1])
            a0 = np.where(vert, np.where(d[:, 0] < 0, 0, 1), np.where(d[:, 1] < 0, 2, 3))
            a0 = np.where(rs.rand(n) < 0.15, rs.randint(0, 4, n), a0)
        else:
            a0 = rs.randint(0, 4, n)
        acts = np.stack([a0, rs.randint(0, 4, n)], 1).astype(np.int64)
        a = torch.from_numpy(acts).cuda()
        k = t % T
        out = (buf[0][k + 1], buf[1][k], buf[2][k])
        obs, rew, done, _ = env.step([a[:, 0].contiguous(), a[:, 1].contiguous()], out=out)
        wo, wr, wd = oracle.step(acts)
        got_o = buf[0][k + 1].cpu().numpy()
        assert np.array_equal(got_o.astype(np.uint8), wo), (env_id, t, np.nonzero((got_o.astype(np.uint8) != wo).any((1, 2, 3)))[0][:8])
        assert np.array_equal(buf[1][k].cpu().numpy(), wr.astype(np.float32)), (env_id, t)
        assert np.array_equal(buf[2][k].cpu().numpy(), wd), (env_id, t)
        n_done += int(wd.sum())
        if policy == "chase":
            pos = np.stack([o.state()["pos"] for o in oracle.envs])
    st = env.core.get_state()
    for i in (list(range(0, n, max(1, n // 97))) + [n - 1]):
        s = oracle.envs[i].state()
        assert np.array_equal(st["pos"][i], s["pos"]) and st["c_far"][i] == s["c_far"] and st["t"][i] == s["t"], i
        assert st["episode"][i] == oracle.L.orc_episode(oracle.envs[i].h), i
    assert env.core.faults() == 0
    assert n_done >= min_done, (env_id, n_done)
    env.close()
    return n_done


def test_config2_headline_4096_pzr_first_and_last_rank():
    """BASELINE configs[2]: Track2D-BlockPartialPZR-v0, 4096 envs per GPU — rank 0 and rank 7 of an 8-GPU weak run."""
    _lockstep("Track2D-BlockPartialPZR-v0", 4096, 0, 40, min_done=1000)
    _lockstep("Track2D-BlockPartialPZR-v0", 4096, 28672, 40, min_done=1000)


def test_config2_strong_shards_512():
    """The strong form of the headline: 4096 envs over 8 GPUs = 512 per GPU; the last rank's shard is [3584, 4096)."""
    _lockstep("Track2D-BlockPartialPZR-v0", 512, 3584, 60, min_done=100)


def test_config1_1024_ram():
    """BASELINE configs[1]: Track2D-BlockPartialRam-v0, 1024 envs (the k_step2<..., RAM> variant)."""
    _lockstep("Track2D-BlockPartialRam-v0", 1024, 0, 45, policy="chase")
    _lockstep("Track2D-BlockPartialRam-v0", 1024, 7168, 30, min_done=100)


def test_config3_1024_maze_nav_last_rank():
    """BASELINE configs[3]: Track2D-MazePartialNav-v0, 8192 envs over 8 GPUs — the shard of rank 7 (env ids 7168..8191)."""
    _lockstep("Track2D-MazePartialNav-v0", 1024, 7168, 40, min_done=100)
    _lockstep("Track2D-MazePartialNav-v0", 1024, 0, 60, policy="chase")


def test_maze_nav_4096_single_gpu_generator_path():
    """MazePartialNav above 2048 envs on one GPU: the generator pass is `k_gen<NAV>` there (one wave per generated episode;
    up to 2048 envs it is `k_gen_nav`, one workgroup per episode, covered above) — same draws, same episodes."""
    _lockstep("Track2D-MazePartialNav-v0", 4096, 4096, 45, min_done=400)


def test_config4_2048_adv_mixed_maps_last_rank():
    """BASELINE configs[4]: Track2D-BlockPartialAdv-v0, 16384 envs over 8 GPUs, Block/Maze 50/50 per batch — rank 7's shard."""
    mts = ["Block" if i % 2 == 0 else "Maze" for i in range(2048)]
    _lockstep("Track2D-BlockPartialAdv-v0", 2048, 14336, 35, map_types=mts, min_done=300)


def test_invariants_16384_mixed_block_maze():
    """SURVEY.md §8c property list on configs[4]'s full 16384-env batch with the 50/50 Block/Maze mix (no oracle)."""
    from active_tracking_rl_amd import registry, vec_env
    n = 16384
    mtc = np.array([registry.MAP_CODE["Block"] if i % 2 == 0 else registry.MAP_CODE["Maze"] for i in range(n)], np.uint8)
    env = vec_env.VecTrack2D("Track2D-BlockPartialAdv-v0", num_envs=n, seed=3, map_type_per_env=mtc)
    obs = env.reset()
    maps, st = env.get_maps(), env.get_state()
    side = st["side"]
    assert np.array_equal(side, np.where(mtc == registry.MAP_CODE["Maze"], 81, 82))
    for s in (81, 82):
        m = maps[side == s][:, :s, :s]
        assert m[:, 0].all() and m[:, -1].all() and m[:, :, 0].all() and m[:, :, -1].all()          # border walls
    blk = maps[side == 82]
    k = blk[:, 1:81, 1:81].reshape(len(blk), -1).sum(1)
    assert k.min() >= 0 and k.max() <= 959                                                          # int(0.15 U * 6400)
    idx = np.arange(n)
    p = st["pos"]
    assert (maps[idx, p[:, 0, 0], p[:, 0, 1]] == 0).all() and (maps[idx, p[:, 1, 0], p[:, 1, 1]] == 0).all()
    d = p[:, 0] - p[:, 1]
    assert ((d >= 0) & (d <= 1)).all()                                    # target spawn in the 2x2 window up-left
    o = obs.cpu().numpy()
    assert set(np.unique(o).tolist()) <= {0.0, 1.0, 2.0, 4.0}
    assert (o[:, 0, 6, 6] == 2).all() and (o[:, 1, 6, 6] == 4).all()
    acts = torch.randint(0, 4, (60, 2, n), device="cuda")
    ndone = 0
    for t in range(60):
        obs, rew, done = env.step(acts[t, 0], acts[t, 1])
        s2 = env.get_state() if t % 20 == 19 else None
        ndone += int(done.sum().item())
        if s2 is not None:
            dd = s2["pos"][:, 1] - s2["pos"][:, 0]
            assert np.array_equal(s2["d2"], (dd * dd).sum(1).astype(np.uint32))
    o = obs.cpu().numpy()
    assert (o[:, 0, 6, 6] == 2).all() and (o[:, 1, 6, 6] == 4).all() and ndone > 2000 and env.faults() == 0
    env.close()
