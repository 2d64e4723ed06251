"""-m gpu: oracle lock-step of the TIMED REGION itself — the rollouts bench.py and main.py actually run.

tests/test_fullsize_parity_gpu.py steps `VecEnv.step` (t2d_step_u8 -> k_step2); since round 3 the rollout driver ends every
env step inside `k_act_step<...>` (fused.act_env_step: both LSTM cells + heads + draws + env step + observation), baked into
hipGraphs that are REPLAYED — under the pipelined schedule on a second HIP stream next to the learner's graph. This file
checks exactly that path against the C oracle (PHILOX mode, same seed, same global env ids):

  * the player is built the way bench.py / tools/config_sweep.py build it (make_player, byte observations, graphs on);
  * two eager iterations (what the drivers' own warm-up does), then >= 3 replayed iterations of GraphedIteration AND of
    PipelinedIteration (two streams; pairs of phases issued back to back so that learner i really runs beside rollout
    i + 1; at 512 envs also on the CU-partitioned stream pair), 200 env steps per case, so generator passes and episode
    switches happen INSIDE replays (one case with a 37-step TimeLimit, so that time-limit dones do too);
  * after every iteration the rollout store is read back — the policy's own sampled actions [T, 2, N], observations
    [T + 1, N, 2, 13, 13], rewards [T, N, 2], done [T, N] — and `oracle.OracleBatch` is stepped with those actions: every
    observation, reward (== float32(oracle float64)) and done flag of every env must be equal, slot 0 of a rollout must be
    the last observation of the one before, and at the end positions / far counters / step counters / episode numbers.

Sizes: the per-GPU shards of every BASELINE.json GPU configuration with the LAST rank's env_id_base (4096 PZR base 0 and
28 672; 512 base 3 584; 1024 Ram; 1024 MazeNav base 7 168; 2048 Adv 50/50 Block/Maze base 14 336).
Reference: train.py:81-88, player_util.py:44-67, envs/gym-track2d/gym_track2d/envs/track_1v1.py:71-127."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

T = 20


class _Case(object):
    def __init__(self, env_id, n, base, network, aux, mode, map_types=None, seed=1, max_steps=500):
        from active_tracking_rl_amd import registry
        from active_tracking_rl_amd.environment import VecEnv
        from active_tracking_rl_amd.train import default_args, make_player
        self.dev = torch.device("cuda:0")
        sp = registry.spec(env_id)
        over = {}
        mts = [sp["map_type"]] * n
        if map_types is not None:
            mts = list(map_types)
            over["map_type_per_env"] = np.array([registry.MAP_CODE[m] for m in mts], np.uint8)
        self.args = default_args(env=env_id, network=network, aux=aux, train_mode=mode, num_envs=n, num_steps=T, seed=seed,
                                 obs_u8=True)
        if max_steps != 500:
            over["max_episode_steps"] = max_steps         # (TimeLimit dones inside the replays, gym_track2d/__init__.py:17)
        env = VecEnv(env_id, n, device="cuda:0", seed=seed, env_id_base=base, obs_u8=True, **over)
        assert env.obs_u8, "byte observations are the production path of every 'Partial' id"
        self.player, self.opt = make_player(self.args, self.dev, 0, 1, env=env)     # (env.reset() happens in here)
        self.env, self.n, self.mode = env, n, mode
        self.oracle = orc.OracleBatch([orc.OracleEnv(mts[i], sp["target_mode"], sp["level"], max_steps, orc.RNG_PHILOX, seed, base + i)
                                       for i in range(n)])
        want = self.oracle.reset()
        got = self.player.state.reshape(n, 2, 13, 13).cpu().numpy()
        assert np.array_equal(got, want), "reset observations"
        self.last = want
        self.steps = self.dones = 0

    def check(self, agent, what):
        """Replay the rollout `agent` has just stored through the oracle."""
        torch.cuda.synchronize(self.dev)
        obs = agent._buf[0].cpu().numpy()
        rew, done = agent._buf[1].cpu().numpy(), agent._buf[2].cpu().numpy()
        acts = agent._cache.actions.cpu().numpy()                       # [T, 2, N], the policy's own draws
        assert obs.shape == (T + 1, self.n, 2, 13, 13) and obs.dtype == np.uint8 and acts.shape == (T, 2, self.n)
        assert acts.min() >= 0 and acts.max() <= 3
        assert np.array_equal(obs[0], self.last), (what, "slot 0 is not the observation the previous rollout ended on")
        for t in range(T):
            wo, wr, wd = self.oracle.step(np.ascontiguousarray(acts[t].T))
            bad = np.nonzero((obs[t + 1] != wo).any((1, 2, 3)))[0]
            assert bad.size == 0, (what, self.steps + t, "observations", bad[:8])
            assert np.array_equal(rew[t], wr.astype(np.float32)), (what, self.steps + t, "rewards")
            assert np.array_equal(done[t], wd), (what, self.steps + t, "done")
            self.dones += int(wd.sum())
        self.last = obs[T].copy()
        self.steps += T

    def final(self, min_done):
        st = self.env.core.get_state()
        for i in (list(range(0, self.n, max(1, self.n // 97))) + [self.n - 1]):
            s = self.oracle.envs[i].state()
            assert np.array_equal(st["pos"][i], s["pos"]) and st["c_far"][i] == s["c_far"] and st["t"][i] == s["t"], i
            assert st["episode"][i] == self.oracle.L.orc_episode(self.oracle.envs[i].h), i
        assert self.env.core.faults() == 0
        assert self.dones >= min_done, (self.dones, min_done)
        assert self.player.model.env_step_fused_seen or any(
            getattr(p.model, "env_step_fused_seen", False) for p in getattr(self, "replicas", [])), \
            "the env step did not run inside k_act_step: this test would not be covering the timed region"

    def eager(self, iters=2):
        """What the drivers' warm-up does (their own is switched off below so that these rollouts can be checked too)."""
        from active_tracking_rl_amd.train import rollout
        for i in range(iters):
            rollout(self.player, T)
            self.check(self.player, "eager %d" % i)
            self.player.optimize(None, self.opt, self.player.model, self.mode, self.dev)

    def close(self):
        self.env.close()


def _synchronous(case, iters=8):
    from active_tracking_rl_amd.train import GraphedIteration
    case.eager()
    it = GraphedIteration(case.player, case.opt, case.args, warmup=0)
    for i in range(iters):
        it.run()
        case.check(case.player, "graph replay %d" % i)
    assert torch.isfinite(case.opt.bucket.flat).all()


def _pipelined(case, pairs=4, cu_partition=False):
    from active_tracking_rl_amd.train import PipelinedIteration, cu_masked_stream
    case.eager()
    it = PipelinedIteration(case.player, case.opt, case.args, warmup=0)
    case.replicas = it.players
    assert not it.serial
    if cu_partition:
        try:
            it.sR, it.sL = cu_masked_stream(case.dev, 128, 128), cu_masked_stream(case.dev, 0, 128)
        except Exception as ex:           # a runtime without CU masks: the shared-chip pair is still two streams
            print("CU-masked streams unavailable: %r" % (ex,))
    for j in range(pairs):
        # two phases back to back: rollout 2j on stream R, then learner 2j on stream L BESIDE rollout 2j + 1; the third call
        # issues learner 2j + 1 beside rollout 2j + 2 ... finish() joins. Replica k's stores hold rollout 2j + k.
        it.run()
        it.run()
        it.finish()
        case.check(it.players[0], "pipelined rollout %d" % (2 * j))
        case.check(it.players[1], "pipelined rollout %d" % (2 * j + 1))
    assert torch.isfinite(case.opt.bucket.flat).all()


CASES = {
    "pzr4096_rank0": (("Track2D-BlockPartialPZR-v0", 4096, 0, "tat-maze-lstm", "reward", -1), 1500),
    "pzr4096_rank7": (("Track2D-BlockPartialPZR-v0", 4096, 28672, "tat-maze-lstm", "reward", -1), 1500),
    "pzr512_rank7": (("Track2D-BlockPartialPZR-v0", 512, 3584, "tat-maze-lstm", "reward", -1), 150),
    "ram1024": (("Track2D-BlockPartialRam-v0", 1024, 0, "maze-lstm", "none", 0), 200),
    "mazenav1024_rank7": (("Track2D-MazePartialNav-v0", 1024, 7168, "maze-lstm", "none", 0), 200),
    "adv2048_mixed_rank7": (("Track2D-BlockPartialAdv-v0", 2048, 14336, "maze-lstm", "none", -1), 500),
    # a 37-step TimeLimit: time-limit dones (and the generator passes they trigger) in every replayed rollout
    "pzr1024_timelimit37": (("Track2D-BlockPartialPZR-v0", 1024, 1024, "tat-maze-lstm", "reward", -1), 2500),
}


def _case(name):
    spec, min_done = CASES[name]
    mts = None
    if "mixed" in name:
        mts = ["Block" if i % 2 == 0 else "Maze" for i in range(spec[1])]
    return _Case(*spec, map_types=mts, max_steps=37 if "timelimit37" in name else 500), min_done


@pytest.mark.parametrize("name", sorted(CASES))
def test_synchronous_graph_replays_match_the_oracle(name):
    case, min_done = _case(name)
    try:
        _synchronous(case)
        case.final(min_done)
    finally:
        case.close()


@pytest.mark.parametrize("name", sorted(CASES))
def test_pipelined_two_stream_replays_match_the_oracle(name):
    case, min_done = _case(name)
    try:
        _pipelined(case)
        case.final(min_done)
    finally:
        case.close()


@pytest.mark.parametrize("name", ["pzr512_rank7", "ram1024", "mazenav1024_rank7", "pzr1024_timelimit37"])
def test_cooperative_two_launch_step_matches_the_oracle(name):
    """The strong-scaling shards with the rollout step as TWO launches (stem + k_coop_step: fc pair, LSTMCell GEMM, cells, heads,
    draws and env step in one launch whose workgroups cooperate per XCD — off by default, ATR_COOP_STEP=1): replayed synchronous
    iterations against the oracle, and no placement / barrier / shape fault raised on the device."""
    case, min_done = _case(name)
    try:
        case.player.model.coop_step = True
        _synchronous(case)
        assert case.player.model.coop_step_seen
        assert case.env.core.faults() == 0
        case.final(min_done)
    finally:
        case.close()


@pytest.mark.parametrize("name,schedule", [("pzr4096_rank7", "synchronous"), ("pzr4096_rank0", "pipelined"),
                                           ("adv2048_mixed_rank7", "synchronous"), ("pzr1024_timelimit37", "pipelined"),
                                           ("mazenav1024_rank7", "synchronous"), ("ram1024", "pipelined")])
def test_gate_product_with_the_cell_in_its_epilogue_matches_the_oracle(name, schedule):
    """Round 6: the one-GEMM step's LSTMCell product as this build's own MFMA kernel with the tracker's cell as its epilogue
    (atr_gate_cell, csrc/gate_cell_hip.hip; ATR_GATE_CELL=1) and k_act_step told that the tracker's hidden row is already there
    (ig[0] = NULL): the replayed timed region — synchronous and pipelined graphs, tat and maze-lstm pairs, Ram / Nav targets, a
    37-step TimeLimit, the last rank's env ids — against the oracle, every observation / reward / done of every env."""
    case, min_done = _case(name)
    try:
        m = case.player.model
        m.gate_cell_kernel, m.gate_cell_min_rows = True, 1024
        (_synchronous if schedule == "synchronous" else _pipelined)(case)
        assert m.gate_cell_seen or schedule == "pipelined"       # (the pipelined schedule's replicas do the stepping)
        assert case.env.core.faults() == 0
        case.final(min_done)
    finally:
        case.close()


def test_cooperative_step_on_its_own_half_of_the_cus_matches_the_oracle_512():
    """... and under the pipelined schedule on the CU-partitioned stream pair (the rollout's 128 CUs to itself: the only form of
    that schedule in which the cooperative step may run — train.PipelinedIteration._set_coop_grid)."""
    from active_tracking_rl_amd.train import PipelinedIteration, cu_masked_stream
    case, min_done = _case("pzr512_rank7")
    try:
        case.player.model.coop_step = True
        case.eager()
        it = PipelinedIteration(case.player, case.opt, case.args, warmup=0)
        case.replicas = it.players
        try:
            it._use_streams(cu_masked_stream(case.dev, 128, 128), cu_masked_stream(case.dev, 0, 128))
        except Exception as ex:
            pytest.skip("CU-masked streams unavailable: %r" % (ex,))
        assert all(p.model.coop_step and p.model.coop_workgroups == 128 for p in it.players)
        for j in range(4):
            it.run()
            it.run()
            it.finish()
            case.check(it.players[0], "pipelined rollout %d" % (2 * j))
            case.check(it.players[1], "pipelined rollout %d" % (2 * j + 1))
        assert all(p.model.coop_step_seen for p in it.players) and case.env.core.faults() == 0
        case.final(min_done)
    finally:
        case.close()


def test_pipelined_cu_partitioned_streams_match_the_oracle_512():
    """The strong-scaling shard on the stream pair tune_streams() picks there: each chain on its own half of the CUs."""
    case, min_done = _case("pzr512_rank7")
    try:
        _pipelined(case, pairs=4, cu_partition=True)
        case.final(min_done)
    finally:
        case.close()
