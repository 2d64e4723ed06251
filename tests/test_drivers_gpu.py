"""-m gpu: the drop-in drivers on the HIP path — gym-protocol single env (Track2DEnv) against the oracle, a short
training run (eager and hipGraph), the vectorised evaluator, reference-named checkpoints and gym_eval.py."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def test_gym_protocol_single_env_matches_oracle():
    from active_tracking_rl_amd.environment import create_env
    from active_tracking_rl_amd.train import default_args
    args = default_args(num_envs=1, seed=7)
    env = create_env("Track2D-BlockPartialPZR-v0", args, num_envs=1)
    assert len(env.observation_space) == 2 and env.observation_space[0].shape == (1, 13, 13)
    assert env.action_space[0].n == 4
    o = orc.OracleEnv("Block", "PZR", 0, 500, orc.RNG_PHILOX, 7, 0)
    obs = env.reset()
    assert obs.shape == (2, 1, 1, 13, 13) and obs.dtype == np.float32
    assert np.array_equal(obs.reshape(2, 13, 13), o.reset().astype(np.float32))
    rs = np.random.RandomState(0)
    for t in range(80):
        a = [np.array(rs.randint(4)), np.array(rs.randint(4))]      # 0-d arrays like model.py:50 produces
        obs, rew, done, info = env.step(a)
        wo, wr, wd, _ = o.step([int(a[0]), int(a[1])])
        assert rew.dtype == np.float64 and isinstance(done, bool)
        assert np.array_equal(obs.reshape(2, 13, 13), wo.astype(np.float32))
        assert np.array_equal(rew, wr.astype(np.float32).astype(np.float64)) and done == wd
        assert abs(info["distance"] - np.sqrt(o.state()["d2"])) < 1e-12
        if done:
            obs = env.reset()
            assert np.array_equal(obs.reshape(2, 13, 13), o.reset().astype(np.float32))
    env.close()


def test_train_eager_and_graphed_then_evaluate_and_checkpoints(tmp_path):
    from active_tracking_rl_amd.model import build_model
    from active_tracking_rl_amd.test import evaluate, save_checkpoints, test
    from active_tracking_rl_amd.train import GraphedIteration, default_args, make_player, rollout
    dev = torch.device("cuda:0")
    args = default_args(num_envs=128, test_eps=6, log_dir=str(tmp_path), split=True, max_step=3,
                        env_base="Track2D-BlockPartialRam-v0")
    player, opt = make_player(args, dev)
    w0 = opt.bucket.flat.clone()
    rollout(player, args.num_steps)
    stats = player.optimize(None, opt, player.model, -1, dev)
    assert all(torch.isfinite(s).all() for s in stats)
    it = GraphedIteration(player, opt, args)
    for _ in range(3):
        it.run()
    torch.cuda.synchronize()
    assert torch.isfinite(opt.bucket.flat).all() and not torch.equal(w0, opt.bucket.flat)
    # env invariants still hold after graph replays (generator launches inside the captured rollout)
    st = player.env.core.get_state()
    maps = player.env.core.get_maps()
    idx = np.arange(128)
    assert (maps[idx, st["pos"][:, 0, 0], st["pos"][:, 0, 1]] == 0).all()
    assert (maps[:, 1:81, 1:81].reshape(128, -1).sum(1) <= 959).all()
    rsum, length = evaluate(player.model, "Track2D-BlockPartialNav-v0", args, dev, 6)
    assert rsum.shape == (6, 2) and (length >= 11).all() and (length <= 500).all()
    train_modes, n_iters = [-1], [5]
    test(args, player.model, train_modes, n_iters, rounds=1)
    assert train_modes[0] == -100                                    # n_iter > max_step -> stop sentinel
    names = sorted(os.listdir(str(tmp_path)))
    assert "all-best-5.dat" in names and "tracker-best.dat" in names and "target-best.dat" in names
    sd = torch.load(os.path.join(str(tmp_path), "tracker-best.dat"))
    assert "encoder.conv1.weight" in sd and tuple(sd["lstm.weight_ih"].shape) == (512, 256)
    m2 = build_model(player.env.observation_space, player.env.action_space, args, dev).to(dev)
    m2.load_state_dict(torch.load(os.path.join(str(tmp_path), "all-best-5.dat")))
    assert all(torch.equal(a, b) for a, b in zip(m2.state_dict().values(), player.model.state_dict().values()))
    # gym_eval.py loads the reference-named checkpoints and writes the CSV row
    csv_path = os.path.join(str(tmp_path), "eval.csv")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "gym_eval.py"), "--env", "Track2D-BlockPartialRam-v0",
                        "--num-episodes", "5", "--load-tracker", os.path.join(str(tmp_path), "tracker-best.dat"),
                        "--load-target", os.path.join(str(tmp_path), "target-best.dat"), "--log-dir",
                        str(tmp_path) + "/", "--csv", csv_path], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = open(csv_path).read().strip().splitlines()
    assert rows[0].startswith("Env,Seed,R_mean") and rows[1].startswith("Track2D-BlockPartialRam-v0,1,")
    player.env.close()


def test_frame_stack_matches_reference_wrapper_semantics():
    """environment.frame_stack (environment.py:128-156): deque(maxlen=stack) per agent, filled with the first frame
    on reset, newest frame last. Checked against the oracle's single frames for stack_frames = 3, gym protocol."""
    from collections import deque
    from active_tracking_rl_amd.environment import create_env
    from active_tracking_rl_amd.train import default_args
    args = default_args(num_envs=1, seed=11, stack_frames=3)
    env = create_env("Track2D-BlockPartialAdv-v0", args, num_envs=1)
    o = orc.OracleEnv("Block", "Adv", 0, 500, orc.RNG_PHILOX, 11, 0)
    first = o.reset().astype(np.float32)
    frames = [deque([first[i][None]] * 3, maxlen=3) for i in range(2)]
    obs = env.reset()
    assert obs.shape == (2, 3, 1, 13, 13)
    assert np.array_equal(obs, np.array([np.stack(frames[i], 0) for i in range(2)]))
    rs = np.random.RandomState(2)
    for t in range(12):
        a = rs.randint(0, 4, 2)
        obs, rew, done, info = env.step(list(a))
        wo, wr, wd, _ = o.step(a)
        for i in range(2):
            frames[i].append(wo[i].astype(np.float32)[None])
        assert np.array_equal(obs, np.array([np.stack(frames[i], 0) for i in range(2)])), t
    env.close()
    # the batched model accepts stacked frames (encoder folds them into the batch): tracker 3 frames, tat target 6
    from active_tracking_rl_amd.model import build_model
    from active_tracking_rl_amd.environment import _spaces
    obs_s, act_s = _spaces()
    m = build_model(obs_s, act_s, args, torch.device("cuda")).cuda()
    assert tuple(m.player0.encoder.fc.weight.shape) == (256, 512 * 3) and tuple(m.player1.encoder.fc.weight.shape) == (256, 512 * 6)
    st = torch.rand(5, 2, 3, 1, 13, 13, device="cuda")
    v, a, e, lp, (h, c), rp = m((st, (torch.zeros(5, 2, 128, device="cuda"), torch.zeros(5, 2, 128, device="cuda"))))
    assert v.shape == (5, 2, 1) and rp.shape == (5, 1)


def test_train_worker_loop_reference_signature(tmp_path):
    """train(rank, args, shared_model, optimizer, train_modes, n_iters, env=None) — train.py:15 — runs, counts its
    iterations in n_iters[rank] and stops by the test.py:129-134 rule (sum of iterations > max_step)."""
    from active_tracking_rl_amd.train import default_args, train
    args = default_args(num_envs=64, max_step=4, log_dir=str(tmp_path), env="Track2D-MazePartialFar-v1")
    train_modes, n_iters = [], []
    player = train(0, args, None, None, train_modes, n_iters)
    assert n_iters == [5] and train_modes == [-1]
    assert player.n_steps == 5 * args.num_steps
    assert torch.isfinite(torch.cat([p.reshape(-1) for p in player.model.parameters()])).all()


def test_full_observation_env_trains():
    """A 'Full' id end to end: VecEnv spaces are (1, 82, 82), the policy falls back to F.conv2d for the big frames,
    one rollout + update runs."""
    from active_tracking_rl_amd.train import default_args, make_player, rollout
    dev = torch.device("cuda:0")
    args = default_args(env="Track2D-BlockFullAdv-v0", network="maze-lstm", aux="none", num_envs=16, num_steps=4)
    player, opt = make_player(args, dev)
    assert player.env.observation_space[0].shape == (1, 82, 82)
    assert tuple(player.model.player0.encoder.fc.weight.shape) == (256, 32 * 21 * 21)
    rollout(player, args.num_steps)
    stats = player.optimize(None, opt, player.model, -1, dev)
    assert all(torch.isfinite(s).all() for s in stats)
    # the pipelined driver keeps such ids on ONE stream (MIOpen's kernels next to another stream's graph hung the device)
    from active_tracking_rl_amd.train import PipelinedIteration
    it = PipelinedIteration(player, opt, args)
    assert it.serial and it.tune_streams() == []
    for _ in range(3):
        it.run()
    it.finish()
    torch.cuda.synchronize()
    assert torch.isfinite(opt.bucket.flat).all()
    player.env.close()


@pytest.mark.parametrize("train_mode,path", [(-1, "pair"), (0, "pair"), (1, "pair"), (-1, "cat"), (1, "cat"), (-1, "coop"),
                                             (0, "coop"), (-1, "coop512")])
def test_cached_rollout_learner_matches_the_recompute_learner(train_mode, path):
    """Actor/learner with the rollout cache (forward evaluated once, in the rollout: model.act_cached +
    forward_sequence_cached) against the recompute learner (forward_sequence) on the SAME rollout: loss terms and
    every parameter gradient agree to fp32 round-off (different GEMM shapes -> different summation orders). Train-mode
    0 / 1: the cached learner does not back-propagate the untrained player's recurrence — its parameters get no (or an
    all-zero) gradient from either learner. The rollout step's three forms: "pair" (small batch: fc pair + gate pair kernels),
    "cat" (the large-batch form forced at a small batch: fc + ReLU written into [features | k h_prev] rows by atr_linear, both
    LSTMCell GEMMs as one K = 384 product, the masked hidden rows written by k_act_step; the learner then reads the features
    in place, strided: atr_relu_backward_ld, atr_embed_add_ld, the grouped weight-gradient launch with a row stride) and
    "coop" (the strong-scaling shards' ONE launch after the stem, k_coop_step: its fc and gate tiles come from
    csrc/coop_gemm.h — this comparison against plain PyTorch fp32 layers is that kernel's policy-half check)."""
    from active_tracking_rl_amd.train import default_args, make_player, rollout
    cat_gemm = path != "pair"
    n_envs = {"pair": 256, "cat": 1024, "coop": 1024, "coop512": 512}[path]
    args = default_args(env="Track2D-BlockPartialPZR-v0", num_envs=n_envs, num_steps=6,
                        network="tat-maze-lstm", seed=11, train_mode=train_mode)
    args.gpu_ids = [0]
    player, optimizer = make_player(args, torch.device("cuda:0"), 0, 1)
    assert player.cache_rollout
    player.model.coop_step = path.startswith("coop")        # (off by default: the four-launch step won the measurement)
    if path == "cat":
        player.model.pair_gemm_max_rows = 0
    rollout(player, args.num_steps, fast=True)
    assert player._cache is not None and (player._cache.fh_all is not None) == cat_gemm
    assert bool(player.model.coop_step_seen) == path.startswith("coop")
    assert player.env.core.faults() == 0
    if cat_gemm:      # the rows the one-GEMM path reads: features next to the PREVIOUS step's hidden row, masked by its done flag
        cch, T_ = player._cache, args.num_steps
        keep = (player._buf[2] == 0).float()                                          # [T, N]
        for t in range(1, T_ + 1):
            want = cch.h_all[:, t] * keep[t - 1].view(1, -1, 1)
            assert torch.equal(cch.fh_all[:, t, :, 256:], want), t
        assert torch.equal(cch.fh_all[:, 0, :, 256:], cch.h_all[:, 0])
        assert not cch.f[0].is_contiguous() and cch.f[0].stride(-2) == 384
    outs = []
    cache = player._cache
    for cached, fused_heads in ((True, True), (True, False), (False, False)):
        player._cache = cache if cached else None
        player.fused_heads = fused_heads
        # the bootstrap value of the tracker-aware target depends on a freshly SAMPLED tracker action: pin the draw
        # (and take the bootstrap forward all three learners share: the fused one is checked in the next test)
        torch.manual_seed(5)
        if getattr(player.model, "_sampler", None) is not None:
            player.model._sampler.counter.zero_()
            player.model._sampler._last = None
        loss, pl, vl, ent, pred = player.loss_recompute(args.train_mode)
        player._cache, player.fused_heads = cache, True
        params = [p for p in player.model.parameters()]
        grads = torch.autograd.grad(loss, params, allow_unused=True)
        outs.append((loss.detach(), pl.detach().mean(0), vl.detach().mean(0), ent.mean(0), pred.detach().mean(0), grads))
    for k in (0, 1):                                      # fused heads, cached -- each against the recompute learner
        la, pla, vla, ea, pa, ga = outs[k]
        lb, plb, vlb, eb, pb, gb = outs[2]
        torch.testing.assert_close(la, lb, rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(pla, plb, rtol=1e-3, atol=1e-4)
        torch.testing.assert_close(vla, vlb, rtol=1e-3, atol=1e-4)
        torch.testing.assert_close(ea, eb, rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(pa.reshape(-1), pb.reshape(-1), rtol=1e-4, atol=1e-4)
        _check_grads(ga, gb)


@pytest.mark.parametrize("env_id,kw", [("Track2D-BlockPartialRPF-v0", {}), ("Track2D-BlockPartialPZR-v0", {"rescale": True}),
                                       ("Track2D-BlockPartialPZR-v0", {"stack_frames": 2})])
def test_large_batch_rollout_with_a_separate_env_step_keeps_the_two_gemm_cell(env_id, kw):
    """Envs whose step cannot run inside k_act_step (RPF targets, --rescale, stacked frames: VecEnv.fused_step_out is None)
    at a batch above cat_gemm_min_rows: nobody would write the masked hidden columns of the one-GEMM [features | k h_prev]
    rows, so the cache must keep the two-GEMM form — and the cached learner must agree with the recompute learner, which
    re-runs the recurrence from the rollout's first LSTM state and shares nothing with the cache."""
    from active_tracking_rl_amd.train import default_args, make_player, rollout
    args = default_args(env=env_id, num_envs=1024, num_steps=6, network="tat-maze-lstm", seed=7, train_mode=-1, **kw)
    args.gpu_ids = [0]
    player, optimizer = make_player(args, torch.device("cuda:0"), 0, 1)
    assert player.model.cat_gate_gemm and args.num_envs >= player.model.cat_gemm_min_rows
    rollout(player, args.num_steps, fast=True)
    if player._cache is None:            # (stacked frames: no in-place rollout store, the recompute learner is the only one)
        assert kw.get("stack_frames", 1) > 1 or kw.get("rescale")
        player.env.close()
        return
    assert player._cache.fh_all is None
    assert not player.model.env_stepped
    outs = []
    cache = player._cache
    for cached in (True, False):
        player._cache = cache if cached else None
        torch.manual_seed(5)
        if getattr(player.model, "_sampler", None) is not None:
            player.model._sampler.counter.zero_()
            player.model._sampler._last = None
        loss, pl, vl, ent, pred = player.loss_recompute(args.train_mode)
        player._cache = cache
        grads = torch.autograd.grad(loss, list(player.model.parameters()), allow_unused=True)
        outs.append((loss.detach(), grads))
    torch.testing.assert_close(outs[0][0], outs[1][0], rtol=1e-4, atol=1e-5)
    _check_grads(outs[0][1], outs[1][1])
    player.env.close()


def test_one_gemm_cell_refuses_a_step_without_the_fused_env_step():
    """model._act_step on a cache of the one-GEMM form without an env_out: a loud error, not stale hidden rows."""
    from active_tracking_rl_amd.train import default_args, make_player, rollout
    args = default_args(env="Track2D-BlockPartialPZR-v0", num_envs=1024, num_steps=3, network="tat-maze-lstm", seed=7)
    args.gpu_ids = [0]
    player, _ = make_player(args, torch.device("cuda:0"), 0, 1)
    rollout(player, args.num_steps, fast=True)
    cache = player._cache
    assert cache.fh_all is not None
    player.model._sampler.reopen_block()
    with pytest.raises(RuntimeError, match="env step inside k_act_step"):
        player.model.act_cached(player.state, cache, 0, None, env_out=None)
    player.model._sampler.end_block()
    player.env.close()


@pytest.mark.parametrize("schedule", ["synchronous", "pipelined"])
def test_burn_in_moves_the_envs_and_nothing_else(schedule):
    """burn_in(k): k iterations whose updates are discarded (what decorrelates the episode phases of a fresh shard before
    training starts — profiles/r05_learning_seeds_*.txt): weights, optimizer state, replica copies, phase and step counters as
    before; the env shard (episode clocks) has moved on; the next real iteration trains."""
    from active_tracking_rl_amd.train import GraphedIteration, PipelinedIteration, default_args, make_player
    dev = torch.device("cuda:0")
    args = default_args(env="Track2D-BlockPartialRam-v0", network="maze-lstm", aux="none", train_mode=0, num_envs=256, seed=5)
    player, opt = make_player(args, dev)
    it = PipelinedIteration(player, opt, args, serial=True) if schedule == "pipelined" else GraphedIteration(player, opt, args)
    tensors = it._schedule_tensors() if schedule == "pipelined" else it._optimizer_tensors()
    before = [t.clone() for t in tensors]
    n0, i0 = player.n_steps, getattr(it, "i", 0)
    t0 = player.env.core.get_state()["t"].copy() if hasattr(player.env.core, "get_state") else None
    it.burn_in(7)
    torch.cuda.synchronize()
    for t, v in zip(tensors, before):
        assert torch.equal(t, v)
    assert player.n_steps == n0 and getattr(it, "i", 0) == i0
    if t0 is not None:
        assert not np.array_equal(player.env.core.get_state()["t"], t0)       # the episode clocks have moved
    it.run()
    if schedule == "pipelined":
        it.run()
        it.finish()
    torch.cuda.synchronize()
    assert not torch.equal(opt.bucket.flat, before[0])
    player.env.close()


def _transplant_rollout(src, dst, path="player", seen=None):
    """Copy every tensor reachable from agent `src`'s rollout state (attributes, lists / tuples / dicts of tensors, the
    RolloutCache) into the tensor at the same attribute path of agent `dst` — the static buffers its captured learner reads."""
    from active_tracking_rl_amd.model import RolloutCache
    seen = set() if seen is None else seen
    n = 0
    if torch.is_tensor(src):
        assert torch.is_tensor(dst) and dst.shape == src.shape and dst.dtype == src.dtype, path
        if dst.data_ptr() != src.data_ptr() and (dst.data_ptr(), dst.shape) not in seen:
            seen.add((dst.data_ptr(), tuple(dst.shape)))
            with torch.no_grad():
                dst.copy_(src.detach())
            n = 1
        return n
    if isinstance(src, (list, tuple)):
        assert isinstance(dst, (list, tuple)) and len(dst) == len(src), path
        return sum(_transplant_rollout(a, b, "%s[%d]" % (path, i), seen) for i, (a, b) in enumerate(zip(src, dst)))
    if isinstance(src, dict):
        return sum(_transplant_rollout(v, dst[k], "%s[%r]" % (path, k), seen) for k, v in src.items()
                   if (torch.is_tensor(v) or isinstance(v, (list, tuple, dict, RolloutCache))) and k in dst)
    if isinstance(src, RolloutCache):
        assert isinstance(dst, RolloutCache), path
        for k in list(src.__dict__):
            if k in ("lazy", "consts", "boot"):       # (weight-derived constants; the bootstrap scratch is written before it is read)
                continue
            if k in dst.__dict__:
                n += _transplant_rollout(src.__dict__[k], dst.__dict__[k], path + "." + k, seen)
        return n
    return 0


@pytest.mark.parametrize("env_id,network,aux,n_envs,train_mode", [
    ("Track2D-BlockPartialPZR-v0", "tat-maze-lstm", "reward", 512, -1),
    ("Track2D-BlockPartialPZR-v0", "tat-maze-lstm", "reward", 512, 0),
    ("Track2D-BlockPartialPZR-v0", "tat-maze-lstm", "reward", 512, 1),
    ("Track2D-MazePartialNav-v0", "maze-lstm", "none", 1024, 0),
    ("Track2D-BlockPartialAdv-v0", "maze-lstm", "none", 1024, -1),
    ("Track2D-BlockPartialPZR-v0", "tat-maze-lstm", "reward", 1024, -1)])
def test_the_two_schedules_learners_give_the_same_gradient_from_the_same_rollout(monkeypatch, env_id, network, aux, n_envs,
                                                                                train_mode):
    """GraphedIteration (synchronous) and PipelinedIteration differ in WHEN a gradient is applied, and must differ in nothing
    else: their captured learners — the master agent's loss + backward inside the synchronous rollout graph, a replica agent's
    learner graph on its own stores and flat weight buffer — are handed the SAME rollout (every store of the master's rollout
    copied into the replica's static buffers, same weights, same draw-stream position for the bootstrap step) and must produce
    the SAME gradient bucket bit for bit. (What round 5's learning tables could not say: whether the no-delay schedule's
    weaker runs on configs[3] come from a defect in its learner — carry, bootstrap, select_params — or from the schedule.)
    Both tat (the bootstrap draws a tracker action) and maze-lstm networks, every training mode, the pair-kernel (512 envs) and
    one-GEMM (1024 envs) rollout stores. The pipelined learner's co-run dW form (another split-K plan) is switched off: it
    changes the summation order by design."""
    from active_tracking_rl_amd.train import GraphedIteration, PipelinedIteration, default_args, make_player
    monkeypatch.setenv("ATR_PIPE_CORUN", "0")
    dev = torch.device("cuda:0")
    args = default_args(env=env_id, network=network, aux=aux, train_mode=train_mode, num_envs=n_envs, seed=19)
    args.gpu_ids = [0]
    player, opt = make_player(args, dev)
    # the pipelined schedule first: its constructor warms up eagerly on the master (whose attributes would otherwise stop
    # naming the synchronous graph's static buffers)
    it_p = PipelinedIteration(player, opt, args, mode=train_mode, serial=True)
    assert not it_p.corun
    it_g = GraphedIteration(player, opt, args, mode=train_mode)
    w0 = opt.bucket.flat.clone()
    for b in it_p.buckets:
        assert torch.equal(b.flat, w0)
    rep = it_p.players[0]
    rep.model._sampler.seed = player.model._sampler.seed           # the bootstrap step's draw: same Philox key ...
    it_p._capture(train_mode, 0)                                    # (the seed is a kernel argument of the captured launches)
    torch.cuda.synchronize()
    opt.bucket.grad.zero_()
    it_g.g_rolls[train_mode].replay()                               # rollout X + the synchronous learner -> opt.bucket.grad
    torch.cuda.synchronize()
    g_sync = opt.bucket.grad.clone()
    assert torch.equal(opt.bucket.flat, w0) and torch.isfinite(g_sync).all() and float(g_sync.abs().sum()) > 0
    moved = _transplant_rollout({k: v for k, v in vars(player).items() if k not in ("model", "env", "args")},
                                {k: v for k, v in vars(rep).items() if k not in ("model", "env", "args")})
    assert moved >= 8, moved
    rep.model._sampler.counter.copy_(player.model._sampler.counter)   # ... and the same counter value
    opt.bucket.grad.zero_()
    it_p.buckets[0].grad.zero_()
    it_p.graphs[(train_mode, 0)][1].replay()                        # the pipelined learner of replica 0 on the transplanted rollout
    torch.cuda.synchronize()
    assert torch.equal(it_p.buckets[0].grad, opt.bucket.grad)       # (the learner graph ends by handing its bucket to the optimizer's)
    diff = (it_p.buckets[0].grad - g_sync).abs().max().item()
    assert torch.equal(it_p.buckets[0].grad, g_sync), "gradient buckets differ: max |diff| %.3e" % diff
    player.env.close()


@pytest.mark.parametrize("train_mode", [-1, 0, 1])
def test_keeping_the_gate_gemm_output_instead_of_the_activated_gates_changes_no_bit(monkeypatch, train_mode):
    """One-GEMM rollout path, round 5: the rollout keeps the gate GEMM's OUTPUT per step (pre-activations without bias) and the
    learner's BPTT kernel re-activates it (atr_lstm_bptt_pre: ((pre + bias) + emb[a_tracker]) then the cell's own sigmoid / tanh) —
    against the round-4 form that stores the activated gates from k_act_step: same seeds, same rollout (observations, actions,
    LSTM states equal bit for bit), and the SAME GRADIENT bit for bit, for every training mode (mode 1: the tracker-aware
    target's embedding row alone in its group)."""
    from active_tracking_rl_amd import fused
    from active_tracking_rl_amd.train import default_args, make_player, rollout
    res = []
    monkeypatch.setattr(fused, "fold_embedding", False)     # (the folded embedding — pre-activation path only — re-associates dW_ih)
    for keep_pre in (True, False):
        args = default_args(env="Track2D-BlockPartialPZR-v0", num_envs=1024, num_steps=8, network="tat-maze-lstm", seed=23,
                            train_mode=train_mode)
        args.gpu_ids = [0]
        player, opt = make_player(args, torch.device("cuda:0"), 0, 1)
        player.model.store_preacts = keep_pre
        player.model.pair_gemm_max_rows = 0
        rollout(player, args.num_steps, fast=True)
        c = player._cache
        assert (c.pre_all is not None) == keep_pre and (c.acts is None) == keep_pre and c.fh_all is not None
        acts_buf, obs = player._actions_buf.clone(), player._buf[0].clone()
        h_all, c_all = c.h_all.clone(), c.c_all.clone()
        player.compute_grads(opt, train_mode)
        torch.cuda.synchronize()
        res.append((acts_buf, obs, h_all, c_all, opt.bucket.grad.clone()))
        player.env.close()
    for a, b in zip(res[0][:4], res[1][:4]):
        assert torch.equal(a, b)
    assert torch.isfinite(res[0][4]).all() and float(res[0][4].abs().sum()) > 0
    assert torch.equal(res[0][4], res[1][4])


@pytest.mark.parametrize("train_mode,n_envs,steps", [(-1, 1024, 8), (1, 1024, 8), (0, 1024, 8), (-1, 4096, 20), (-1, 1024, 3)])
def test_folded_tracker_action_embedding_gives_the_gradient_of_the_explicit_one(monkeypatch, train_mode, n_envs, steps):
    """Round 6: on the one-GEMM rollout path the learner no longer materialises f + fc_action_tracker(one_hot(a_tracker)) for the
    target's dW_ih product nor gathers dL/df by action for the embedding's gradient (two passes over [T N, 256] tensors each):
    the BPTT launch also leaves the column sums of dG by the row's tracker action, S [4, 512], and atr_embed_fold makes
    dW_ih += S^T E and d fc_action_tracker = S W_ih from them. Same rollout, fold on / off: every other gradient bit-identical,
    the target's weight_ih and fc_action_tracker equal to summation-order round-off (checked against a float64 evaluation of
    the same expressions from the explicit path's own dG is not needed: the explicit path IS the previous round's tested one).
    (1024 envs x 3 steps = 3072 rows: below the grouped weight-gradient launch's threshold — the fold then follows the product on
    the spot instead of as the group's post-flush hook.)"""
    from active_tracking_rl_amd import fused
    from active_tracking_rl_amd.train import default_args, make_player, rollout
    res = []
    for fold in (True, False):
        monkeypatch.setattr(fused, "fold_embedding", fold)
        args = default_args(env="Track2D-BlockPartialPZR-v0", num_envs=n_envs, num_steps=steps, network="tat-maze-lstm", seed=29,
                            train_mode=train_mode)
        args.gpu_ids = [0]
        player, opt = make_player(args, torch.device("cuda:0"), 0, 1)
        player.model.pair_gemm_max_rows = 0
        rollout(player, args.num_steps, fast=True)
        assert player._cache.pre_all is not None
        acts_buf = player._actions_buf.clone()
        opt.bucket.grad.zero_()
        player.compute_grads(opt, train_mode)
        torch.cuda.synchronize()
        named = {n: p.grad.clone() if p.grad is not None else None for n, p in player.model.named_parameters()}
        res.append((acts_buf, opt.bucket.grad.clone(), named))
        player.env.close()
    assert torch.equal(res[0][0], res[1][0])                 # the same rollout
    touched = ("player1.lstm.weight_ih", "player1.fc_action_tracker.weight", "player1.fc_action_tracker.bias")
    for name, g_fold in res[0][2].items():
        g_ref = res[1][2][name]
        if g_ref is None or g_fold is None:
            assert g_ref is None and g_fold is None, name
            continue
        if name in touched and train_mode != 0:
            scale = float(g_ref.abs().max()) + 1e-12
            assert float((g_fold - g_ref).abs().max()) <= 2e-5 * scale + 1e-9, (name, float((g_fold - g_ref).abs().max()), scale)
            assert float(g_ref.abs().max()) > 0, name
        else:
            assert torch.equal(g_fold, g_ref), name
    assert torch.isfinite(res[0][1]).all()


@pytest.mark.parametrize("n", [4096, 1000, 130])
def test_gate_product_with_cell_epilogue_against_plain_pytorch(n):
    """atr_gate_cell (csrc/gate_cell_hip.hip): pre = [features | k h_prev] [W_ih | W_hh]^T for both players against torch.bmm
    in float64 (2e-5 of the row norm bound), the cell epilogue against nn.LSTMCell fp32 semantics evaluated in float64 — gates
    (i, f, g, o), c' = f (k c) + i g, h' = o tanh(c') — to 2e-5 (the hardware exp / rcp forms of csrc/atr_cell.h), ragged row
    counts (rows past N are neither read beyond the last row nor written), and a player without the cell leaves h / c alone."""
    from active_tracking_rl_amd import fused
    dev = torch.device("cuda:0")
    torch.manual_seed(n)
    R, Fd = 128, 256
    K = Fd + R
    fh_store = torch.randn(2, n, K + 32, device=dev) * 0.5           # (row stride > K: a column block of wider rows)
    fh = fh_store[:, :, :K]
    w = torch.randn(2, 4 * R, K, device=dev) * 0.05
    bias = [torch.randn(4 * R, device=dev) * 0.1 for _ in range(2)]
    c_prev = [torch.randn(n, R, device=dev) for _ in range(2)]
    done = (torch.rand(n, device=dev) < 0.2).to(torch.uint8)
    pre = torch.full((2, n, 4 * R), float("nan"), device=dev)
    guard = torch.full((2, 4 * R), 5.0, device=dev)                   # (allocated right behind: rows past N must not land here)
    h = [torch.full((n, R), float("nan"), device=dev) for _ in range(2)]
    c = [torch.full((n, R), float("nan"), device=dev) for _ in range(2)]
    fused.gate_cell(fh, w, bias, pre, c_prev, done, h, c, cell=(True, False))
    torch.cuda.synchronize()
    ref = torch.bmm(fh.double(), w.double().transpose(1, 2))
    assert float((pre.double() - ref).abs().max()) < 2e-5
    assert torch.isnan(h[1]).all() and torch.isnan(c[1]).all()       # player 1: product only
    assert torch.equal(guard, torch.full_like(guard, 5.0))
    g = pre[0].double() + bias[0].double()                            # (the cell consumes the kernel's own fp32 product)
    i_, f_, g_, o_ = g[:, :R].sigmoid(), g[:, R:2 * R].sigmoid(), g[:, 2 * R:3 * R].tanh(), g[:, 3 * R:].sigmoid()
    k = (done == 0).double().unsqueeze(1)
    cn = f_ * (k * c_prev[0].double()) + i_ * g_
    hn = o_ * cn.tanh()
    assert float((c[0].double() - cn).abs().max()) < 2e-5 and float((h[0].double() - hn).abs().max()) < 2e-5
    fused.gate_cell(fh, w, bias, pre, c_prev, done, h, c, cell=(True, True))
    torch.cuda.synchronize()
    assert not torch.isnan(h[1]).any() and not torch.isnan(c[1]).any()


def test_mfma_actor_step_keeps_its_activated_gates_store_when_preacts_are_on():
    """ATR_MFMA_MIN_ROWS <= N (bench.py --actor-step mfma) with the default store_preacts: _act_step takes the per-player MFMA
    actor step, which writes acts[i] — new_cache must not have dropped that store for the one-GEMM step's pre-activation store
    (one predicate, model._env_fused_static, decides both). The rollout and the cached learner run and agree with the
    recompute learner."""
    from active_tracking_rl_amd.train import default_args, make_player, rollout
    args = default_args(env="Track2D-BlockPartialPZR-v0", num_envs=1024, num_steps=5, network="tat-maze-lstm", seed=13, train_mode=-1)
    args.gpu_ids = [0]
    player, opt = make_player(args, torch.device("cuda:0"), 0, 1)
    m = player.model
    m.fused_actor_step, m.mfma_step_min_rows = True, 768
    assert m.store_preacts and not m._env_fused_static(1024, 128)
    rollout(player, args.num_steps, fast=True)
    c = player._cache
    assert c is not None and c.acts is not None and c.pre_all is None and c.fh_all is None
    outs = []
    for cached in (True, False):
        player._cache = c if cached else None
        torch.manual_seed(5)
        m._sampler.counter.zero_()
        m._sampler._last = None
        loss, pl, vl, ent, pred = player.loss_recompute(args.train_mode)
        player._cache = c
        grads = torch.autograd.grad(loss, list(m.parameters()), allow_unused=True)
        outs.append((loss.detach(), grads))
    torch.testing.assert_close(outs[0][0], outs[1][0], rtol=1e-4, atol=1e-5)
    _check_grads(outs[0][1], outs[1][1])
    player.env.close()


def test_bootstrap_values_and_rollout_bookkeeping_kernels():
    """The learner's bootstrap V(s_T) through the rollout's fused kernels (model.boot_values: one more actor step into
    scratch + critic heads) against the plain model pieces evaluated with the tracker action it drew; the rollout
    prologue / epilogue kernels (LSTM state hand-over, episode lengths, keep mask) against the tensor expressions they
    replace — on a rollout with episode ends (max_episode_steps=7)."""
    import torch.nn.functional as F
    from active_tracking_rl_amd.environment import VecEnv
    from active_tracking_rl_amd.train import default_args, make_player, rollout
    for n in (256, 3072):                                  # below / above the MFMA actor-step threshold
        args = default_args(env="Track2D-BlockPartialPZR-v0", num_envs=n, num_steps=7, network="tat-maze-lstm", seed=3)
        env = VecEnv(args.env, n, device="cuda:0", seed=3, obs_u8=True, max_episode_steps=7)   # time limit at every 7th step
        player, _ = make_player(args, torch.device("cuda:0"), 0, 1, env=env)
        rollout(player, args.num_steps, fast=True)         # a first rollout so that the LSTM state is non-trivial
        eps0, hx0, cx0 = player.eps_len.clone(), player.hxs.clone(), player.cxs.clone()
        rollout(player, args.num_steps, fast=True)
        cache, buf, T = player._cache, player._buf, args.num_steps
        dones = buf[2]
        assert int(dones.sum()) > 0 and int(dones[-1].sum()) > 0
        # prologue: slot 0 of the cache = the state the rollout started from
        assert torch.equal(cache.h_all[:, 0], hx0.transpose(0, 1)) and torch.equal(cache.c_all[:, 0], cx0.transpose(0, 1))
        # epilogue
        k = (dones[-1] == 0).float().view(-1, 1, 1)
        assert torch.equal(player.hxs, cache.h_all[:, T].transpose(0, 1) * k)
        assert torch.equal(player.cxs, cache.c_all[:, T].transpose(0, 1) * k)
        nd = (dones == 0).to(torch.int32)
        alive = torch.flip(torch.cumprod(torch.flip(nd, [0]), 0), [0])
        assert torch.equal(player.eps_len, eps0 * alive[0] + alive.sum(0))
        assert torch.equal(player._keep, (dones == 0).float())
        # the epilogue with a carry to publish (what the graphed drivers give it): final LSTM state, last observation and
        # last done flags land in the carry tensors in the same launch
        from active_tracking_rl_amd import fused
        co = dict(state=torch.zeros_like(player.state), hxs=torch.zeros_like(player.hxs), cxs=torch.zeros_like(player.cxs),
                  done=torch.zeros_like(player.done))
        eps1 = eps0.clone()
        keep1 = torch.empty_like(player._keep)
        fused.rollout_end(cache.h_all, cache.c_all, dones, co["hxs"], co["cxs"], eps1, keep1, obs_src=player.state,
                          obs_dst=co["state"], done_dst=co["done"])
        assert torch.equal(co["hxs"], player.hxs) and torch.equal(co["cxs"], player.cxs) and torch.equal(keep1, player._keep)
        assert torch.equal(co["state"], player.state) and torch.equal(co["done"], dones[-1]) and torch.equal(eps1, player.eps_len)
        # bootstrap values
        model = player.model
        v = torch.zeros((n, 2, 1), device="cuda")
        model._sampler.reopen_block()
        model.boot_values(player.state, cache, player.done, v)
        model._sampler.end_block()
        a0 = cache.boot.actions[0]
        with torch.no_grad():
            x = player.state.float()
            p0, p1 = model.player0, model.player1
            f0 = p0.encoder(x[:, 0])
            f1 = p1.encoder(x.reshape(n, -1, *x.shape[3:])) + p1.fc_action_tracker(F.one_hot(a0, 4).float())
            ref = []
            for i, (p, f) in enumerate(((p0, f0), (p1, f1))):
                h, _ = p.lstm(f, (player.hxs[:, i], player.cxs[:, i]))
                ref.append(p.critic(h))
            ref = torch.stack(ref, 1)
        torch.testing.assert_close(v, ref, rtol=1e-4, atol=1e-5)
        assert int(a0.min()) >= 0 and int(a0.max()) <= 3 and len(torch.unique(a0)) == 4
        env.close()


def test_fused_adam_step_matches_the_tensor_expression():
    """atr_adam_step (SharedAdam.step as one elementwise launch + one scalar launch) against the same optimizer with the
    fused path off, over several steps: parameters and all moments to fp32 round-off, step counter / beta powers exact."""
    from active_tracking_rl_amd.shared_optim import SharedAdam
    torch.manual_seed(0)
    ws = [torch.randn(1000, 37, device="cuda"), torch.randn(4097, device="cuda")]
    pa = [torch.nn.Parameter(w.clone()) for w in ws]
    pb = [torch.nn.Parameter(w.clone()) for w in ws]
    oa, ob = SharedAdam(pa, lr=1e-3, amsgrad=True), SharedAdam(pb, lr=1e-3, amsgrad=True)
    ob.fused = False
    for step in range(6):
        for a, b in zip(pa, pb):
            g = torch.randn_like(a) * (10.0 if step == 2 else 1.0)
            a.grad.copy_(g); b.grad.copy_(g)
        oa.step(); ob.step()
        assert torch.equal(oa._scalars, ob._scalars)
        torch.testing.assert_close(oa.bucket.flat, ob.bucket.flat, rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(oa.exp_avg, ob.exp_avg, rtol=1e-6, atol=1e-9)
        torch.testing.assert_close(oa.exp_avg_sq, ob.exp_avg_sq, rtol=1e-6, atol=1e-12)
        torch.testing.assert_close(oa.max_exp_avg_sq, ob.max_exp_avg_sq, rtol=1e-6, atol=1e-12)
    assert float(oa.step_t) == 6.0


def test_fused_rmsprop_and_torch_form_adam_match_the_tensor_expressions():
    """atr_rmsprop_step and atr_adam_step(torch_eps) against the same optimizers with the fused path off, and against
    torch.optim on the GPU."""
    import argparse
    from active_tracking_rl_amd.shared_optim import make_optimizer
    for name, ref_cls in (("Adam", torch.optim.Adam), ("RMSprop", torch.optim.RMSprop)):
        torch.manual_seed(0)
        ws = [torch.randn(513, 37, device="cuda"), torch.randn(4097, device="cuda")]
        pa = [torch.nn.Parameter(w.clone()) for w in ws]
        pb = [torch.nn.Parameter(w.clone()) for w in ws]
        pc = [torch.nn.Parameter(w.clone()) for w in ws]
        ns = argparse.Namespace(optimizer=name, shared_optimizer=False, lr=1e-3, amsgrad=True)
        oa, ob, oc = make_optimizer(pa, ns), make_optimizer(pb, ns), ref_cls(pc, lr=1e-3)
        ob.fused = False
        for step in range(5):
            for a, b, c in zip(pa, pb, pc):
                g = torch.randn_like(a)
                a.grad.copy_(g); b.grad.copy_(g); c.grad = g.clone()
            oa.step(); ob.step(); oc.step()
            torch.testing.assert_close(oa.bucket.flat, ob.bucket.flat, rtol=1e-6, atol=1e-7)
            for a, c in zip(pa, pc):
                torch.testing.assert_close(a.detach(), c.detach(), rtol=2e-5, atol=1e-7)


def _check_grads(ga, gb):
    n_checked = 0
    for a, b in zip(ga, gb):
        if a is None or b is None:          # "no gradient" and "all-zero gradient" are the same statement
            other = b if a is None else a
            assert other is None or float(other.abs().max()) == 0.0
            continue
        scale = float(b.abs().max()) + 1e-7
        assert float((a - b).abs().max()) <= 2e-3 * scale + 1e-7, (a.shape, float((a - b).abs().max()), scale)
        n_checked += 1
    assert n_checked >= 10


def test_rescale_wrapper_matches_the_reference_formula():
    """--rescale (environment.Rescale, environment.py:51-56): float32 ((clip(x,0,255) - 0) * 2) / 255 + (-1)."""
    import argparse
    from active_tracking_rl_amd.environment import create_env
    args = argparse.Namespace(stack_frames=1, seed=3, gpu_ids=[0], rescale=True, single=False, inv=False, num_envs=64)
    plain = argparse.Namespace(**dict(vars(args), rescale=False))
    a, b = create_env("Track2D-BlockPartialPZR-v0", args), create_env("Track2D-BlockPartialPZR-v0", plain)
    oa, ob = a.reset().cpu().numpy(), b.reset().cpu().numpy()
    want = (((np.float32(ob).clip(0.0, 255.0) - 0.0) * (1.0 - -1.0)) / 255.0) + -1.0
    assert oa.dtype == np.float32 and np.array_equal(oa, want.astype(np.float32))
    act = [torch.zeros(64, dtype=torch.int64, device="cuda"), torch.ones(64, dtype=torch.int64, device="cuda")]
    oa, ob = a.step(act)[0].cpu().numpy(), b.step(act)[0].cpu().numpy()
    want = (((np.float32(ob).clip(0.0, 255.0) - 0.0) * 2.0) / 255.0) + -1.0
    assert np.array_equal(oa, want.astype(np.float32))
    assert a.rollout_buffers(5) is None
    a.close(); b.close()


def test_inv_flag_follows_the_reference_rescale_wrapper():
    """--rescale --inv (environment.Rescale, environment.py:49,58-76): a coin flip per EPISODE; an inverted episode's reset
    observation is -rescale(ob), its step observations 255 - rescale(ob) (the reference's own arithmetic, on the already
    rescaled image); envs that auto-reset inside a step draw a fresh flag and report -rescale(first observation)."""
    import argparse
    from active_tracking_rl_amd.environment import create_env
    n = 256
    args = argparse.Namespace(stack_frames=1, seed=3, gpu_ids=[0], rescale=True, single=False, inv=True, num_envs=n)
    plain = argparse.Namespace(**dict(vars(args), inv=False))
    torch.manual_seed(0)
    a, b = create_env("Track2D-BlockPartialPZR-v0", args), create_env("Track2D-BlockPartialPZR-v0", plain)
    oa, ob = a.reset(), b.reset()
    flags = a._inv_flags.clone()
    assert 0.25 * n < int(flags.sum()) < 0.75 * n
    f = flags.view(n, 1, 1, 1, 1, 1)
    assert torch.equal(oa, torch.where(f, -ob, ob))
    seen_reset = 0
    for t in range(40):
        act = [torch.randint(0, 4, (n,), device="cuda"), torch.randint(0, 4, (n,), device="cuda")]
        (oa, _, da, _), (ob, _, db, _) = a.step(act), b.step(act)
        assert torch.equal(da, db)
        fresh = da.bool()
        new = a._inv_flags
        assert torch.equal(new[~fresh], flags[~fresh])                      # flags only change at episode boundaries
        f, r = new.view(n, 1, 1, 1, 1, 1), fresh.view(n, 1, 1, 1, 1, 1)
        assert torch.equal(oa, torch.where(f & r, -ob, torch.where(f & ~r, 255.0 - ob, ob)))
        flags = new.clone()
        seen_reset += int(fresh.sum())
    assert seen_reset > 20
    a.close(); b.close()


def test_byte_observations_train_exactly_like_float_observations():
    """obs_u8 (t2d_step_u8 -> atr_stem_*_u8: the observation crosses HBM as bytes and is decoded in conv1) against
    the float32 path from the same seeds: same trajectories, same loss, same gradients, bit for bit, through the
    cached-rollout learner and a hipGraph-replayed iteration."""
    from active_tracking_rl_amd.train import GraphedIteration, default_args, make_player, rollout
    dev = torch.device("cuda:0")
    res = []
    for u8 in (True, False):
        args = default_args(env="Track2D-BlockPartialPZR-v0", num_envs=192, num_steps=8, seed=4, obs_u8=u8)
        player, opt = make_player(args, dev)
        assert player.env.obs_u8 == u8 and player.state.dtype == (torch.uint8 if u8 else torch.float32)
        rollout(player, args.num_steps)
        assert player._buf[0].dtype == player.state.dtype
        obs_seq = player._buf[0].float().clone()
        stats = player.optimize(None, opt, player.model, -1, dev)
        it = GraphedIteration(player, opt, args)
        for _ in range(2):
            it.run()
        torch.cuda.synchronize()
        res.append((obs_seq, [s.clone() for s in stats], opt.bucket.flat.clone(), player.state.float().clone()))
        player.env.close()
    (oa, sa, wa, xa), (ob, sb, wb, xb) = res
    assert torch.equal(oa, ob) and torch.equal(xa, xb)
    assert all(torch.equal(a, b) for a, b in zip(sa, sb))
    assert torch.equal(wa, wb)


@pytest.mark.parametrize("fixture", ["episodes.npz", "episodes_rpf.npz", "episodes_full.npz"])
def test_numpy_rng_mode_replays_the_reference_from_the_seed_alone(fixture):
    """Track2DEnv(rng="numpy", seed=s) — the product's reference-exact mode (host MT19937 + A* of track2d_np.h feeding
    t2d_inject / the target action of t2d_step) — against the reference's golden episodes given ONLY (env id, seed,
    the policy's actions): reset observations, every step's observations, float64 rewards, done (TimeLimit included)
    and info['distance'], across consecutive episodes of one stream, for Adv/PZR/Far/Ram/Nav/RPF on Block/Maze/Empty
    maps, Partial and Full observations."""
    from conftest import GOLDEN
    from active_tracking_rl_amd.environment import Track2DEnv
    g = np.load(os.path.join(GOLDEN, fixture))
    obs_kind = "Full" if "full" in fixture else "Partial"
    n_steps = 0
    for name in [str(n) for n in g["names"]]:
        mp, mode, lvl, seed, _ = [str(x) for x in g[name + "/meta"]]
        env = Track2DEnv("Track2D-%s%s%s-v%s" % (mp, obs_kind, mode, lvl), rng="numpy", seed=int(seed))
        for ep in range(int(g[name + "/n_eps"])):
            p = "%s/ep%d_" % (name, ep)
            obs0 = env.reset()
            hw = g[p + "obs0"].shape[-2:]
            assert obs0.dtype == np.float32 and np.array_equal(obs0.reshape(2, *hw), g[p + "obs0"].astype(np.float32)), p
            acts, want_obs, want_rew, want_done = g[p + "act_in"], g[p + "obs"], g[p + "rew"], g[p + "done"]
            for t in range(len(acts)):
                obs, rew, done, info = env.step([np.array(acts[t, 0]), np.array(acts[t, 1])])
                assert np.array_equal(obs.reshape(2, *hw), want_obs[t].astype(np.float32)), (p, t)
                assert np.array_equal(rew, want_rew[t].astype(np.float32).astype(np.float64)), (p, t)
                assert done == bool(want_done[t]), (p, t)
                dr = g[p + "pos"][t, 1] - g[p + "pos"][t, 0]
                assert abs(info["distance"] - float(np.sqrt(float((dr * dr).sum())))) < 1e-12
                n_steps += 1
        env.close()
    assert n_steps > 100


def test_graphed_iteration_selects_the_training_mode_per_call():
    """One rollout graph per training mode (the evaluator's train_modes schedule, test.py:84-92): replaying mode 0 (tracker
    only, --init-step phase) must leave the target's weights untouched, mode -1 must move both players, and switching
    back and forth keeps the env / LSTM carry consistent (finite weights, env invariants)."""
    from active_tracking_rl_amd.train import GraphedIteration, default_args, make_player
    dev = torch.device("cuda:0")
    args = default_args(env="Track2D-BlockPartialPZR-v0", num_envs=128, seed=9)
    player, opt = make_player(args, dev)
    it = GraphedIteration(player, opt, args)                 # captures mode -1; two eager warm-up updates happened
    snap = lambda m: torch.cat([p.detach().reshape(-1) for p in m.parameters()]).clone()
    # fresh moments so that a zero gradient means a zero Adam step for the target in mode 0
    for st in (opt.exp_avg, opt.exp_avg_sq, opt.max_exp_avg_sq):
        st.zero_()
    t0, g0 = snap(player.model.player0), snap(player.model.player1)
    it.run(0)
    torch.cuda.synchronize()
    t1, g1 = snap(player.model.player0), snap(player.model.player1)
    assert not torch.equal(t0, t1) and torch.equal(g0, g1)
    it.run(-1)
    it.run(0)
    it.run(-1)
    torch.cuda.synchronize()
    t2, g2 = snap(player.model.player0), snap(player.model.player1)
    assert not torch.equal(g1, g2) and not torch.equal(t1, t2)
    assert sorted(it.g_rolls.keys()) == [-1, 0]
    assert torch.isfinite(opt.bucket.flat).all()
    st = player.env.core.get_state()
    maps = player.env.core.get_maps()
    assert (maps[np.arange(128), st["pos"][:, 0, 0], st["pos"][:, 0, 1]] == 0).all()
    player.env.close()


@pytest.mark.parametrize("env_id,n,u8,tat", [("Track2D-BlockPartialPZR-v0", 512, True, True),
                                              ("Track2D-BlockPartialRam-v0", 257, False, False),
                                              ("Track2D-MazePartialFar-v0", 96, True, True),
                                              ("Track2D-MazePartialNav-v0", 384, True, False),
                                              ("Track2D-BlockPartialNav-v1", 130, True, True)])
def test_act_env_step_equals_cells_draws_and_env_step(env_id, n, u8, tat):
    """atr_act_env_step (k_act_step: both players' cells + heads + draws + the env step in ONE launch) against the launches
    it replaces — atr_lstm_cell_forward_act1 for the tracker, again for the target (+ emb[a_tracker] when tracker-aware),
    then t2d_step(_u8) with the two action tensors — from identical inputs on two env handles with the same seed: hidden /
    cell states, activated gates, actions, observations, rewards, done flags and env state must be bit-identical, over
    enough steps for episodes to end (in-launch auto-reset, generator passes) and with an odd batch (dead lane half)."""
    from active_tracking_rl_amd import fused, vec_env
    dev = torch.device("cuda:0")
    R, A = 128, 4
    torch.manual_seed(5)
    envs = [vec_env.VecTrack2D(env_id, num_envs=n, seed=11) for _ in range(2)]
    o0 = envs[0].reset()
    assert torch.equal(o0, envs[1].reset())
    actors = [torch.nn.Linear(R, A).to(dev) for _ in range(2)]
    for l in actors:
        l.weight.data.mul_(8.0)          # spread the action distribution a little
    bias = [torch.randn(4 * R, device=dev) * 0.1 for _ in range(2)]
    emb = torch.randn(A, 4 * R, device=dev) * 0.5 if tat else None
    samplers = [fused.ActionSampler(dev, seed=77) for _ in range(2)]
    mk = lambda *s: [torch.zeros(*s, device=dev) for _ in range(2)]
    h, c = [mk(2, n, R) for _ in range(2)], [mk(2, n, R) for _ in range(2)]     # [impl][slot] -> [2, n, R]
    for k in range(2):
        c[k][0].copy_(torch.randn(2, n, R, device=dev))
        c[1][0].copy_(c[0][0])
    acts = [torch.zeros(2, n, 4 * R, device=dev) for _ in range(2)]
    actions = [torch.zeros(2, n, dtype=torch.int64, device=dev) for _ in range(2)]
    odt = torch.uint8 if u8 else torch.float32
    outs = [(torch.zeros(n, 2, 13, 13, dtype=odt, device=dev), torch.zeros(n, 2, device=dev),
             torch.zeros(n, dtype=torch.uint8, device=dev)) for _ in range(2)]
    done_prev = [None, None]
    n_done = 0
    for t in range(70):
        ig = torch.randn(2, n, 4 * R, device=dev)
        hg = torch.randn(2, n, 4 * R, device=dev) * 0.5
        cur, nxt = t % 2, (t + 1) % 2
        for s_ in samplers:
            s_.begin_block()
        # (a) fused
        fused.act_env_step(envs[0], [ig[0], ig[1]], [hg[0], hg[1]], bias, [c[0][cur][0], c[0][cur][1]], done_prev[0],
                           [h[0][nxt][0], h[0][nxt][1]], [c[0][nxt][0], c[0][nxt][1]], [acts[0][0], acts[0][1]], samplers[0],
                           actors, actions[0], emb=emb, env_out=outs[0])
        # (b) the launches it replaces
        for p in range(2):
            fused.lstm_cell_act_into(ig[p], hg[p], c[1][cur][p], done_prev[1], h[1][nxt][p], c[1][nxt][p], acts[1][p],
                                     samplers[1], actors[p], actions[1][p], emb=emb if (tat and p == 1) else None,
                                     act_in=actions[1][0] if (tat and p == 1) else None, bias=bias[p])
        (envs[1].step_u8 if u8 else envs[1].step)(actions[1][0], actions[1][1], out=outs[1])
        for s_ in samplers:
            s_.end_block()
        assert torch.equal(actions[0], actions[1]), t
        assert torch.equal(h[0][nxt], h[1][nxt]) and torch.equal(c[0][nxt], c[1][nxt]) and torch.equal(acts[0], acts[1]), t
        for a_, b_ in zip(outs[0], outs[1]):
            assert torch.equal(a_, b_), t
        done_prev = [outs[0][2].clone(), outs[1][2].clone()]
        n_done += int(outs[0][2].sum().item())
    sa, sb = envs[0].get_state(), envs[1].get_state()
    for k_ in ("pos", "c_far", "t", "episode", "d2"):
        assert np.array_equal(sa[k_], sb[k_]), k_
    assert n_done > n // 4 and envs[0].faults() == 0
    # the policy half alone (the learner's bootstrap step): env handle None
    for s_ in samplers:
        s_.begin_block()
    ig, hg = torch.randn(2, n, 4 * R, device=dev), torch.randn(2, n, 4 * R, device=dev)
    fused.act_env_step(None, [ig[0], ig[1]], [hg[0], hg[1]], bias, [c[0][0][0], c[0][0][1]], done_prev[0],
                       [h[0][1][0], h[0][1][1]], [c[0][1][0], c[0][1][1]], None, samplers[0], actors, actions[0], emb=emb)
    for p in range(2):
        fused.lstm_cell_act_into(ig[p], hg[p], c[1][0][p], done_prev[1], h[1][1][p], c[1][1][p], None, samplers[1], actors[p],
                                 actions[1][p], emb=emb if (tat and p == 1) else None,
                                 act_in=actions[1][0] if (tat and p == 1) else None, bias=bias[p])
    assert torch.equal(actions[0], actions[1]) and torch.equal(h[0][1], h[1][1]) and torch.equal(c[0][1], c[1][1])
    for e_ in envs:
        e_.close()


@pytest.mark.gpu
def test_pipelined_learner_in_the_dw_kernels_corun_form_computes_the_same_updates():
    """From 1024 envs up PipelinedIteration captures its learner graphs with the grouped weight-gradient GEMM in its co-run
    form (fused.gemm_tn_corun: one workgroup per CU, 16-row chunks, its own K split). Same seed, same rollouts, program order
    on one stream (serial=True): the master weights after four updates with the form on and off agree to fp32 summation-order
    noise, the env shard went through the same states (the rollouts of the first two iterations do not depend on any update),
    and the form is what the default picks at this size and not below it."""
    from active_tracking_rl_amd import train
    from active_tracking_rl_amd.train import PipelinedIteration, default_args, make_player
    dev = torch.device("cuda:0")
    out = []
    for corun in ("1", "0"):
        os.environ["ATR_PIPE_CORUN"] = corun
        try:
            args = default_args(num_envs=1024, seed=5)
            player, opt = make_player(args, dev)
            it = PipelinedIteration(player, opt, args, serial=True)
            assert it.corun == (corun == "1")
            for _ in range(4):
                it.run()
            it.finish()
            torch.cuda.synchronize()
            out.append((opt.bucket.flat.clone(), player.env.core.get_state()["pos"].copy()))
            player.env.close()
        finally:
            os.environ.pop("ATR_PIPE_CORUN", None)
    (w1, pos1), (w0, pos0) = out
    assert torch.isfinite(w1).all()
    assert float((w1 - w0).abs().max()) < 2e-4, float((w1 - w0).abs().max())
    assert np.mean(np.all(pos1 == pos0, axis=tuple(range(1, pos1.ndim)))) > 0.9      # (sampling diverges only through the weights)
    for n, want in ((1024, True), (512, False)):
        args = default_args(num_envs=n, seed=5)
        player, opt = make_player(args, dev)
        assert PipelinedIteration(player, opt, args, serial=True).corun == want and train.CORUN_MIN_ENVS == 1024
        player.env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("env_id,network,mode", [("Track2D-BlockPartialPZR-v0", "tat-maze-lstm", -1),
                                                 ("Track2D-BlockPartialNav-v0", "maze-lstm", 0)])
def test_pipelined_schedule_is_the_same_dataflow_on_one_stream_and_on_two(env_id, network, mode):
    """PipelinedIteration (rollout i + 1 on one HIP stream under learner + update i on another; gradients one update late):
    the delay lives in the double-buffered replica weights, not in the timing — replaying the same graphs in program order
    on one stream (serial=True) must leave bit-identical master weights, replica weights, env state and carried LSTM state.
    And the schedule itself: after i updates the replica that just finished holds theta_i, the other one theta_{i-1}."""
    from active_tracking_rl_amd.train import PipelinedIteration, default_args, make_player
    dev = torch.device("cuda:0")
    res = []
    for serial in (True, False):
        args = default_args(env=env_id, network=network, aux="reward" if "tat" in network else "none", train_mode=mode,
                            num_envs=256, seed=11)
        player, opt = make_player(args, dev)
        it = PipelinedIteration(player, opt, args, serial=serial)
        thetas = [opt.bucket.flat.clone()]
        for _ in range(7):
            it.run()
            it.finish()
            torch.cuda.synchronize()
            thetas.append(opt.bucket.flat.clone())
        if not serial:                      # stream trials are ordinary iterations: run them here, mirror them below
            trials = it.tune_streams(candidates=2, iters=4, partitions=(128,), keep_updates=True)
            assert len(trials) == 4 and sum(c for _, c, _ in trials) == 1 and "partition" in trials[-1][2]
        else:
            for _ in range(4 * 2 * (2 + 4)):                # (4 candidates, two passes, 2 warm-up + 4 timed phases each)
                it.run()
        it.finish()
        torch.cuda.synchronize()
        assert it.i == 7 + 48
        k = (it.i - 1) & 1
        assert torch.equal(it.buckets[k].flat, opt.bucket.flat)                    # O(i): theta -> F_k
        assert not torch.equal(it.buckets[1 - k].flat, opt.bucket.flat)            # the other replica is one update behind
        assert all(not torch.equal(a, b) for a, b in zip(thetas[:-1], thetas[1:]))  # every iteration updates theta
        assert torch.isfinite(opt.bucket.flat).all()
        st = player.env.core.get_state()
        res.append((thetas + [opt.bucket.flat.clone()], [b.flat.clone() for b in it.buckets],
                    {k_: v.clone() for k_, v in it.carry.items()}, st["pos"].copy(), [s.clone() for s in it.stats]))
        if not serial:
            # by default the stream trials are NOT training: master weights, optimizer state, replica weights, phase and step
            # counters come back as they were (main.py tunes before iteration 0: the first logged iteration is the first update)
            snap = [t.clone() for t in it._schedule_tensors()]
            i_before, n_before = it.i, it.master.n_steps
            trials = it.tune_streams(candidates=1, iters=3)
            assert sum(c for _, c, _ in trials) == 1
            assert all(torch.equal(a, b) for a, b in zip(snap, it._schedule_tensors()))
            assert (it.i, it.master.n_steps) == (i_before, n_before)
            it.run(); it.run(); it.finish()
            torch.cuda.synchronize()
            assert torch.isfinite(opt.bucket.flat).all() and not torch.equal(snap[0], opt.bucket.flat)
        player.env.close()
    (ta, fa, ca, pa, sa), (tb, fb, cb, pb, sb) = res
    assert all(torch.equal(a, b) for a, b in zip(ta, tb))
    assert all(torch.equal(a, b) for a, b in zip(fa, fb))
    assert all(torch.equal(ca[k_], cb[k_]) for k_ in ca)
    assert np.array_equal(pa, pb)
    assert all(torch.equal(a, b) for a, b in zip(sa, sb))


def test_numpy_rng_batch_replays_all_reference_episodes_concurrently_in_one_handle():
    """environment.NumpyVecEnv — the reference-exact mode for a BATCH: all 16 multi-episode cases of episodes.npz (Block / Maze /
    Empty maps x PZR / Adv / Far / Ram / Nav targets x levels 0 / 1, each captured from the reference env on its own seed), four
    copies of each = 64 envs in ONE device handle, advanced in lock step from (env id, seed, the recorded policy actions) alone:
    every env's maps / spawns / scripted-target actions come from its own numpy-legacy stream on host threads (the reference's
    own A* paths for Nav), the device steps all 64 per launch. Every step's observations, float64 rewards, done flags and
    info['distance'] must equal the reference's, and so must the first observation of every following episode (the stream
    goes on across episodes exactly as in the reference's worker loop, train.py:73-74)."""
    import time
    from conftest import GOLDEN
    from active_tracking_rl_amd.environment import NumpyVecEnv
    g = np.load(os.path.join(GOLDEN, "episodes.npz"))
    names = [str(n) for n in g["names"]] * 4
    ids, seeds, eps = [], [], []
    for name in names:
        mp, mode, lvl, seed, _ = [str(x) for x in g[name + "/meta"]]
        ids.append("Track2D-%sPartial%s-v%s" % (mp, mode, lvl))
        seeds.append(int(seed))
        eps.append([{k: g["%s/ep%d_%s" % (name, e, k)] for k in ("obs0", "act_in", "obs", "rew", "done", "pos")}
                    for e in range(int(g[name + "/n_eps"]))])
    n = len(names)
    assert n == 64
    env = NumpyVecEnv(ids, seeds)
    obs = env.reset().cpu().numpy()
    for i in range(n):
        assert np.array_equal(obs[i], eps[i][0]["obs0"].astype(np.float32)), (names[i], "first reset")
    cur = [[0, 0] for _ in range(n)]              # (episode, step) of every env; episode == len(eps[i]): finished
    steps = 0
    t0 = time.time()
    while any(c[0] < len(eps[i]) for i, c in enumerate(cur)):
        act = np.zeros((2, n), np.int64)
        for i, (e, t) in enumerate(cur):
            if e < len(eps[i]):
                act[:, i] = eps[i][e]["act_in"][t]
        obs, rew, done, info = env.step([act[0], act[1]])
        obs, rew, done = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
        cut = np.zeros(n, bool)
        for i, (e, t) in enumerate(cur):
            if e >= len(eps[i]):
                continue
            E = eps[i][e]
            last = t == len(E["act_in"]) - 1
            assert bool(done[i]) == bool(E["done"][t]), (names[i], e, t, "done")
            assert np.array_equal(rew[i].astype(np.float64), E["rew"][t].astype(np.float32).astype(np.float64)), (names[i], e, t)
            dr = E["pos"][t, 1] - E["pos"][t, 0]
            assert abs(info["distance"][i] - float(np.sqrt(float((dr * dr).sum())))) < 1e-12
            if not done[i]:
                assert np.array_equal(obs[i], E["obs"][t].astype(np.float32)), (names[i], e, t, "obs")
            elif e + 1 < len(eps[i]):             # finished: the env restarted from its stream inside step()
                assert last
                assert np.array_equal(obs[i], eps[i][e + 1]["obs0"].astype(np.float32)), (names[i], e + 1, "obs0 after done")
            steps += 1
            if last:
                cur[i] = [e + 1, 0]
                cut[i] = not done[i] and e + 1 < len(eps[i])       # the capture stopped this episode early: reset() as it did
            else:
                cur[i][1] = t + 1
        if cut.any():
            fresh = env.reset(cut).cpu().numpy()
            for i in np.nonzero(cut)[0]:
                assert np.array_equal(fresh[i], eps[i][cur[i][0]]["obs0"].astype(np.float32)), (names[i], cur[i][0], "obs0")
    dt = time.time() - t0
    env.close()
    assert steps > 5000
    print("numpy-rng batch: %d env steps of 64 concurrent reference replays in %.2f s (%.0f env steps/s incl. the checks)"
          % (steps, dt, steps / dt))


def test_device_astar_matches_the_reference_solver():
    """The HIP restatement of AstarSolver (csrc/track2d_hip.hip astar_np: heapq's sift order, [f, node] list comparison, the
    inverted replace test) on the 49 searches recorded from the reference (tests/golden/astar.npz): solvable or not, and the
    action list itself — tie-breaks included, not just its length."""
    from conftest import GOLDEN, unpack_maze
    from active_tracking_rl_amd import np_mode
    d = np.load(os.path.join(GOLDEN, "astar.npz"))
    names = sorted(set(k.split("/")[0] for k in d.keys() if "/" in k))
    assert len(names) >= 40
    n_unsolvable = 0
    for nme in names:
        ok, acts = np_mode.astar_device(unpack_maze(d[nme + "/maze"], d[nme + "/side"]), d[nme + "/start"], d[nme + "/goal"])
        assert ok == bool(d[nme + "/solvable"]), nme
        n_unsolvable += not ok
        if ok:
            assert np.array_equal(acts, d[nme + "/actions"]), nme
    assert n_unsolvable >= 1


@pytest.mark.parametrize("targets", ["unscripted", "ram"])
def test_device_side_numpy_streams_replay_the_reference_episodes_from_the_seed_alone(targets):
    """NumpyVecEnv(device_generators=True): the numpy-legacy streams live on the DEVICE (t2d_np_attach -> k_gen_np: MT19937, numpy's
    doubles / bounded integers / whole Fisher-Yates permutations, init_maze, both map generators, sample_goal,
    sample_close_states, get_around restated draw for draw, one wavefront per env) and a finished env restarts inside the step
    launch from an episode pre-generated out of its own stream. Every multi-episode capture of episodes.npz whose target draws
    nothing itself (PZR / Adv / Far on Block / Maze / Empty maps, levels 0 / 1), four copies each, advanced in lock step from
    (env id, seed, recorded policy actions) ALONE — no host stream, no injection: every observation, reward and done flag of
    every episode equals the reference's, the first observation of every following episode included.
    targets="ram" (round 6): the captures with the scripted Ram target (RamAgent, navigator.py:73-93: its coin / action / run
    length draws come from the same stream BETWEEN the resets) next to two unscripted ones in ONE handle. Nothing can be
    generated ahead of time there: the handle runs without the in-launch auto-reset, a finished env's next episode is drawn
    inside the masked reset that follows its terminal step (init_maze, then RamAgent.reset), and RamAgent.step runs on the
    device ahead of every step launch (k_ram_np) — the recorded TARGET actions are never fed in, the device's Ram target must
    reproduce them from the seed for the observations to match. The same handle holds the captures with the Nav target: the
    reference's Navigator — goal draws (whole permutations of the free cells), the retry loop, plan B and AstarSolver with heapq's
    exact sift order, list-comparison tie-breaks and its inverted replace test — runs on the device too (one lane per search),
    and the device's target must walk the reference's own A* paths, not merely paths of the same length."""
    from conftest import GOLDEN
    from active_tracking_rl_amd.environment import NumpyVecEnv
    g0 = np.load(os.path.join(GOLDEN, "episodes.npz"))
    if targets == "ram":
        grpf = np.load(os.path.join(GOLDEN, "episodes_rpf.npz"))
        names = [(g0, str(n)) for n in g0["names"] if str(g0[str(n) + "/meta"][1]) in ("Ram", "Nav")] * 4
        names += [(grpf, str(n)) for n in grpf["names"]] * 2          # the RPF patrol (plans on the generator's map, walks the env's)
        names += [(g0, str(n)) for n in g0["names"] if str(g0[str(n) + "/meta"][1]) in ("PZR", "Adv")][:2]
        modes = [str(gg[n + "/meta"][1]) for gg, n in names]
        assert modes.count("Ram") >= 8 and modes.count("Nav") >= 8 and modes.count("RPF") >= 8
    else:
        names = [(g0, str(n)) for n in g0["names"] if str(g0[str(n) + "/meta"][1]) in ("PZR", "Adv", "Far")] * 4
    ids, seeds, eps = [], [], []
    for g, name in names:
        mp, mode, lvl, seed, _ = [str(x) for x in g[name + "/meta"]]
        ids.append("Track2D-%sPartial%s-v%s" % (mp, mode, lvl))
        seeds.append(int(seed))
        eps.append([{k: g["%s/ep%d_%s" % (name, e, k)] for k in ("obs0", "act_in", "obs", "rew", "done", "pos")}
                    for e in range(int(g[name + "/n_eps"]))])
        for E in eps[-1]:      # (the RPF captures step on THROUGH done — the raw env never resets itself; a batch env restarts a finished
            dn = np.nonzero(E["done"])[0]          # env at once, so an episode is the capture up to its first done. The patrol draws
            if len(dn) and dn[0] + 1 < len(E["done"]):     # nothing at step time: the next capture's reset follows in the stream.)
                for k in ("act_in", "obs", "rew", "done", "pos"):
                    E[k] = E[k][:dn[0] + 1]
    n = len(names)
    if targets == "unscripted":
        assert n >= 24 and len(set(i.split("Partial")[0] for i in ids)) == 3          # Block, Maze and Empty maps among them
    names = [n for _, n in names]
    env = NumpyVecEnv(ids, seeds, device_generators=True)
    assert env._interleaved == (targets == "ram") and bool(env.core.auto_reset) == (targets != "ram")
    is_ram = np.array([any(m in i for m in ("Ram", "Nav", "RPF")) for i in ids])    # (scripted on the device: the recorded target action is not fed)
    obs = env.reset().cpu().numpy()
    for i in range(n):
        assert np.array_equal(obs[i], eps[i][0]["obs0"].astype(np.float32)), (names[i], "first reset")
    cur = [[0, 0] for _ in range(n)]
    steps = 0
    while any(c[0] < len(eps[i]) for i, c in enumerate(cur)):
        act = np.zeros((2, n), np.int64)
        for i, (e, t) in enumerate(cur):
            if e < len(eps[i]):
                act[:, i] = eps[i][e]["act_in"][t]
        act[1, is_ram] = 3 - act[1, is_ram]          # (a Ram env's recorded target action must NOT be what moves its target)
        obs, rew, done, info = env.step([act[0], act[1]])
        obs, rew, done = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
        cut = np.zeros(n, bool)
        for i, (e, t) in enumerate(cur):
            if e >= len(eps[i]):
                continue
            E = eps[i][e]
            last = t == len(E["act_in"]) - 1
            assert bool(done[i]) == bool(E["done"][t]), (names[i], e, t, "done")
            assert np.array_equal(rew[i].astype(np.float64), E["rew"][t].astype(np.float32).astype(np.float64)), (names[i], e, t)
            dr = E["pos"][t, 1] - E["pos"][t, 0]          # info['distance'] is the STEP's, also when the env restarted in the launch
            assert abs(info["distance"][i] - float(np.sqrt(float((dr * dr).sum())))) < 1e-12, (names[i], e, t, "distance")
            if not done[i]:
                assert np.array_equal(obs[i], E["obs"][t].astype(np.float32)), (names[i], e, t, "obs")
            elif e + 1 < len(eps[i]):             # finished: restarted inside the launch, from the env's own stream
                assert last
                assert np.array_equal(obs[i], eps[i][e + 1]["obs0"].astype(np.float32)), (names[i], e + 1, "obs0 after done")
            steps += 1
            if last:
                cur[i] = [e + 1, 0]
                cut[i] = not done[i] and e + 1 < len(eps[i])       # the capture stopped this episode early: reset() as it did
            else:
                cur[i][1] = t + 1
        if cut.any():
            fresh = env.reset(cut).cpu().numpy()
            for i in np.nonzero(cut)[0]:
                assert np.array_equal(fresh[i], eps[i][cur[i][0]]["obs0"].astype(np.float32)), (names[i], cur[i][0], "obs0")
    assert env.core.faults() == 0
    env.close()
    assert steps > (2000 if targets == "unscripted" else 1000)


def test_create_env_numpy_rng_batch_equals_single_numpy_envs():
    """create_env(..., num_envs=4, rng="numpy") is the batch form of the reference-exact mode: env i runs np.random.seed(seed + i),
    i.e. what four single create_env(..., rng="numpy") envs seeded seed + i do (the reference worker's env.seed(seed + rank),
    train.py:23), observation for observation and reward for reward."""
    import argparse
    from active_tracking_rl_amd.environment import NumpyVecEnv, create_env
    mk = lambda seed: argparse.Namespace(stack_frames=1, seed=seed, rescale=False, gpu_ids=[0])
    batch = create_env("Track2D-BlockPartialNav-v0", mk(31), num_envs=4, rng="numpy")
    assert isinstance(batch, NumpyVecEnv)
    singles = [create_env("Track2D-BlockPartialNav-v0", mk(31 + i), num_envs=1, rng="numpy") for i in range(4)]
    ob = batch.reset().cpu().numpy()
    so = [e.reset() for e in singles]
    so = [x.cpu().numpy() if torch.is_tensor(x) else np.asarray(x) for x in so]
    for i in range(4):
        assert np.array_equal(ob[i], so[i].reshape(ob[i].shape).astype(np.float32)), i
    rs = np.random.RandomState(3)
    for t in range(120):
        a = rs.randint(0, 7, size=(2, 4))
        ob, rew, done, _ = batch.step([a[0], a[1]])
        ob, rew, done = ob.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
        for i, e in enumerate(singles):
            o1, r1, d1, _ = e.step([int(a[0, i]), int(a[1, i])])
            o1 = o1.cpu().numpy() if torch.is_tensor(o1) else np.asarray(o1)
            r1 = r1.cpu().numpy() if torch.is_tensor(r1) else np.asarray(r1)
            assert bool(d1) == bool(done[i]), (t, i)
            assert np.array_equal(np.asarray(r1, np.float32).reshape(-1), rew[i].reshape(-1)), (t, i)
            if d1:
                o1 = e.reset()
                o1 = o1.cpu().numpy() if torch.is_tensor(o1) else np.asarray(o1)
            assert np.array_equal(ob[i], o1.reshape(ob[i].shape).astype(np.float32)), (t, i)
    batch.close()
    for e in singles:
        e.close()


def test_create_env_numpy_device_streams_equal_the_host_streams():
    """create_env(..., num_envs=n, rng="numpy-device"): the numpy-legacy streams on the device (t2d_np_attach, k_gen_np) against the
    host-stream batch (rng="numpy": csrc/np_mode.cpp + t2d_inject) on the same seeds — two independent restatements of the
    reference's draws, one in HIP and one in C++ — over many short episodes with in-launch restarts on the device side: every
    observation, reward and done flag equal."""
    import argparse
    from active_tracking_rl_amd.environment import NumpyVecEnv, create_env
    mk = lambda seed: argparse.Namespace(stack_frames=1, seed=seed, rescale=False, gpu_ids=[0])
    for env_id in ("Track2D-BlockPartialPZR-v0", "Track2D-MazePartialAdv-v0", "Track2D-EmptyPartialFar-v1"):
        dev_env = create_env(env_id, mk(77), num_envs=24, rng="numpy-device")
        host_env = create_env(env_id, mk(77), num_envs=24, rng="numpy")
        assert isinstance(dev_env, NumpyVecEnv) and dev_env.device_generators and not host_env.device_generators
        assert torch.equal(dev_env.reset(), host_env.reset())
        rs = np.random.RandomState(9)
        ends = 0
        for t in range(150):
            a = rs.randint(0, 4, size=(2, 24))
            od, rd, dd, _ = dev_env.step([a[0], a[1]])
            oh, rh, dh, _ = host_env.step([a[0], a[1]])
            assert torch.equal(dd, dh) and torch.equal(rd, rh) and torch.equal(od, oh), (env_id, t)
            ends += int(dd.sum())
        assert ends > 24
        dev_env.close(); host_env.close()
    # the scripted Ram and Nav targets (round 6): their streams on the device too — RamAgent.step / Navigator.step (heap A*) as k_ram_np,
    # episodes drawn inside the masked reset — against the host streams (np_mode.cpp's RamAgent / Navigator / Astar), over episode ends
    # and many re-plans; info['distance'] included
    for env_id in ("Track2D-BlockPartialRam-v0", "Track2D-MazePartialRam-v1", "Track2D-MazePartialNav-v0", "Track2D-BlockPartialNav-v1",
                   "Track2D-BlockPartialRPF-v0", "Track2D-MazePartialRPF-v1"):
        dev_env = create_env(env_id, mk(5), num_envs=24, rng="numpy-device")
        host_env = create_env(env_id, mk(5), num_envs=24, rng="numpy")
        assert dev_env.device_generators and dev_env._interleaved and not host_env.device_generators
        assert torch.equal(dev_env.reset(), host_env.reset())
        rs = np.random.RandomState(3)
        ends = 0
        for t in range(200):
            a = rs.randint(0, 4, size=(2, 24))
            od, rd, dd, idv = dev_env.step([a[0], a[1]])
            oh, rh, dh, ih = host_env.step([a[0], a[1]])
            assert torch.equal(dd, dh) and torch.equal(rd, rh) and torch.equal(od, oh), (env_id, t)
            assert np.array_equal(idv["distance"], ih["distance"]), (env_id, t)
            ends += int(dd.sum())
        assert ends > 24
        if "Ram" in env_id:
            tg = dev_env.core.get_target()
            assert (tg["len"] >= 1).all() and (tg["len"] <= 9).all() and (tg["cursor"] < tg["len"]).all()
        assert dev_env.core.faults() == 0
        dev_env.close(); host_env.close()


def test_max_grad_norm_is_applied_inside_the_captured_update_graphs():
    """--max-grad-norm under the graphed drivers: the clip (clip_grad_norm_ on the flat bucket, player_util.py:157's intent) is
    captured in the update graph of GraphedIteration and PipelinedIteration — the gradient the last update consumed has at most
    that norm under every driver (eager loop, synchronous graphs, pipelined graphs), the unclipped run's is larger and its
    weights differ, and the clipped eager and graphed runs stay close (same update rule on near-identical rollouts)."""
    from active_tracking_rl_amd.train import GraphedIteration, PipelinedIteration, default_args, make_player, rollout
    dev = torch.device("cuda:0")

    def run(kind, max_norm):
        args = default_args(env="Track2D-BlockPartialPZR-v0", num_envs=256, seed=13, max_grad_norm=max_norm)
        player, opt = make_player(args, dev)
        if kind == "eager":
            for _ in range(2):
                rollout(player, args.num_steps)
                player.optimize(None, opt, player.model, args.train_mode, dev)
        elif kind == "graph":
            it = GraphedIteration(player, opt, args, warmup=0)
            for _ in range(2):
                it.run()
        else:
            it = PipelinedIteration(player, opt, args, warmup=0, serial=True)
            for _ in range(2):
                it.run()
            it.finish()
        torch.cuda.synchronize()
        w = opt.bucket.flat.clone()
        gn = float(opt.bucket.grad.norm())
        player.env.close()
        return w, gn
    w_eager, gn = run("eager", 0.05)
    assert gn <= 0.05 * 1.0001, gn                      # the gradient the last update saw was clipped
    w_graph, gn_g = run("graph", 0.05)
    assert gn_g <= 0.05 * 1.0001
    assert (w_eager - w_graph).abs().max() < 5e-3
    w_free, gn_free = run("graph", None)
    assert gn_free > 0.05 and not torch.equal(w_free, w_graph)
    w_pipe, gn_p = run("pipelined", 0.05)               # (one update of delay: another trajectory, but clipped all the same)
    assert gn_p <= 0.05 * 1.0001 and torch.isfinite(w_pipe).all()
