"""train — the worker loop of train.py:15-113, one process per GPU instead of 16 Hogwild CPU workers.

    train(rank, args, shared_model, optimizer, train_modes, n_iters, env=None)

keeps the reference signature. Semantics per iteration are the reference's (sync weights -> <= num_steps env
steps -> optimize -> log), with these MI355X-first substitutions:
  * each rank owns `args.num_envs` envs (a shard of the global batch; Philox streams are keyed by the global env
    id so the union of shards equals the unsharded batch) stepped by the HIP kernels;
  * `shared_model` is this rank's replica (weights stay bit-identical across ranks because every rank applies
    the same all-reduced gradient), so `load_state_dict(shared_model.state_dict())` (train.py:71) is a no-op
    and is skipped;
  * the rollout is always `num_steps` long — finished envs auto-reset in the step launch (see player_util).
`train_modes` / `n_iters` may be plain lists (single process) or mp.Manager lists as in main.py:103-105.
"""
import os
import time

import torch

from .environment import create_env
from .model import build_model
from .player_util import Agent
from .shared_optim import SharedAdam, make_optimizer  # noqa: F401  (SharedAdam re-exported for callers)


def default_args(**over):
    """The flag defaults of main.py:16-50 as a namespace (plus num_envs / max_grad_norm / obs_u8: keep the training
    env's observations as bytes up to the policy's conv stem where the kernels allow it, see environment.VecEnv)."""
    import argparse
    d = dict(lr=0.001, gamma=0.9, tau=1.00, entropy=0.01, entropy_target=0.2, seed=1, workers=1, num_steps=20,
             test_eps=100, env='Track2D-BlockPartialPZR-v0', env_base='Track2D-BlockPartialNav-v0', optimizer='Adam',
             amsgrad=True, load_model_dir=None, log_dir='logs/', network='tat-maze-lstm', aux='reward', gpu_ids=[0],
             obs='img', single=False, gray=False, crop=False, inv=False, rescale=False, render=False,
             shared_optimizer=True, split=False, train_mode=-1, stack_frames=1, input_size=80, rnn_out=128,
             sleep_time=0, max_step=150000, init_step=-1, num_envs=4096, max_grad_norm=None, obs_u8=True)
    d.update(over)
    return argparse.Namespace(**d)


def select_params(model, train_mode):
    """train.py:39-44: which half of the two-player model the optimizer owns."""
    if train_mode == 0:
        return list(model.player0.parameters())
    if train_mode == 1:
        return list(model.player1.parameters())
    return list(model.parameters())


def make_player(args, device, rank=0, world_size=1, env=None, model=None, optimizer=None):
    """Build env shard + replica + optimizer + Agent for one rank."""
    if device.type == 'cuda':
        from . import gemm_tuning
        gemm_tuning.enable()                  # read-only TunableOp picks for the policy GEMMs (no-op without the file)
    torch.manual_seed(args.seed)          # same init on every rank (replicas must start identical)
    if env is None:
        env = create_env(args.env, args, num_envs=args.num_envs, device=str(device),
                         env_id_base=rank * args.num_envs)
    if model is None:
        model = build_model(env.observation_space, env.action_space, args, device).to(device)
    model.train()
    if optimizer is None:
        optimizer = make_optimizer(select_params(model, args.train_mode), args)
    torch.manual_seed(args.seed + rank)   # per-rank action sampling (train.py:20)
    if device.type == 'cuda':
        torch.cuda.manual_seed(args.seed + rank)
    player = Agent(model, env, args, None, device)
    player.w_entropy_target = args.entropy_target
    player.reset()
    return player, optimizer


def rollout(player, num_steps, fast=True):
    """train.py:79-88 without the early break (done is a per-env mask). fast=True: actor/learner split — the policy
    steps without autograd (action_rollout) and Agent.loss_recompute re-evaluates the stored rollout time-batched;
    fast=False: the reference-shaped per-step autograd path (action_train + loss)."""
    if hasattr(player.model, "cache_dense"):
        player.model.cache_dense(True)   # expand conv weights once per rollout (released in compute_grads)
    if fast:
        player.begin_rollout(num_steps)
        for _ in range(num_steps):
            player.action_rollout()
        player.end_rollout()
    else:
        player.update_rnn_hiden()
        for _ in range(num_steps):
            player.action_train()
    # A rollout that is not a whole number of generator stamp cycles restarts the stamps (so that a captured rollout
    # replays consistently); otherwise forked generator launches stay in flight under the learner's kernels.
    if hasattr(player.env, "flush") and num_steps % getattr(player.env, "generator_cycle", 1) != 0:
        player.env.flush()


class GraphedIteration(object):
    """One A3C iteration (20-step rollout -> loss -> backward | all-reduce | SharedAdam) replayed as two hipGraphs.

    The rollout is launch-bound in eager mode (~60 small kernels per env step); capturing it removes the host
    from the loop. The gradient all-reduce stays an eager RCCL call between the two graphs, so the captured
    regions contain no collective. State carried between iterations (obs, LSTM state, done, episode lengths) lives
    in static tensors that the captured region reads first and writes last; the env state itself is device-resident
    inside the HIP library.

    The training mode (which player's loss is differentiated, player_util.py:147-152) is a constant of the captured
    loss, so there is ONE rollout graph PER MODE, captured the first time `run(mode)` sees it: the evaluator's
    `train_modes[rank]` schedule (test.py:84-92: tracker-only until --init-step) selects the graph per iteration.
    All graphs read and write the same carry tensors, so switching modes does not disturb the env/LSTM state."""

    def __init__(self, player, optimizer, args, warmup=2, fast=True, mode=None, keep_warmup_updates=False):
        """mode: the training mode of the first iterations (main.py starts in mode 0 under --init-step); the eager
        warm-up iterations and the first captured graph use it. The warm-up iterations are real rollouts + updates run to
        settle allocations before the capture; unless keep_warmup_updates is set their effect on the parameters and the
        optimizer state (step counter, moments) is rolled back, so that iteration 0 of the run is the first replay."""
        self.player, self.optimizer, self.args, self.fast = player, optimizer, args, fast
        self.mode0 = args.train_mode if mode is None else int(mode)
        dev = player.device
        snap = None if keep_warmup_updates else self._optimizer_tensors()
        saved = [t.clone() for t in snap] if snap is not None else None
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._eager()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        if saved is not None:
            with torch.no_grad():
                for t, v in zip(snap, saved):
                    t.copy_(v)
        self.carry = dict(state=player.state.clone(), hxs=player.hxs.detach().clone(),
                          cxs=player.cxs.detach().clone(), done=player.done.clone(), eps_len=player.eps_len.clone())
        self.g_rolls, self.stats_by_mode = {}, {}
        self._capture(self.mode0)
        self.g_opt = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_opt, capture_error_mode="thread_local"):
            optimizer.step()
        if saved is not None:          # (capturing a graph does not execute it; restore anyway in case a backend ran it)
            with torch.no_grad():
                for t, v in zip(snap, saved):
                    t.copy_(v)

    def _optimizer_tensors(self):
        """Every tensor an update writes: the flat parameter bucket and the optimizer's own state tensors."""
        opt = self.optimizer
        bucket = getattr(opt, "bucket", None)
        if bucket is None:
            return None
        seen, out = set(), []
        for t in [bucket.flat] + [v for v in vars(opt).values() if isinstance(v, torch.Tensor)]:
            if t.data_ptr() not in seen and t.numel() > 0:
                seen.add(t.data_ptr())
                out.append(t)
        return out

    def _bind_carry(self):
        p = self.player
        p.state, p.hxs, p.cxs = self.carry["state"], self.carry["hxs"], self.carry["cxs"]
        p.done, p.eps_len = self.carry["done"], self.carry["eps_len"]

    def _capture(self, mode):
        player, args = self.player, self.args
        if hasattr(player.env, "flush"):
            player.env.flush()           # no generator launch in flight and stamp 0 when the capture starts
        torch.cuda.synchronize(player.device)
        g = torch.cuda.CUDAGraph()
        # thread_local: an RCCL watchdog thread polling events must not invalidate the capture
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            self._bind_carry()
            rollout(player, args.num_steps, fast=self.fast)
            stats = player.compute_grads(self.optimizer, mode)
            if hasattr(player.env, "generator_join"):
                player.env.generator_join()   # the env's forked generator launches end inside the captured region
            for k, src in (("state", player.state), ("hxs", player.hxs.detach()), ("cxs", player.cxs.detach()),
                           ("done", player.done), ("eps_len", player.eps_len)):
                if self.carry[k].data_ptr() != src.data_ptr():    # (the epilogue kernel advances eps_len in place)
                    self.carry[k].copy_(src)
        self._bind_carry()
        self.g_rolls[mode], self.stats_by_mode[mode] = g, stats
        self.g_roll, self.stats = g, stats          # the most recently captured pair (kept for callers/tools)
        return g

    def _eager(self):
        rollout(self.player, self.args.num_steps, fast=self.fast)
        self.player.optimize(None, self.optimizer, self.player.model, self.mode0, self.player.device)

    def run(self, mode=None):
        mode = self.mode0 if mode is None else int(mode)
        g = self.g_rolls.get(mode)
        if g is None:
            g = self._capture(mode)
        g.replay()
        self.player.allreduce_grads(self.optimizer)
        self.g_opt.replay()
        self.player.n_steps += self.args.num_steps
        self.stats = self.stats_by_mode[mode]
        return self.stats


def sync_train_modes(train_modes, device, src=0):
    """Every rank must differentiate the same loss before the all-reduce: rank `src` (the only one running the
    evaluator, which owns the schedule — test.py:84-92,129-134) broadcasts its train_modes list."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return train_modes
    t = torch.tensor([int(m) for m in train_modes], dtype=torch.int64, device=device)
    dist.broadcast(t, src=src)
    train_modes[:] = [int(v) for v in t.tolist()]
    return train_modes


def train(rank, args, shared_model, optimizer, train_modes, n_iters, env=None):
    gpu_id = args.gpu_ids[rank % len(args.gpu_ids)]
    device = torch.device('cuda:%d' % gpu_id)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    training_mode = args.train_mode
    while len(train_modes) <= rank:
        train_modes.append(training_mode)
        n_iters.append(0)
    player, optimizer = make_player(args, device, rank, world, env=env, model=shared_model, optimizer=optimizer)
    from .utils import ScalarWriter, log_train_scalars
    writer = ScalarWriter(os.path.join(args.log_dir, 'Agent:{}'.format(rank)))
    n_iter = 0
    try:
        while True:
            t0 = time.time()
            rollout(player, args.num_steps)
            training_mode = train_modes[rank]
            policy_loss, value_loss, entropies, pred_loss = player.optimize(
                None, optimizer, shared_model, training_mode, device)
            n_iter += 1
            n_iters[rank] = n_iter
            if n_iter % 10 == 0:
                torch.cuda.synchronize(device)
                fps = args.num_steps * player.num_envs / (time.time() - t0)
                log_train_scalars(writer, (policy_loss, value_loss, entropies, pred_loss), training_mode, fps, player.n_steps,
                                  player.num_agents)
                writer.flush()
            if train_modes[rank] == -100 or n_iter * world > args.max_step:   # test.py:129-134 stop rule
                break
        player.env.close()
    except KeyboardInterrupt:
        player.env.close()
    return player
