"""test — the evaluator of test.py:16-136, vectorised: `test_eps` evaluation episodes run as ONE batch of envs on the
GPU (argmax policy, Agent.action_test) instead of one env in a side process.

Kept from the reference: evaluation on `--env-base` (default Track2D-BlockPartialNav-v0, main.py:27), the summary line
("ave eps reward / ave eps length / reward step", test.py:101-109), the checkpoint names and contents
(`all-best-{n_iter}.dat` / `all-new.dat`, and with --split `tracker-{best,new}.dat`, `target-{best,new}.dat`:
state_dicts with the reference's keys, test.py:111-127), the train_modes schedule (test.py:84-92, including the
--train-mode 2 tracker / target alternation, which needs the --adv-step flag the reference forgot to define), the
`test/reward{i}`, `test/fps`, `test/eps_len` scalars (test.py:93-97; one record per evaluation episode, as there) and the
stop sentinel -100 once n_iter > max_step (test.py:129-134). Success = episode length >= 500 (gym_eval.py:114-115).
"""
import logging
import os
import time

import numpy as np
import torch

from .environment import create_env
from .model import build_model
from .player_util import Agent
from .utils import ScalarWriter, check_path, setup_logger


@torch.no_grad()
def evaluate(model, env_id, args, device, episodes, seed=None):
    """Run `episodes` envs of `env_id` in parallel until each has finished ONE episode. Returns per-episode reward
    sums [episodes, 2] and lengths [episodes] (numpy)."""
    ev = create_env(env_id, args, num_envs=max(2, episodes), device=str(device),
                    env_id_base=getattr(args, "eval_env_id_base", 1 << 20))
    n = ev.num_envs
    was_training = model.training
    model.eval()
    player = Agent(model, ev, args, None, device)
    player.reset()
    rsum = torch.zeros(n, player.num_agents, device=device)
    length = torch.zeros(n, dtype=torch.int32, device=device)
    alive = torch.ones(n, dtype=torch.bool, device=device)
    for _ in range(ev.core_max_steps()):
        player.action_test()
        rsum += player.reward * alive.unsqueeze(1)
        length += alive.to(length.dtype)
        alive &= (player.done == 0)
        if not bool(alive.any()):
            break
    ev.close()
    if was_training:
        model.train()
    return rsum[:episodes].cpu().numpy(), length[:episodes].cpu().numpy()


def schedule_train_modes(args, train_modes, n_iter, state):
    """test.py:84-92, verbatim semantics: tracker only while n_iter < --init-step; with --train-mode 2 the mode of every rank
    flips (1 - mode) whenever more than `iter_th` iterations have passed since the last flip — iter_th starts at --init-step
    and becomes --init-step after a flip to the tracker, --adv-step after a flip away from it — and is otherwise reset to
    --train-mode. `state` carries last_iter / iter_th between calls (locals of the reference's one long loop). The reference
    applies this at the end of every evaluation episode; here it runs once per evaluation round (all episodes of a round
    finish together)."""
    state.setdefault("last_iter", 0)
    state.setdefault("iter_th", args.init_step)
    for rank in range(len(train_modes)):
        if n_iter < args.init_step:
            train_modes[rank] = 0
        elif args.train_mode == 2 and n_iter - state["last_iter"] > state["iter_th"]:
            train_modes[rank] = 1 - train_modes[rank]
            state["last_iter"] = n_iter
            adv_step = getattr(args, "adv_step", None)
            if adv_step is None:
                raise AttributeError("--train-mode 2 needs --adv-step (test.py:90 of the reference reads args.adv_step, which "
                                     "its main.py never defines)")
            state["iter_th"] = args.init_step if train_modes[rank] == 0 else adv_step
        else:
            train_modes[rank] = args.train_mode
    return train_modes


def save_checkpoints(model, args, n_iter, best):
    """test.py:111-127."""
    check_path(args.log_dir)
    if best:
        model_dir = os.path.join(args.log_dir, 'all-best-{0}.dat'.format(n_iter))
        tracker_model_dir = os.path.join(args.log_dir, 'tracker-best.dat')
        target_model_dir = os.path.join(args.log_dir, 'target-best.dat')
    else:
        model_dir = os.path.join(args.log_dir, 'all-new.dat')
        tracker_model_dir = os.path.join(args.log_dir, 'tracker-new.dat')
        target_model_dir = os.path.join(args.log_dir, 'target-new.dat')
    cpu = lambda sd: {k: v.detach().cpu().clone() for k, v in sd.items()}
    torch.save(cpu(model.state_dict()), model_dir)
    if args.split:
        torch.save(cpu(model.player0.state_dict()), tracker_model_dir)
        if not args.single:
            torch.save(cpu(model.player1.state_dict()), target_model_dir)
    return model_dir


def test(args, shared_model, train_modes, n_iters, rounds=None, state=None):
    """Evaluator loop with the reference signature. `shared_model` is the (rank-0) replica being trained; call it
    between training iterations or from a side thread. `rounds` bounds the number of evaluation rounds (None = until
    the stop rule fires, as in the reference). `state`: a dict the caller keeps between bounded calls, so that the
    elapsed time, the best score and the one-off flag dump continue across them as in the reference's single loop."""
    gpu_id = args.gpu_ids[-1]
    device = torch.device('cuda:%d' % gpu_id)
    check_path(args.log_dir)
    name = '{}_log'.format(args.env)
    setup_logger(name, os.path.join(args.log_dir, 'logger'))
    log = logging.getLogger(name)
    state = {} if state is None else state
    if not state.get("started"):
        for k, v in vars(args).items():
            log.info('{0}: {1}'.format(k, v))
        state.update(started=True, start_time=time.time(), max_score=-100)
    if state.get("writer") is None:
        state["writer"] = ScalarWriter(os.path.join(args.log_dir, 'Test'))          # test.py:19
    writer = state["writer"]
    env_id = args.env if args.env_base is None else args.env_base
    start_time, max_score = state["start_time"], state["max_score"]
    done_rounds = 0
    while rounds is None or done_rounds < rounds:
        t0 = time.time()
        rsum, length = evaluate(shared_model, env_id, args, device, args.test_eps)
        n_iter = int(sum(n_iters))
        schedule_train_modes(args, train_modes, n_iter, state)                       # test.py:84-92
        # test.py:93-97: one record per evaluation episode at step n_iter. test/fps in the reference is the env steps per
        # second of the evaluator's one env; here the episodes of a round run as one batch, so it is the round's env steps
        # (all episodes) over its wall time
        fps = float(length.sum()) / max(time.time() - t0, 1e-9)
        for ep in range(len(length)):
            for i in range(rsum.shape[1]):
                writer.add_scalar('test/reward' + str(i), rsum[ep, i], n_iter)
            writer.add_scalar('test/fps', fps, n_iter)
            writer.add_scalar('test/eps_len', length[ep], n_iter)
        writer.flush()
        ave_reward_sum = rsum[:, :2].sum(0) / args.test_eps
        len_mean = length.sum() / args.test_eps
        reward_step = rsum[:, :2].sum(0) / max(length.sum(), 1)
        log.info("Time {0}, ave eps reward {1}, ave eps length {2}, reward step {3}".format(
            time.strftime("%Hh %Mm %Ss", time.gmtime(time.time() - start_time)), ave_reward_sum, len_mean, reward_step))
        best = ave_reward_sum[0] >= max_score
        if best:
            max_score = state["max_score"] = ave_reward_sum[0]
        save_checkpoints(shared_model, args, n_iter, best)
        done_rounds += 1
        if n_iter > args.max_step:                             # test.py:129-134
            for rank in range(len(train_modes)):
                train_modes[rank] = -100
            break
    return max_score
