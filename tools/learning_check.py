"""Learning sanity check: train the tracker against the scripted Ram target and print the mean tracker reward per
env-step and the done rate over time (both should improve markedly over a random-policy start)."""
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")   # (see bench.py: room for the CU-partitioned stream pair)

import argparse
import time

import torch

from active_tracking_rl_amd.train import GraphedIteration, PipelinedIteration, default_args, make_player

ap = argparse.ArgumentParser()
ap.add_argument("--env", default="Track2D-BlockPartialRam-v0")
ap.add_argument("--iters", type=int, default=1500)
ap.add_argument("--num-envs", type=int, default=4096)
ap.add_argument("--network", default="maze-lstm")
ap.add_argument("--train-mode", type=int, default=0)
ap.add_argument("--schedule", choices=("synchronous", "pipelined"), default="pipelined")
ap.add_argument("--seed", type=int, default=None, help="default: default_args' seed")
a = ap.parse_args()
dev = torch.device("cuda:0")
args = default_args(env=a.env, network=a.network, aux="reward" if "tat" in a.network else "none", train_mode=a.train_mode,
                    num_envs=a.num_envs, **({"seed": a.seed} if a.seed is not None else {}))
player, opt = make_player(args, dev)
pipelined = a.schedule == "pipelined"
it = PipelinedIteration(player, opt, args) if pipelined else GraphedIteration(player, opt, args)
if pipelined:
    it.tune_streams()
print("schedule: %s" % a.schedule, flush=True)
rew_acc = torch.zeros(2, device=dev)
done_acc = torch.zeros((), device=dev)
t0 = time.time()
for i in range(1, a.iters + 1):
    it.run()
    if pipelined:     # the replica that just rolled out, read on the rollout stream (ahead of the next rollout's writes)
        src = it.players[(it.i - 1) & 1]
        with torch.cuda.stream(it.sR):
            rew_acc += src.reward.mean(0)
            done_acc += src.done.float().mean()
    else:
        rew_acc += player.reward.mean(0)          # last step of the rollout (static buffer of the captured graph)
        done_acc += player.done.float().mean()
    if i % 100 == 0:
        if pipelined:
            it.finish()
        torch.cuda.synchronize()
        print("iter %5d  env-steps %9d  mean reward/step tracker %+.3f target %+.3f  done rate %.4f  (%.1fs)" % (
            i, i * args.num_steps * a.num_envs, rew_acc[0].item() / 100, rew_acc[1].item() / 100,
            done_acc.item() / 100, time.time() - t0))
        rew_acc.zero_(); done_acc.zero_()
        torch.cuda.synchronize()
