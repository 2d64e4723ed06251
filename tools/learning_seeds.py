"""Learning sanity across seeds and schedules in ONE process (VERDICT r04 "Next round" 2): for every (schedule, seed) a fresh
player trains `--iters` iterations; the mean tracker reward per env step over windows of 100 iterations is recorded at every
`--every` iterations. Schedules: synchronous (GraphedIteration), pipelined (PipelinedIteration on two streams),
pipelined-serial (the same double-buffered dataflow in program order on one stream).

    python tools/learning_seeds.py --env Track2D-MazePartialNav-v0 --num-envs 1024 --network maze-lstm --train-mode 0 \
        --iters 3000 --seeds 1 2 3 4 5 6 7 8 > profiles/r05_learning_seeds_nav_mode0.txt
"""
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")   # (see bench.py: room for the CU-partitioned stream pair)

import argparse
import time

import torch

from active_tracking_rl_amd.train import GraphedIteration, PipelinedIteration, default_args, make_player

ap = argparse.ArgumentParser()
ap.add_argument("--env", default="Track2D-MazePartialNav-v0")
ap.add_argument("--iters", type=int, default=3000)
ap.add_argument("--every", type=int, default=500)
ap.add_argument("--num-envs", type=int, default=1024)
ap.add_argument("--network", default="maze-lstm")
ap.add_argument("--train-mode", type=int, default=0)
ap.add_argument("--seeds", type=int, nargs="+", default=[1, 2, 3, 4, 5, 6, 7, 8])
ap.add_argument("--schedules", nargs="+", default=["synchronous", "pipelined", "pipelined-serial"])
ap.add_argument("--burn-in", type=int, default=0,
                help="iterations of the schedule run BEFORE training whose updates are rolled back (weights, optimizer state, counters "
                     "restored): the env shard keeps advancing under the initial policy, so the envs' episode phases drift apart — "
                     "what PipelinedIteration.tune_streams() does as a side effect (2 passes x ~7 stream pairs x 10 iterations)")
ap.add_argument("--no-tune", action="store_true", help="pipelined: skip tune_streams() (and with it its ~140 untrained iterations)")
ap.add_argument("--trace-seeds", type=int, nargs="*", default=[],
                help="for these seeds also print, every --trace-every iterations up to --trace-iters: the tracker's entropy per step, "
                     "value loss, policy loss (the iteration's loss statistics), its action histogram over the rollout just made "
                     "and its mean reward per step")
ap.add_argument("--lr", type=float, default=None, help="learning rate (default: main.py's 1e-3)")
ap.add_argument("--entropy", type=float, default=None, help="the tracker's entropy weight (default: main.py's 0.01)")
ap.add_argument("--trace-iters", type=int, default=200)
ap.add_argument("--trace-every", type=int, default=10)
a = ap.parse_args()
dev = torch.device("cuda:0")
print("# %s  %d envs  %s  train-mode %d  %d iterations; mean tracker (target) reward per env step over the 100 iterations before "
      "each checkpoint" % (a.env, a.num_envs, a.network, a.train_mode, a.iters), flush=True)
marks = list(range(a.every, a.iters + 1, a.every))
print("# burn-in %d untrained iterations%s%s%s" % (a.burn_in, "; pipelined without tune_streams()" if a.no_tune else "",
                                                   "; lr %g" % a.lr if a.lr is not None else "",
                                                   "; entropy weight %g" % a.entropy if a.entropy is not None else ""), flush=True)
print("# schedule seed " + " ".join("@%d" % m for m in marks), flush=True)
final = {}
for sched in a.schedules:
    for seed in a.seeds:
        args = default_args(env=a.env, network=a.network, aux="reward" if "tat" in a.network else "none",
                            train_mode=a.train_mode, num_envs=a.num_envs, seed=seed)
        if a.lr is not None:
            args.lr = a.lr
        if a.entropy is not None:
            args.entropy = a.entropy
        player, opt = make_player(args, dev)
        pipelined = sched != "synchronous"
        if pipelined:
            it = PipelinedIteration(player, opt, args, serial=(sched == "pipelined-serial"))
            if not it.serial and not a.no_tune:
                it.tune_streams()
        else:
            it = GraphedIteration(player, opt, args)
        if a.burn_in > 0:
            tensors = it._schedule_tensors() if pipelined else it._optimizer_tensors()
            saved = [t.clone() for t in tensors]
            i0 = getattr(it, "i", 0)
            n0 = (it.master if pipelined else player).n_steps
            for _ in range(a.burn_in + (a.burn_in & 1)):
                it.run()
            if pipelined:
                it.finish()
            torch.cuda.synchronize()
            with torch.no_grad():
                for t, v in zip(tensors, saved):
                    t.copy_(v)
            if pipelined:
                it.i, it.master.n_steps = i0, n0
            else:
                player.n_steps = n0
            torch.cuda.synchronize()
        rew_acc = torch.zeros(2, device=dev)
        row, t0 = [], time.time()
        for i in range(1, a.iters + 1):
            it.run()
            in_window = any(m - 100 < i <= m for m in marks)
            if in_window:
                if pipelined:     # the replica that just rolled out, read on the stream its rollout ran on
                    src = it.players[(it.i - 1) & 1]
                    if it.serial:
                        rew_acc += src.reward.mean(0)
                    else:
                        with torch.cuda.stream(it.sR):
                            rew_acc += src.reward.mean(0)
                else:
                    rew_acc += player.reward.mean(0)
            if seed in a.trace_seeds and i <= a.trace_iters and (i % a.trace_every == 0 or i <= 5):
                if pipelined:
                    it.finish()
                torch.cuda.synchronize()
                src = it.players[(it.i - 1) & 1] if pipelined else player
                st = [x.detach().float().reshape(-1).tolist() for x in it.stats]        # policy, value, entropy [, aux]: per player
                acts = src._actions_buf[:, 0].reshape(-1)
                hist = torch.bincount(acts, minlength=4).float() / acts.numel()
                print("trace %-17s seed %2d it %4d  entropy/step %.4f  value loss %.4f  policy loss %+.4f  tracker actions %s  reward/step %+.3f"
                      % (sched, seed, i, st[2][0] / args.num_steps, st[1][0], st[0][0], " ".join("%.2f" % h for h in hist.tolist()),
                         src.reward.mean(0)[0].item()), flush=True)
            if i in marks:
                if pipelined:
                    it.finish()
                torch.cuda.synchronize()
                row.append((rew_acc[0].item() / 100, rew_acc[1].item() / 100))
                rew_acc.zero_()
                torch.cuda.synchronize()
        if pipelined:
            it.finish()
        torch.cuda.synchronize()
        print("%-17s %2d  " % (sched, seed) + "  ".join("%+.3f (%+.3f)" % r for r in row) + "   [%.0fs]" % (time.time() - t0),
              flush=True)
        final.setdefault(sched, []).append(row[-1][0])
        player.env.close()
        del it, player, opt
        torch.cuda.empty_cache()
print("# final tracker reward, mean / min / max over seeds")
for sched, v in final.items():
    print("# %-17s %+.3f / %+.3f / %+.3f" % (sched, sum(v) / len(v), min(v), max(v)))
