#!/usr/bin/env python
"""main.py — drop-in for the reference's main.py (same flags, main.py:16-50) on the MI355X-native path.

    python main.py --shared-optimizer --split --train-mode -1 --env Track2D-BlockPartialPZR-v0 --num-envs 4096
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 main.py ... (one rank per GPU)

What changes: `--workers` CPU processes become `--num-envs` envs per GPU stepped by the HIP kernels; the Hogwild shared
model becomes data parallel with one all-reduce of the flat gradient bucket per update — `--schedule pipelined` (default:
the next rollout runs while the learner of the last one does, every gradient exactly one update late — the bounded form
of the reference's worker asynchrony) or `--schedule synchronous` (no delay); the evaluator (`test`) runs on rank 0
between training iterations instead of in a forked process. `--gpu-ids` picks the device of a
single-process run; under torch.distributed.run each rank uses LOCAL_RANK.
"""
from __future__ import print_function, division
import os
os.environ["OMP_NUM_THREADS"] = "1"
import argparse
import sys
import time
from datetime import datetime


def _wants_two_hw_queues(argv):
    """The pipelined schedule's CU partition (train.PipelinedIteration.tune_streams) gives each chain a CU-masked stream with a
    hardware queue of its own; iterations slow down 2-3x once a process owns more than four hardware queues (measured), so the
    ordinary streams share two. The HIP runtime reads GPU_MAX_HW_QUEUES when it initialises: the decision is taken from the raw
    command line HERE, before torch (or anything else that may touch the runtime at import time) is imported — and only for
    the one configuration that builds a PipelinedIteration on a single GPU. (N > 1 keeps the runtime default: measured with
    a 1-rank RCCL group, profiles/r04_multirank_*.)"""
    pre = argparse.ArgumentParser(add_help=False)
    pre.add_argument('--schedule', default='pipelined')
    pre.add_argument('--no-graph', action='store_true')
    known, _ = pre.parse_known_args(argv)
    return (known.schedule == 'pipelined' and not known.no_graph and int(os.environ.get("WORLD_SIZE", "1")) == 1
            and "GPU_MAX_HW_QUEUES" not in os.environ)


if __name__ == '__main__' and _wants_two_hw_queues(sys.argv[1:]):
    os.environ["GPU_MAX_HW_QUEUES"] = "2"

import torch
import torch.distributed as dist

from active_tracking_rl_amd import build
from active_tracking_rl_amd.test import test
from active_tracking_rl_amd.train import GraphedIteration, PipelinedIteration, make_player, sync_train_modes
from active_tracking_rl_amd.utils import ScalarWriter, log_train_scalars

parser = argparse.ArgumentParser(description='A3C (MI355X data-parallel)')
parser.add_argument('--lr', type=float, default=0.001, metavar='LR', help='learning rate (2D: 0.001, 3D: 0.0001)')
parser.add_argument('--gamma', type=float, default=0.9, metavar='G', help='discount factor for rewards (default: 0.9)')
parser.add_argument('--tau', type=float, default=1.00, metavar='T', help='parameter for GAE (default: 1.00)')
parser.add_argument('--entropy', type=float, default=0.01, metavar='E', help='parameter for entropy(for tracker)')
parser.add_argument('--entropy-target', type=float, default=0.2, metavar='EC', help='parameter for entropy(for target)')
parser.add_argument('--seed', type=int, default=1, metavar='S', help='random seed (default: 1)')
parser.add_argument('--workers', type=int, default=1, metavar='W', help='accepted for compatibility (see --num-envs)')
parser.add_argument('--num-envs', type=int, default=4096, metavar='N', help='parallel envs per GPU')
parser.add_argument('--num-steps', type=int, default=20, metavar='NS', help='number of forward steps in A3C')
parser.add_argument('--test-eps', type=int, default=100, metavar='TE', help='evaluation episodes per round')
parser.add_argument('--test-every', type=int, default=200, metavar='TI', help='training iterations between evaluations')
parser.add_argument('--env', default='Track2D-BlockPartialPZR-v0', metavar='ENV', help='environment to train on')
parser.add_argument('--env-base', default='Track2D-BlockPartialNav-v0', metavar='ENVB', help='environment to test on ')
parser.add_argument('--optimizer', default='Adam', metavar='OPT', help='shares optimizer choice of Adam or RMSprop')
parser.add_argument('--amsgrad', default=True, metavar='AM', help='Adam optimizer amsgrad parameter')
parser.add_argument('--load-model-dir', default=None, metavar='LMD', help='checkpoint to load')
parser.add_argument('--log-dir', default='logs/', metavar='LG', help='folder to save logs')
parser.add_argument('--network', default='tat-maze-lstm', metavar='M', help='config Network Architecture')
parser.add_argument('--aux', default='reward', metavar='A', help='auxiliary task: reward/none')
parser.add_argument('--gpu-ids', type=int, default=[0], nargs='+', help='GPU to use in a single-process run')
parser.add_argument('--obs', default='img', metavar='O', help='img or vector')
parser.add_argument('--single', dest='single', action='store_true', help='run on single agent env')
parser.add_argument('--gray', dest='gray', action='store_true', help='gray image')
parser.add_argument('--crop', dest='crop', action='store_true', help='crop image')
parser.add_argument('--inv', dest='inv', action='store_true', help='inverse image')
parser.add_argument('--rescale', dest='rescale', action='store_true', help='rescale image to [-1, 1]')
parser.add_argument('--render', dest='render', action='store_true', help='(not supported on the batched path)')
parser.add_argument('--shared-optimizer', dest='shared_optimizer', action='store_true',
                    help='SharedAdam / SharedRMSprop numerics; without it torch.optim.Adam / RMSprop numerics (train.py:45-49)')
parser.add_argument('--split', dest='split', action='store_true', help='split model to save')
parser.add_argument('--train-mode', type=int, default=-1, metavar='TM', help='which agent to train(0:tracker 1:target)')
parser.add_argument('--stack-frames', type=int, default=1, metavar='SF', help='Choose number of observations to stack')
parser.add_argument('--input-size', type=int, default=80, metavar='IS', help='input image size')
parser.add_argument('--rnn-out', type=int, default=128, metavar='LO', help='rnn output size')
parser.add_argument('--sleep-time', type=int, default=0, metavar='ST', help='accepted for compatibility')
parser.add_argument('--max-step', type=int, default=150000, metavar='MS', help='max learning steps (iterations)')
parser.add_argument('--init-step', type=int, default=-1, metavar='IS', help='steps not update target at beginning')
parser.add_argument('--max-grad-norm', type=float, default=None, help='clip (off by default, as the reference effectively is)')
parser.add_argument('--f32-obs', dest='obs_u8', action='store_false', help='float32 observations between env and policy (default: bytes, decoded in conv1)')
parser.add_argument('--no-graph', action='store_true', help='run iterations eagerly instead of as hipGraphs')
parser.add_argument('--schedule', choices=('pipelined', 'synchronous'), default='pipelined',
                    help='pipelined: rollout i+1 overlaps learner i on a second HIP stream (gradients one update late); '
                         'synchronous: rollout, learner, update in sequence')
parser.add_argument('--log-every', type=int, default=100, metavar='LE',
                    help='training iterations between train/* scalar records (each record joins both streams of the pipelined '
                         'schedule and reads ~10 scalars back: keep it well above 1)')
parser.add_argument('--burn-in', type=int, default=150, metavar='BI',
                    help='iterations before training whose updates are discarded: the envs of a fresh shard all start an episode '
                         'at step 0 together, and the first updates would fit that one phase (0: start synchronised, as a single '
                         'reference worker does)')
parser.add_argument('--adv-step', type=int, default=None, metavar='AS',
                    help="--train-mode 2 only: iterations the TARGET trains before the evaluator hands back to the tracker "
                         "(test.py:88-91 of the reference reads args.adv_step, which its own main.py never defines)")

if __name__ == '__main__':
    args = parser.parse_args()
    if args.train_mode == 2 and args.adv_step is None:
        parser.error("--train-mode 2 (tracker / target alternation, test.py:88-91) needs --adv-step: the reference reads "
                     "args.adv_step there without defining the flag and stops with an AttributeError at the first switch")
    if args.max_grad_norm is not None and args.max_grad_norm <= 0:
        parser.error("--max-grad-norm must be positive")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # (GPU_MAX_HW_QUEUES for the single-GPU pipelined schedule was decided before torch was imported: _wants_two_hw_queues)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(args.gpu_ids[0])))
    # test hooks (as bench.py's BENCH_*): N ranks sharing ONE device over gloo exercise the multi-rank code path on a 1-GPU box
    backend = os.environ.get("ATR_DIST_BACKEND", "nccl")
    if os.environ.get("ATR_SINGLE_DEVICE") == "1":
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    args.gpu_ids = [local_rank]
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    if rank == 0:
        build.build()
    if world > 1:
        dist.barrier()
    args.log_dir = os.path.join(args.log_dir, args.env, datetime.now().strftime('%b%d_%H-%M'))
    player, optimizer = make_player(args, device, rank, world)
    if args.load_model_dir is not None:
        saved_state = torch.load(args.load_model_dir, map_location=lambda storage, loc: storage)
        player.model.load_state_dict(saved_state)
    # until the evaluator first speaks, the schedule of test.py:84-92 applies from iteration 0: tracker only while
    # n_iter < --init-step
    first_mode = 0 if args.init_step > 0 else args.train_mode
    train_modes, n_iters = [first_mode] * world, [0] * world
    sched = None
    if not args.no_graph:
        if args.schedule == 'pipelined':
            sched = PipelinedIteration(player, optimizer, args, mode=first_mode)
            sched.tune_streams()
        else:
            sched = GraphedIteration(player, optimizer, args, mode=first_mode)
        if args.burn_in > 0 and args.load_model_dir is None:
            # (not counted in n_steps or the logs: the updates are discarded, only the episode clocks of the shard move on)
            sched.burn_in(args.burn_in, first_mode)
            if rank == 0:
                print("burn-in: %d iterations (%d env steps per rank) with their updates discarded"
                      % (args.burn_in, args.burn_in * args.num_steps * player.num_envs), file=sys.stderr, flush=True)
        elif args.burn_in > 0 and rank == 0:
            print("burn-in skipped: resuming from --load-model-dir", file=sys.stderr, flush=True)
    elif args.burn_in > 0 and rank == 0:
        print("warning: --burn-in %d ignored with --no-graph (the eager loop has no rollback of its updates)" % args.burn_in,
              file=sys.stderr, flush=True)
    step = sched.run if sched is not None else None
    drain = getattr(sched, "finish", lambda: None)          # pipelined: both streams joined before the host reads anything
    it = 0
    eval_state = {}
    # train/* scalars of train.py:97-104 from the graphed path: the loss statistics of an iteration are static outputs of the
    # captured graph (device tensors the next replay overwrites), so reading them costs one synchronisation per --log-every
    # iterations and nothing in between; train/fps = env steps of this rank since the last record / wall time
    writer = ScalarWriter(os.path.join(args.log_dir, 'Agent:{}'.format(rank)))
    t_log, it_log = time.time(), 0
    while True:
        if train_modes[rank] == -100:                       # the evaluator's stop sentinel (test.py:129-134)
            break
        if step is not None:
            stats = step(train_modes[rank])                 # one hipGraph per training mode (test.py:84-92 schedule)
        else:
            from active_tracking_rl_amd.train import rollout
            rollout(player, args.num_steps)
            stats = player.optimize(None, optimizer, player.model, train_modes[rank], device)
        it += 1
        n_iters[:] = [it] * world
        if args.log_every > 0 and it % args.log_every == 0:
            drain()
            torch.cuda.synchronize(device)
            faults = player.env.core.faults() if hasattr(getattr(player.env, "core", None), "faults") else 0
            if faults:        # sticky device fault word (include/track2d.h t2d_get_faults): bits 1-3 = the cooperative step's barriers
                raise RuntimeError("env handle reports device faults 0x%x at iteration %d: the rollout state is not trustworthy "
                                   "(ATR_COOP_STEP=0 selects the four-launch step)" % (faults, it))
            now = time.time()
            fps = (it - it_log) * args.num_steps * player.num_envs / max(now - t_log, 1e-9)
            log_train_scalars(writer, stats, train_modes[rank], fps, it * args.num_steps * player.num_envs, player.num_agents)
            writer.flush()
            t_log, it_log = time.time(), it
        if it % args.test_every == 0 or it > args.max_step:
            drain()
            torch.cuda.synchronize(device)
            if rank == 0:
                test(args, player.model, train_modes, n_iters, rounds=1, state=eval_state)
            sync_train_modes(train_modes, device)           # rank 0 owns the schedule; a broadcast is also a barrier
            t_log, it_log = time.time(), it                 # (the evaluation's wall time is not training time)
        if it > args.max_step:
            break
    drain()
    torch.cuda.synchronize(device)
    if world > 1:       # synchronous data parallel: every rank applied the same all-reduced gradients to the same start
        flat = getattr(optimizer, "bucket", None)
        if flat is not None:
            ref = flat.flat.detach().clone()
            dist.broadcast(ref, src=0)
            same = torch.tensor([1 if torch.equal(ref, flat.flat) else 0], dtype=torch.int64, device=device)
            dist.all_reduce(same, op=dist.ReduceOp.MIN)
            if rank == 0:
                print("replicas bit-identical across %d ranks after %d iterations: %s" % (world, it, bool(same.item())),
                      file=sys.stderr, flush=True)
            if not bool(same.item()):
                raise RuntimeError("rank %d: master weights differ from rank 0's after %d iterations" % (rank, it))
    writer.close()
    player.env.close()
    if world > 1:
        dist.destroy_process_group()
