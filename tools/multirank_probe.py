"""What ONE GPU can say about the multi-rank configuration: the pipelined / synchronous schedules at the
per-GPU shard sizes of the strong form, inside an RCCL process group of ONE rank with the gradient all-reduce really issued
(ATR_FORCE_ALLREDUCE=1: the collective's stream hand-overs and host work are the multi-rank ones; the transfer itself is not —
a 1-rank all-reduce moves nothing), for
    * the HIP runtime's hardware-queue budget: the default against GPU_MAX_HW_QUEUES=2 (set by the caller, before HIP starts:
      tools/multirank_probe.sh runs this file once per setting), which decides whether the CU-partitioned stream pair wins;
    * the all-reduce captured inside the update graph against eager between the learner's graph and the update graph.
One line per (envs, schedule, capture) with ms per iteration (median of 5 regions of 60 iterations) and the stream pair the trial
picked. Run under torch.distributed.run --nproc-per-node 1 (rendezvous on 127.0.0.1)."""
import os
import statistics
import sys
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ["ATR_FORCE_ALLREDUCE"] = "1"
rank = int(os.environ.get("RANK", "0"))
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
if "RANK" in os.environ:
    dist.init_process_group("nccl", device_id=dev)
else:
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
w = torch.ones(1, device=dev)
dist.all_reduce(w)                     # brings the communicator (and its stream) up
torch.cuda.synchronize()

from active_tracking_rl_amd.train import GraphedIteration, PipelinedIteration, default_args, make_player  # noqa: E402

sizes = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["512", "1024"])]
queues = os.environ.get("GPU_MAX_HW_QUEUES", "default")
for n in sizes:
    for sched in ("pipelined", "synchronous"):
        for cap in (False, True):
            os.environ["ATR_CAPTURE_ALLREDUCE"] = "1" if cap else "0"
            args = default_args(num_envs=n)
            player, opt = make_player(args, dev)
            try:
                if sched == "pipelined":
                    it = PipelinedIteration(player, opt, args)
                    trials = it.tune_streams()
                    pick = [lb for _, c, lb in trials if c][0]
                    drain = it.finish
                else:
                    it = GraphedIteration(player, opt, args)
                    pick, drain = "-", (lambda: None)
                for _ in range(10):
                    it.run()
                drain()
                torch.cuda.synchronize()
                ts = []
                for _ in range(5):
                    t0 = time.perf_counter()
                    for _ in range(60):
                        it.run()
                    drain()
                    torch.cuda.synchronize()
                    ts.append((time.perf_counter() - t0) / 60 * 1e3)
                ms = statistics.median(ts)
                print("queues=%-7s envs=%4d %-11s all-reduce %-8s %7.3f ms/iter %6.2f M env steps/s   streams: %s" % (
                    queues, n, sched, "captured" if cap else "eager", ms, n * args.num_steps / ms / 1e3, pick), flush=True)
            except Exception as ex:
                print("queues=%-7s envs=%4d %-11s all-reduce %-8s FAILED: %r" % (queues, n, sched, "captured" if cap else "eager", ex),
                      flush=True)
            player.env.close()
            del it, player, opt
            torch.cuda.empty_cache()
dist.destroy_process_group()
