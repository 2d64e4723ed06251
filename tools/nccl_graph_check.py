"""Single-rank RCCL smoke: init the nccl backend with world_size 1 and interleave an eager all_reduce of the flat
gradient bucket between the two hipGraph replays of an iteration (the N>1 schedule of train.GraphedIteration)."""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from active_tracking_rl_amd.train import GraphedIteration, default_args, make_player  # noqa: E402

dev = torch.device("cuda:0")
w = torch.zeros(1, device=dev)
dist.all_reduce(w)
torch.cuda.synchronize()
args = default_args(num_envs=1024)
player, opt = make_player(args, dev)
it = GraphedIteration(player, opt, args)
t0 = time.time()
for i in range(30):
    it.g_roll.replay()
    dist.all_reduce(opt.bucket.grad, op=dist.ReduceOp.SUM)
    opt.bucket.grad.div_(1)
    it.g_opt.replay()
torch.cuda.synchronize()
print("ok: 30 graphed iterations with eager RCCL all_reduce in between, %.1f ms/iter" % ((time.time() - t0) * 1e3 / 30),
      "finite:", bool(torch.isfinite(opt.bucket.flat).all()))
dist.destroy_process_group()
