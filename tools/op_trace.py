"""Which torch ops make the small kernels of one A3C iteration: one eager iteration of the headline configuration under
torch.profiler, aggregated by (op, python source line). Output: ops sorted by launch count."""
import sys
from collections import defaultdict

import torch
from torch.profiler import ProfilerActivity, profile

from active_tracking_rl_amd.train import default_args, make_player, rollout

over = dict(env="Track2D-BlockPartialPZR-v0", num_envs=4096, network="tat-maze-lstm", aux="reward", train_mode=-1)
if len(sys.argv) > 1:   # python tools/op_trace.py <env> <num_envs> <network> <aux> <train_mode>
    over = dict(env=sys.argv[1], num_envs=int(sys.argv[2]), network=sys.argv[3], aux=sys.argv[4], train_mode=int(sys.argv[5]))
args = default_args(**over)
dev = torch.device("cuda:0")
player, opt = make_player(args, dev, 0, 1)


def it():
    rollout(player, args.num_steps)
    player.optimize(None, opt, player.model, args.train_mode, dev)


for _ in range(3):
    it()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    it()
    torch.cuda.synchronize()
ev = prof.events()
agg = defaultdict(lambda: [0, 0.0])
for e in ev:
    if e.device_type.name != "CPU" or not e.kernels:
        continue
    # innermost repo frame of the python stack
    where = "?"
    for fr in (e.stack or []):
        if "/active_tracking_rl_amd/" in fr or "main.py" in fr:
            where = fr.split("/active_tracking_rl_amd/")[-1] if "/active_tracking_rl_amd/" in fr else fr
            break
    for k in e.kernels:
        a = agg[(e.name, where, k.name[:70])]
        a[0] += 1
        a[1] += k.duration
rows = sorted(agg.items(), key=lambda kv: -kv[1][0])
tot = sum(v[0] for v in agg.values())
print("# %d kernel launches in one eager iteration" % tot)
for (op, where, kn), (n, us) in rows:
    print("%4d %8.1f us  %-28s %-58s %s" % (n, us, op[:28], where[:58], kn))
