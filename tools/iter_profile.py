"""Per-iteration kernel table of an A3C iteration (hipGraph replay): run under rocprofv3 --kernel-trace --stats, then
tools/summarize_prof.py; `calls / ITERS` and `total / ITERS` are per iteration (plus two eager warm-up iterations).
  python tools/iter_profile.py [ITERS] [env] [num_envs] [network] [aux] [train_mode]      (default: the headline config)"""
import sys
import torch
from active_tracking_rl_amd.train import GraphedIteration, default_args, make_player
ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 50
over = {}
if len(sys.argv) > 2:
    over = dict(env=sys.argv[2], num_envs=int(sys.argv[3]), network=sys.argv[4], aux=sys.argv[5], train_mode=int(sys.argv[6]))
dev = torch.device("cuda:0")
args = default_args(**over)
player, opt = make_player(args, dev)
it = GraphedIteration(player, opt, args)
for _ in range(ITERS):
    it.run()
torch.cuda.synchronize()
