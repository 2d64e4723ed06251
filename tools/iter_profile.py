"""Per-iteration kernel table of the headline A3C iteration (hipGraph replay): run under rocprofv3 --kernel-trace --stats,
then tools/summarize_prof.py; `calls / ITERS` and `total / ITERS` are per iteration."""
import sys
import torch
from active_tracking_rl_amd.train import GraphedIteration, default_args, make_player
ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda:0")
args = default_args()
player, opt = make_player(args, dev)
it = GraphedIteration(player, opt, args)
for _ in range(ITERS):
    it.run()
torch.cuda.synchronize()
