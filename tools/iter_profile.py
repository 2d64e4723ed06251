"""Per-iteration kernel table of an A3C iteration (hipGraph replay): run under rocprofv3 --kernel-trace --stats, then
tools/summarize_prof.py; `calls / ITERS` and `total / ITERS` are per iteration (plus two eager warm-up iterations).
  python tools/iter_profile.py [ITERS] [env] [num_envs] [network] [aux] [train_mode]      (default: the headline config)
ITER_PROFILE_SCHEDULE=pipelined profiles the two-stream schedule instead (kernel durations BESIDE the other chain)."""
import os
import sys
import torch
from active_tracking_rl_amd.train import GraphedIteration, PipelinedIteration, default_args, make_player
ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 50
over = {}
if len(sys.argv) > 2:
    over = dict(env=sys.argv[2], num_envs=int(sys.argv[3]), network=sys.argv[4], aux=sys.argv[5], train_mode=int(sys.argv[6]))
dev = torch.device("cuda:0")
args = default_args(**over)
player, opt = make_player(args, dev)
if os.environ.get("ITER_PROFILE_SCHEDULE") == "pipelined":
    it = PipelinedIteration(player, opt, args)
    it.tune_streams()
else:
    it = GraphedIteration(player, opt, args)
for _ in range(ITERS):
    it.run()
if hasattr(it, "finish"):
    it.finish()
torch.cuda.synchronize()
