"""Re-generate active_tracking_rl_amd/tunableop_gfx950.csv on an MI355X: runs the BASELINE training iteration
eagerly with PyTorch TunableOp in tuning mode so every GEMM shape of the path is searched once."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from active_tracking_rl_amd import gemm_tuning  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else gemm_tuning.RESULTS
if os.path.exists(out):
    os.remove(out)
assert gemm_tuning.enable(tune=True, filename=out)
from active_tracking_rl_amd.train import default_args, make_player, rollout  # noqa: E402

dev = torch.device("cuda:0")
for net, aux in (("tat-maze-lstm", "reward"), ("maze-lstm", "none")):
    for n in (4096, 2048, 1024, 512):       # the headline batch and the strong-scaling shard sizes
        args = default_args(network=net, aux=aux, num_envs=n)
        player, opt = make_player(args, dev)
        for _ in range(2):
            rollout(player, args.num_steps)
            player.optimize(None, opt, player.model, -1, dev)
        torch.cuda.synchronize()
        player.env.close()
torch.cuda.tunable.write_file(out) if hasattr(torch.cuda.tunable, "write_file") else None
print("wrote", out, sum(1 for _ in open(out)), "lines")
