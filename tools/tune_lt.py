"""Record the kernel choices of csrc/lt_gemm.cpp (hipBLASLt called directly) for the shapes the rollout step uses, into
active_tracking_rl_amd/lt_tuning_gfx950.json: every BASELINE configuration's shard size is run for one eager rollout with
ATR_LT_TUNING=0 (time the library's candidates), the winners are written as {problem key: candidate index}. Later runs pre-select
them (fused.lt_choices) — same kernels in every run, nothing timed at first use (what tunableop_gfx950.csv is for torch's GEMMs).
    ATR_LT_TUNING=0 python tools/tune_lt.py"""
import json
import os
os.environ["ATR_LT_TUNING"] = "0"

import torch

from active_tracking_rl_amd import fused
from active_tracking_rl_amd.train import default_args, make_player, rollout

dev = torch.device("cuda:0")
CASES = [dict(env="Track2D-BlockPartialPZR-v0", network="tat-maze-lstm", aux="reward", train_mode=-1, num_envs=n) for n in (768, 1024, 2048, 4096, 8192)]
CASES += [dict(env="Track2D-BlockPartialRam-v0", network="maze-lstm", aux="none", train_mode=0, num_envs=n) for n in (1024, 2048, 4096)]
for over in CASES:
    args = default_args(**over)
    player, opt = make_player(args, dev)
    for _ in range(2):
        rollout(player, args.num_steps)
        player.optimize(None, opt, player.model, args.train_mode, dev)
    torch.cuda.synchronize()
    player.env.close()
    del player, opt
    torch.cuda.empty_cache()
ch = fused.lt_chosen()
out = {"note": "candidate index (position in hipblasLtMatmulAlgoGetHeuristic's list for the problem, 512 requested) of the kernel "
               "csrc/lt_gemm.cpp timed fastest; written by tools/tune_lt.py on an MI355X",
       "torch": torch.__version__, "choices": {k: v[0] for k, v in sorted(ch.items())},
       "timing_us": {k: round(v[2], 2) for k, v in sorted(ch.items())}, "candidates": {k: v[1] for k, v in sorted(ch.items())}}
json.dump(out, open(fused.LT_TUNING_FILE, "w"), indent=1)
for k, v in sorted(ch.items()):
    print("%-90s candidate %3d of %3d  %.2f us" % (k, v[0], v[1], v[2]))
