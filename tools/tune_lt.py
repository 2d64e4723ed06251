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
libinfo = fused.lt_library()
out = {"note": "per problem: the kernel csrc/lt_gemm.cpp timed fastest — its position in hipblasLtMatmulAlgoGetHeuristic's list (512 "
               "requested, the workspace limit is part of the key) AND its solution index (hipblaslt_ext::getIndexFromAlgo). Only used "
               "when `hipblaslt` equals the loaded library's version + revision (fused.lt_choices); written by tools/tune_lt.py on an MI355X",
       "torch": torch.__version__, "hipblaslt": {"version": libinfo["version"], "git": libinfo["git"]},
       "header_version": libinfo["header_version"],
       "choices": {k: {"index": v["chosen"], "solution": v["solution"], "candidates": v["candidates"], "us": round(v["best_us"], 2),
                       "kernel": v["kernel"]} for k, v in sorted(ch.items())}}
json.dump(out, open(fused.LT_TUNING_FILE, "w"), indent=1)
print("hipBLASLt %s %s (header %s)" % (libinfo["version"], libinfo["git"], libinfo["header_version"]))
for k, v in sorted(ch.items()):
    print("%-100s candidate %3d of %3d  solution %6d  %.2f us  %s" % (k, v["chosen"], v["candidates"], v["solution"], v["best_us"], v["kernel"][:60]))
