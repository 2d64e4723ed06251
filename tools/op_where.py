"""Which python line launches each small kernel of the LEARNER (compute_grads + optimizer) of one eager iteration:
TorchDispatchMode over the ops, python stack of the innermost repo frame.   python tools/op_where.py [num_envs]"""
import sys
import traceback
from collections import OrderedDict

import torch
from torch.utils._python_dispatch import TorchDispatchMode

from active_tracking_rl_amd.train import default_args, make_player, rollout

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
args = default_args(env="Track2D-BlockPartialPZR-v0", num_envs=n, network="tat-maze-lstm", aux="reward", train_mode=-1)
dev = torch.device("cuda:0")
player, opt = make_player(args, dev, 0, 1)
for _ in range(2):
    rollout(player, args.num_steps)
    player.optimize(None, opt, player.model, args.train_mode, dev)
rollout(player, args.num_steps)
seen = OrderedDict()


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(k in name for k in ("view", "reshape", "as_strided", "detach", "alias", "slice", "select", "unsqueeze",
                                       "squeeze", "transpose", "permute", "expand", "t.default", "unbind", "empty", "_unsafe_view")):
            where = "?"
            for fr in reversed(traceback.extract_stack()):
                if "/active_tracking_rl_amd/" in fr.filename and "op_where" not in fr.filename:
                    where = "%s:%d %s" % (fr.filename.split("/active_tracking_rl_amd/")[-1], fr.lineno, fr.name)
                    break
            key = (name, where)
            seen[key] = seen.get(key, 0) + 1
        return func(*args, **(kwargs or {}))


with Spy():
    player.optimize(None, opt, player.model, args.train_mode, dev)
torch.cuda.synchronize()
for (name, where), c in seen.items():
    print("%3d  %-40s %s" % (c, name, where))
