import torch, sys
from active_tracking_rl_amd.vec_env import VecTrack2D
n=1024
env=VecTrack2D("Track2D-MazePartialNav-v0", num_envs=n, seed=1)
env.reset()
out=None
tot=0
for it in range(10):
    o,r,d=env.step_random(20, 7)
    tot+=20
    torch.cuda.synchronize()
    f=env.faults()
    print("steps", tot, "inline re-plans per step: episodes <= 20 steps old %.2f, older %.2f" % (((f >> 8) & 0xfff) / tot, (f >> 20) / tot), flush=True)
