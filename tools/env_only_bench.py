"""Env-only throughput probe: random-policy steps with on-device actions, one launch per step (default) or the
persistent mode of SURVEY.md 8(d)(ii) (--fused: up to 10 steps per launch, every step's outputs written)."""
import argparse
import time

import torch

from active_tracking_rl_amd.vec_env import VecTrack2D

ap = argparse.ArgumentParser()
ap.add_argument("--env", default="Track2D-BlockPartialPZR-v0")
ap.add_argument("--n", type=int, default=4096)
ap.add_argument("--steps", type=int, default=2000)
ap.add_argument("--warmup", type=int, default=200)
ap.add_argument("--no-auto-reset", action="store_true")
ap.add_argument("--no-obs", action="store_true")
ap.add_argument("--fused", action="store_true", help="t2d_rollout_random: up to 10 steps per launch, all outputs kept")
ap.add_argument("--pregrow", action="store_true", help="Maze maps grown ahead of the generator pass, forked behind it (t2d_pregrow auto mode)")
args = ap.parse_args()
env = VecTrack2D(args.env, num_envs=args.n, seed=1, auto_reset=not args.no_auto_reset)
if args.pregrow:
    env.pregrow_auto(True)
out = (env.reset(), torch.empty((args.n, 2), device="cuda"), torch.empty((args.n,), dtype=torch.uint8, device="cuda"))
if args.no_obs:
    import ctypes as C
    class _Null(object):
        def data_ptr(self): return None
    _real = out
    def _sr(steps):
        from active_tracking_rl_amd.vec_env import _check
        _check(env.L.t2d_step_random(env.h, int(steps), 1, None, C.c_void_p(out[1].data_ptr()), C.c_void_p(out[2].data_ptr()), env._stream()))
    env.step_random = lambda steps, seed, o: _sr(steps)
if args.fused:
    T = 20      # one A3C rollout's worth of outputs per call
    h, w = env.obs_hw
    fout = (torch.empty((T, args.n, 2, h, w), device="cuda"), torch.empty((T, args.n, 2), device="cuda"),
            torch.empty((T, args.n), dtype=torch.uint8, device="cuda"))
    def _fused(steps, seed, o):
        for _ in range(steps // T):
            env.rollout_random(T, seed, fout)
    env.step_random = _fused
    args.steps -= args.steps % T
env.step_random(args.warmup, 1, out)
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.time()
ev0.record()
env.step_random(args.steps, 1, out)
ev1.record()
torch.cuda.synchronize()
dt = time.time() - t0
gpu_ms = ev0.elapsed_time(ev1)
print("%senv=%s N=%d auto_reset=%s steps=%d wall=%.3fs gpu=%.3fms  %.1f us/step  %.3e env-steps/s  (%.1f GB/s algorithmic)" % (
    "FUSED(<=10 steps/launch) " if args.fused else "", args.env, args.n, not args.no_auto_reset, args.steps, dt, gpu_ms, 1e3 * gpu_ms / args.steps,
    args.n * args.steps / (gpu_ms * 1e-3), 1723.0 * args.n * args.steps / (gpu_ms * 1e-3) / 1e9))
