"""create_env and the env-side protocol objects — the drop-in seam of environment.py:11-32,128-175.

Two adapters sit on the same C ABI (vec_env.VecTrack2D -> libtrack2d_hip.so):

  VecEnv       the batched protocol the MI355X rollout driver uses: device tensors in, device tensors out,
               `observation_space` / `action_space` are per-agent LISTS exactly like the reference's
               (len(env.observation_space) is how Agent learns num_agents, player_util.py:13).
  Track2DEnv   a single env that speaks the reference's gym protocol verbatim (numpy float32 obs
               [A, stack, 1, 13, 13], float64 rewards [A], bool done, info['distance']) so reference-style
               loops (gym_eval.py:97-121, random_agent_multi.py:17-53) run unchanged on top of the HIP path.

frame_stack (environment.py:128-156) is folded in: obs are float32, `stack_frames` most recent frames per
agent, the deque filled with the first frame on reset. Rescale (--rescale, --inv) is folded in too; listspace
(--single) and UnrealPreprocess belong to the single-agent / image envs.
"""
import os

import numpy as np
import torch

from . import registry
from .vec_env import VecTrack2D


class Discrete(object):
    """gym.spaces.Discrete stand-in (gym is not a dependency of this package)."""

    def __init__(self, n):
        self.n = n

    def sample(self):
        return int(np.random.randint(self.n))


class Box(object):
    """gym.spaces.Box stand-in: Box(low=0, high=6, shape=(1,13,13)) — track_1v1.py:257-259."""

    def __init__(self, low, high, shape, dtype=np.float32):
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype


def _spaces(obs_hw=(13, 13), num_actions=4):
    obs = [Box(0, 6, (1,) + tuple(obs_hw), np.float32) for _ in range(2)]
    act = [Discrete(num_actions) for _ in range(2)]       # define_action, track_1v1.py:242-249: 4, or 8 with 'Moore'
    return obs, act


class VecEnv(object):
    """Batched env: reset() -> obs [N, A, stack, 1, 13, 13] f32 (device);
    step([a_tracker [N], a_target [N]]) -> (obs, rewards [N, A] f32, done [N] uint8, info)."""

    def __init__(self, env_id, num_envs, device="cuda:0", seed=1, stack_frames=1, env_id_base=0, auto_reset=True,
                 rescale=False, obs_u8=False, async_gen=None, inv=False, **overrides):
        self.env_id = env_id
        self.num_envs = num_envs
        self.stack_frames = int(stack_frames)
        self.rescale = bool(rescale)    # environment.Rescale (environment.py:35-79): [0,255] -> [-1,1]
        # --inv (environment.py:49,62-76): per EPISODE a coin flip (choose_rand_seed); an inverted episode's reset observation
        # is -ob, its step observations are mx_d - ob = 255 - ob (of the RESCALED image — the reference's own arithmetic, kept)
        self.inv = bool(inv) and self.rescale
        self._inv_flags = None
        # t2d_generator_async (map generation forked onto the library's stream, under the following steps) is opt-in:
        # inside a captured hipGraph every fork/join of a side branch costs ~200 us on this ROCm (measured,
        # profiles/r02_async_generator_ab.txt), more than the generator launch it hides
        if async_gen is None:
            async_gen = os.environ.get("T2D_ASYNC_GEN", "0") == "1"
        self.core = VecTrack2D(env_id, num_envs=num_envs, device=device, seed=seed, env_id_base=env_id_base,
                               auto_reset=auto_reset, async_gen=async_gen, **overrides)
        self.observation_space, self.action_space = _spaces(self.core.obs_hw, self.core.num_actions)
        # obs_u8: observations stay bytes between the step kernel and the policy's conv stem (t2d_step_u8 ->
        # atr_stem_*_u8), i.e. frame_stack's np.float32 cast (environment.py:138,146) is fused into conv1. Only where
        # the kernels exist ('Partial' ids without Nav/RPF targets) and no host-side frame processing is asked for.
        # (an odd env count leaves every other slot of the [T+1, N, 2, 13, 13] byte rollout store 2-byte aligned — the step
        # kernel and the rollout prologue move dwords — so odd batches keep float observations)
        self.obs_u8 = (bool(obs_u8) and self.stack_frames == 1 and not self.rescale and self.core.supports_u8
                       and self.num_envs % 2 == 0)
        if self.rescale:
            for box in self.observation_space:
                box.low, box.high = -1.0, 1.0
        self.device = self.core.device
        self._frames = None
        self._seed = seed

    def seed(self, seed=None):
        """Accepted for protocol compatibility. In the reference env.seed() has no effect on trajectories
        (track_1v1.py:129-132 sets an RNG that is never read); here the Philox key is fixed at construction."""
        return [seed]

    def _stack(self, obs, done=None, fill=False):
        # obs [N, A, 13, 13] -> [N, A, stack, 1, 13, 13]
        if self.rescale:   # Rescale.rescale, same float32 operation order: ((clip(x) - 0) * 2) / 255 + (-1)
            obs = obs.clamp(0.0, 255.0).mul(2.0).div(255.0).add(-1.0)
            if self.inv:
                n = obs.shape[0]
                if fill or self._inv_flags is None:      # reset(): every env draws its flag and returns -ob
                    self._inv_flags = torch.rand(n, device=obs.device) < 0.5
                    fresh = torch.ones(n, dtype=torch.bool, device=obs.device)
                else:                                    # step(): auto-reset envs start a new episode (new flag, -ob)
                    fresh = done.bool() if done is not None else torch.zeros(n, dtype=torch.bool, device=obs.device)
                    self._inv_flags = torch.where(fresh, torch.rand(n, device=obs.device) < 0.5, self._inv_flags)
                shp = (n,) + (1,) * (obs.dim() - 1)
                f, r = self._inv_flags.view(shp), fresh.view(shp)
                obs = torch.where(f & r, -obs, torch.where(f & ~r, 255.0 - obs, obs))
        cur = obs.unsqueeze(2).unsqueeze(3)
        if self.stack_frames == 1:
            return cur
        if fill or self._frames is None:
            self._frames = cur.expand(-1, -1, self.stack_frames, -1, -1, -1).clone()
        else:
            nxt = torch.cat([self._frames[:, :, 1:], cur], dim=2)
            if done is not None:  # auto-reset envs restart their stack from the first frame of the new episode
                d = done.bool().view(-1, 1, 1, 1, 1, 1)
                nxt = torch.where(d, cur.expand_as(nxt), nxt)
            self._frames = nxt
        return self._frames

    def reset(self):
        obs = self.core.reset()
        if self.obs_u8:
            obs = obs.to(torch.uint8)        # values 0/1/2/4: exact
        return self._stack(obs, fill=True)

    def step(self, actions, out=None):
        """out: optional (obs [N,A,h,w] f32 (u8 with obs_u8), rew [N,A] f32, done [N] u8) device tensors the kernel
        writes directly (a slot of rollout_buffers)."""
        a0 = actions[0]
        a1 = actions[1] if len(actions) > 1 else None
        obs, rew, done = (self.core.step_u8 if self.obs_u8 else self.core.step)(a0, a1, out=out)
        return self._stack(obs, done), rew, done, {}

    def fused_step_out(self, out):
        """(core, obs, rew, done) for fused.act_env_step — the env step inside the policy step's last launch — when this
        env can take it ('Partial' observations, no Nav/RPF target, no host-side frame processing), else None."""
        if not self.core.supports_u8 or self.stack_frames != 1 or self.rescale:
            return None
        obs, rew, done = out
        if obs.dtype != (torch.uint8 if self.obs_u8 else torch.float32):
            return None
        return (self.core, obs, rew, done)

    def after_fused_step(self, env_out):
        """What step() returns, for a step that fused.act_env_step already ran into env_out's buffers."""
        _, obs, rew, done = env_out
        return self._stack(obs, done), rew, done, {}

    def rollout_buffers(self, num_steps):
        """Storage for one rollout the step kernel writes in place — obs [T+1,N,A,h,w] (slot 0 = the state the
        rollout starts from), rewards [T,N,A], done [T,N] — so the learner reads the rollout without a stacking
        copy (110 MB per 20-step, 4096-env rollout). None when frames are stacked host-side."""
        if self.stack_frames != 1 or self.rescale:
            return None
        h, w = self.core.obs_hw
        dev = self.device
        return (torch.empty((num_steps + 1, self.num_envs, 2, h, w), device=dev,
                            dtype=torch.uint8 if self.obs_u8 else torch.float32),
                torch.empty((num_steps, self.num_envs, 2), dtype=torch.float32, device=dev),
                torch.empty((num_steps, self.num_envs), dtype=torch.uint8, device=dev))

    def flush(self):
        self.core.flush()

    def generator_join(self):
        self.core.generator_join()

    def pregrow(self, fork=True):
        """Maze maps of the coming episodes grown ahead of the generator pass (VecTrack2D.pregrow); no-op without Maze envs."""
        self.core.pregrow(fork)

    @property
    def generator_cycle(self):
        return self.core.generator_cycle

    def core_max_steps(self):
        """Upper bound on the episode length (TimeLimit, gym_track2d/__init__.py:17)."""
        return registry.MAX_EPISODE_STEPS

    def close(self):
        self.core.close()

    def render(self, *a, **k):
        raise NotImplementedError("matplotlib rendering (track_1v1.py:170-216) is out of scope")


class NumpyVecEnv(object):
    """N envs in the REFERENCE-EXACT mode at once (VecEnv's batched protocol, rng="numpy"): env i's maps, spawns, goals and
    scripted Ram / Nav / RPF target come from its own numpy-legacy stream seeded np.random.seed(seeds[i]) — host C++
    (np_mode.NpBatchSource: MT19937, numpy's draws, heapq-faithful A*, spread over host threads) — and go to the device through
    t2d_inject / as the step's target action; the device does the per-step work for all N in one launch. An env that finishes
    is reset at once from ITS stream (the next episode of that seed, as the reference's worker loop does: train.py:73-74) and
    reports the new episode's first observation, like VecEnv. Per env the episodes are the reference's from the seed alone.
    env_ids: one id or one per env (same observation type; map type, target mode and level may differ per env).
    Costs a host round trip per step (done flags down, target actions / new episodes up): a parity tool, not the
    throughput path."""

    def __init__(self, env_ids, seeds, device="cuda:0", threads=0, device_generators=False):
        """device_generators: the numpy-legacy streams live ON THE DEVICE (np_mode.attach_device_streams -> csrc k_gen_np: MT19937,
        numpy's doubles / bounded integers / whole Fisher-Yates permutations, the reference's map generators and samplers, one
        wavefront per env) and finished envs restart inside the step launch from episodes pre-generated out of their own stream:
        no host round trip, same episodes. For ids whose target draws nothing itself (Adv, PZR, Far) — and, since round 6, the
        scripted Ram target, whose draws interleave with the resets: such a handle runs without the in-launch auto-reset, step()
        restarts finished envs with a masked reset that draws their next episode at that moment, and RamAgent.step() itself runs
        on the device from the env's stream ahead of every step (csrc k_ram_np) — and the Nav target: Navigator.reset / .step with
        the reference's heap A* (heapq's sift order, list-comparison ties, the inverted replace test) restated on the device,
        one lane per search — and the RPF patrol (the same Navigator planning on the generator's map, whose four patrol cells are
        free, while the env's own map may keep them as walls: track_1v1.py:233-236)."""
        from .np_mode import NpBatchSource
        n = len(seeds)
        ids = [env_ids] * n if isinstance(env_ids, str) else list(env_ids)
        assert len(ids) == n
        sp = [registry.spec(i) for i in ids]
        assert len(set(x["obs_type"] for x in sp)) == 1, "one observation type per handle"
        self.num_envs, self.env_ids = n, ids
        self.device_generators = bool(device_generators)
        self._interleaved = False
        if self.device_generators:
            from .np_mode import attach_device_streams
            self.src = None
            # a scripted target's draws interleave with the resets: no episode can be made ahead of time, so no in-launch auto-reset
            self._interleaved = any(x["target_mode"] in ("Ram", "Nav", "RPF") for x in sp)
            self.core = VecTrack2D(ids[0], num_envs=n, device=device, seed=int(seeds[0]), auto_reset=not self._interleaved,
                                   map_type_per_env=np.array([registry.MAP_CODE[x["map_type"]] for x in sp], np.uint8),
                                   target_mode_per_env=np.array([registry.TARGET_CODE[x["target_mode"]] for x in sp], np.uint8),
                                   level_per_env=np.array([x["level"] for x in sp], np.uint8), obs_type=sp[0]["obs_type"])
            attach_device_streams(self.core, seeds)
            self.device = self.core.device
            self.observation_space, self.action_space = _spaces(self.core.obs_hw, self.core.num_actions)
            self._scripted_idx = np.zeros(0, np.int64)
            return
        self.src = NpBatchSource([x["map_type"] for x in sp], [x["target_mode"] for x in sp], [x["level"] for x in sp], seeds,
                                 threads)
        modes = np.array([registry.TARGET_CODE["Ext"] if self.src.scripted[i] else registry.TARGET_CODE[sp[i]["target_mode"]]
                          for i in range(n)], np.uint8)           # scripted targets are driven from the host (T2D_TGT_EXT)
        self.core = VecTrack2D(ids[0], num_envs=n, device=device, seed=int(seeds[0]), auto_reset=False,
                               map_type_per_env=np.array([registry.MAP_CODE[x["map_type"]] for x in sp], np.uint8),
                               target_mode_per_env=modes, level_per_env=np.array([x["level"] for x in sp], np.uint8),
                               obs_type=sp[0]["obs_type"])
        self.device = self.core.device
        self.observation_space, self.action_space = _spaces(self.core.obs_hw, self.core.num_actions)
        self._scripted_idx = np.nonzero(self.src.scripted)[0]

    def _inject(self, idx, mazes, sides, pos, goals):
        for j, i in enumerate(idx):                    # (per env: sides differ between Maze and Block / Empty maps)
            s_ = int(sides[j])
            self.core.inject(np.ascontiguousarray(mazes[j, :s_, :s_]), pos[j], goals[j], first=int(i))

    def reset(self, mask=None):
        """New episodes for every env (or those where mask is set) from their streams; returns all observations."""
        if self.device_generators:             # the next episode of every (masked) env's own stream, generated on the device
            m = None if mask is None else torch.as_tensor(np.asarray(mask), dtype=torch.uint8, device=self.device)
            return self.core.reset(m)
        idx = np.arange(self.num_envs) if mask is None else np.nonzero(np.asarray(mask))[0]
        self._inject(idx, *self.src.reset(idx))
        return self.core.observe()

    def step(self, actions):
        """actions: [a_tracker [N], a_target [N]] (device or host integers; the target's entry is ignored where the target is
        scripted) -> (obs [N,2,h,w] f32, rewards [N,2] f32, done [N] u8, info {'distance': [N] f64})."""
        dev = self.device
        a0 = torch.as_tensor(actions[0], dtype=torch.int64, device=dev).reshape(-1).contiguous()
        a1 = torch.as_tensor(actions[1], dtype=torch.int64, device=dev).reshape(-1).clone()
        if len(self._scripted_idx):                    # track_1v1.py:80-84: the scripted target overrides action[1]
            ta = self.src.target_actions(self._scripted_idx)
            a1[torch.as_tensor(self._scripted_idx, device=dev)] = torch.as_tensor(ta, dtype=torch.int64, device=dev)
        obs, rew, done = self.core.step(a0, a1.contiguous())
        if self.device_generators:
            d2 = self.core.get_state()["d2"].astype(np.float64)
            if self._interleaved:              # Ram targets: finished envs restart NOW, from where their stream stands
                obs = self.core.reset(done, out=obs)       # (fresh first observations for them, everybody else's as they were)
            else:                              # restarted inside the launch: d2 is the new episode's, the step's own was parked
                fin = done.cpu().numpy() != 0
                if fin.any():
                    d2[fin] = self.core.terminal_d2().astype(np.float64)[fin]
            return obs, rew, done, {"distance": np.sqrt(d2)}
        d2 = self.core.get_state()["d2"].astype(np.float64)
        fin = np.nonzero(done.cpu().numpy())[0]
        if len(fin):
            self._inject(fin, *self.src.reset(fin))
            fresh = self.core.observe()
            sel = torch.as_tensor(fin, device=dev)
            obs[sel] = fresh[sel]
        return obs, rew, done, {"distance": np.sqrt(d2)}

    def close(self):
        self.core.close()
        if self.src is not None:
            self.src.close()


class Track2DEnv(object):
    """One env behind the reference's exact gym protocol (TimeLimit + frame_stack included).

    rng="philox" (default): maps / spawns / scripted targets come from the device generators (Philox streams).
    rng="numpy": the reference-exact mode — np.random.seed(seed) semantics: maps, spawns, goals and the scripted
    Ram / Nav / RPF targets are produced on the host by the numpy-legacy stream + heapq-faithful A* of
    include/track2d_np.h (np_mode.NpEpisodeSource) in the reference's draw order and injected (t2d_inject; the target's
    action as the step's target action), so an episode is the reference's, bit for bit, from the seed alone. The
    argument-less np.random.seed() calls inside generators.py:41,56 are NOT replayed (they make the reference itself
    irreproducible; the golden vectors were captured with them neutralised)."""

    def __init__(self, env_id, device="cuda:0", seed=1, stack_frames=1, rescale=False, rng="philox", inv=False):
        if rng not in ("philox", "numpy"):
            raise ValueError("rng must be 'philox' or 'numpy'")
        self.rng = rng
        self._np = None
        over = {}
        if rng == "numpy":
            from .np_mode import NpEpisodeSource
            sp = registry.spec(env_id)
            self._np = NpEpisodeSource(sp["map_type"], sp["target_mode"], sp["level"], seed)
            if self._np.scripted:      # the host drives the target; rewards use w_p = 0 either way (track_1v1.py:147-152)
                over = dict(target_mode_per_env=np.array([registry.TARGET_CODE["Ext"]], np.uint8))
        self.vec = VecEnv(env_id, 1, device=device, seed=seed, stack_frames=stack_frames, auto_reset=False,
                          rescale=rescale, inv=inv, **over)
        self.observation_space, self.action_space = self.vec.observation_space, self.vec.action_space

    def seed(self, seed=None):
        """rng="numpy": np.random.seed(seed) on the env's stream (what actually determines the reference's episodes;
        the reference's own env.seed() is a no-op for them, track_1v1.py:129-132)."""
        if self._np is not None and seed is not None:
            self._np.seed(seed)
        return self.vec.seed(seed)

    def reset(self):
        if self._np is None:
            return self.vec.reset()[0].cpu().numpy()
        maze, pos, goals = self._np.reset()
        core = self.vec.core
        core.inject(maze, pos, goals)                      # also zeroes the step / far counters (Track1v1Env.reset)
        return self.vec._stack(core.observe(), fill=True)[0].cpu().numpy()

    def step(self, action):
        a = [int(np.asarray(x).reshape(-1)[0]) for x in list(action)[:2]]
        if len(a) == 1:
            a.append(0)
        if self._np is not None and self._np.scripted:
            a[1] = self._np.target_action()                # track_1v1.py:80-84: the scripted target overrides action[1]
        a = [torch.tensor([x], dtype=torch.int64, device=self.vec.device) for x in a]
        obs, rew, done, _ = self.vec.step(a)
        d2 = int(self.vec.core.get_state()["d2"][0])
        info = {"distance": float(np.sqrt(float(d2)))}           # track_1v1.py:118
        return obs[0].cpu().numpy(), rew[0].double().cpu().numpy(), bool(done[0].item()), info

    def close(self):
        self.vec.close()
        if self._np is not None:
            self._np.close()

    def render(self, *a, **k):
        return self.vec.render()


def create_env(env_id, args, num_envs=None, device=None, env_id_base=0, obs_u8=None, rng=None):
    """environment.create_env (environment.py:11-32) for the Track2D ids.

    args carries the reference's flags (stack_frames, seed, ...) plus optionally `num_envs` and `gpu_ids`.
    Returns a VecEnv when num_envs (argument or args.num_envs) > 1, else a gym-protocol Track2DEnv."""
    if '2D' not in env_id:
        raise NotImplementedError("only the Track2D-* ids are in scope (Unreal envs need UE4 binaries)")
    registry.spec(env_id)
    if getattr(args, "single", False):
        # listspace (environment.py:159-175) exists for single-agent image envs. On a Track2D id the reference itself fails
        # with --single: listspace.reset wraps the two-agent observation [2,1,13,13] in a one-element list, frame_stack then
        # sees ONE agent whose frame is [2,1,13,13], and CNN_maze's conv receives a 5-D tensor.
        raise NotImplementedError("--single (listspace) wraps single-agent image envs; Track2D envs have two agents "
                                  "(the reference fails on this combination too)")
    rescale = bool(getattr(args, "rescale", False))
    inv = bool(getattr(args, "inv", False)) and rescale
    n = num_envs if num_envs is not None else getattr(args, "num_envs", 1)
    if device is None:
        gpu_ids = getattr(args, "gpu_ids", [0])
        gid = gpu_ids[0] if isinstance(gpu_ids, (list, tuple)) else gpu_ids
        device = "cuda:%d" % max(int(gid), 0)
    stack = getattr(args, "stack_frames", 1)
    seed = getattr(args, "seed", 1)
    if obs_u8 is None:
        obs_u8 = bool(getattr(args, "obs_u8", False))
    rng = rng if rng is not None else getattr(args, "rng", "philox")
    if n > 1 and rng in ("numpy", "numpy-device"):      # the reference-exact mode for a batch: env i on np.random.seed(seed + env_id_base + i)
        if stack != 1 or rescale:
            raise NotImplementedError("rng='numpy' with num_envs > 1 returns raw float32 observations (no frame stack / rescale)")
        # rng="numpy-device" / args.np_device: the streams on the device (t2d_np_attach) where the target mode allows it
        on_dev = rng == "numpy-device" or bool(getattr(args, "np_device", False))
        return NumpyVecEnv(env_id, [int(seed) + int(env_id_base) + i for i in range(n)], device=device, device_generators=on_dev)
    if n > 1:
        return VecEnv(env_id, n, device=device, seed=seed, stack_frames=stack, env_id_base=env_id_base, rescale=rescale,
                      obs_u8=obs_u8, inv=inv)
    return Track2DEnv(env_id, device=device, seed=seed, stack_frames=stack, rescale=rescale, rng=rng, inv=inv)
