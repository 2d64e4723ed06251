"""N>1 path on CPU: world_size-2 gloo run of the data-parallel update (Agent.allreduce_grads + SharedAdam on the
flat bucket). Checks that (a) the all-reduced gradient is the mean of the per-rank gradients, (b) replicas stay
bit-identical after updates although every rank sees different envs, (c) the sharded update equals the single-
process update over the union of the shards."""
import os
import socket
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from active_tracking_rl_amd.environment import _spaces
from active_tracking_rl_amd.model import build_model
from active_tracking_rl_amd.player_util import Agent
from active_tracking_rl_amd.shared_optim import SharedAdam
from active_tracking_rl_amd.train import default_args, rollout, select_params


class FakeVecEnv(object):
    """Action-independent synthetic env on CPU: obs/reward/done streams are a function of (global env id, t)."""

    def __init__(self, ids):
        self.ids = list(ids)
        self.num_envs = len(self.ids)
        self.observation_space, self.action_space = _spaces()
        self.t = 0

    def _gen(self):
        obs, rew, done = [], [], []
        for g in self.ids:
            rs = np.random.RandomState(1000 * g + self.t)
            obs.append(rs.choice([0, 1, 2, 4], size=(2, 1, 1, 13, 13)).astype(np.float32))
            rew.append(rs.uniform(-1, 1, 2).astype(np.float32))
            done.append(1 if rs.rand() < 0.1 else 0)
        return (torch.from_numpy(np.stack(obs)), torch.from_numpy(np.stack(rew)),
                torch.tensor(done, dtype=torch.uint8))

    def reset(self):
        self.t = 0
        return self._gen()[0]

    def step(self, actions):
        self.t += 1
        o, r, d = self._gen()
        return o, r, d, {}


def _player(ids, seed=3):
    args = default_args(num_envs=len(ids), num_steps=5)
    dev = torch.device("cpu")
    torch.manual_seed(seed)
    env = FakeVecEnv(ids)
    model = build_model(env.observation_space, env.action_space, args, dev)
    opt = SharedAdam(select_params(model, args.train_mode), lr=args.lr)
    player = Agent(model, env, args, None, dev)
    player.reset()
    return player, opt, args


def _deterministic_sampling():
    # make action sampling a pure function of the logits so that sharded and unsharded runs agree
    torch.Tensor.multinomial = lambda self, n, *a, **k: self.argmax(1, keepdim=True)


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _deterministic_sampling()
    torch.set_num_threads(1)
    ids = [rank * 3 + i for i in range(3)]          # env_id_base = rank * num_envs
    player, opt, args = _player(ids)
    for it in range(2):
        rollout(player, args.num_steps)
        player.compute_grads(opt, args.train_mode)
        local = opt.bucket.grad.clone()
        gathered = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        player.allreduce_grads(opt)
        assert torch.allclose(opt.bucket.grad, sum(gathered) / world, rtol=1e-6, atol=1e-8)
        for r in range(1, world):
            assert not torch.equal(gathered[0], gathered[r])       # ranks really saw different envs
        opt.step()
        flats = [torch.zeros_like(opt.bucket.flat) for _ in range(world)]
        dist.all_gather(flats, opt.bucket.flat)
        for r in range(1, world):
            assert torch.equal(flats[0], flats[r]), "replicas diverged (rank %d)" % r
    if rank == 0:
        torch.save(opt.bucket.flat, os.path.join(out_dir, "flat.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 4])
def test_two_rank_gloo_update_matches_single_process(world):
    """world = 4 (round 6): env_id_base for ranks 2 and 3, a four-way mean, four replicas bit-identical after every update."""
    out_dir = tempfile.mkdtemp()
    mp.spawn(_worker, args=(world, _free_port(), out_dir), nprocs=world, join=True)
    sharded = torch.load(os.path.join(out_dir, "flat.pt"))
    saved = torch.Tensor.multinomial
    try:
        _deterministic_sampling()
        player, opt, args = _player(list(range(3 * world)))
        for it in range(2):
            rollout(player, args.num_steps)
            player.optimize(None, opt, player.model, args.train_mode, torch.device("cpu"))
    finally:
        torch.Tensor.multinomial = saved
    np.testing.assert_allclose(opt.bucket.flat.numpy(), sharded.numpy(), rtol=2e-4, atol=2e-6)


def test_flat_bucket_views_and_sharedadam_numerics():
    """SharedAdam over the flat bucket == the reference update rule (shared_optim.py:149-173) in float64."""
    torch.manual_seed(0)
    lin = torch.nn.Linear(7, 5)
    opt = SharedAdam(lin.parameters(), lr=1e-3)
    assert lin.weight.data_ptr() == opt.bucket.flat.data_ptr()           # params are views of the bucket
    p = opt.bucket.flat.double().clone()
    m = torch.zeros_like(p); v = torch.zeros_like(p); vmax = torch.zeros_like(p)
    for t in range(1, 6):
        opt.zero_grad()
        (lin(torch.randn(4, 7)) ** 2).sum().backward()
        assert lin.weight.grad.data_ptr() == opt.bucket.grad.data_ptr()   # grads accumulate into the bucket
        g = opt.bucket.grad.double().clone()
        opt.step()
        m = m * 0.9 + 0.1 * g; v = v * 0.999 + 0.001 * g * g; vmax = torch.maximum(vmax, v)
        step_size = 1e-3 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        p = p - step_size * m / (vmax.sqrt() + 1e-3)
        np.testing.assert_allclose(opt.bucket.flat.numpy(), p.float().numpy(), rtol=1e-5, atol=1e-7)


def test_rmsprop_and_per_worker_optimizer_numerics():
    """The other three optimizers main.py / train.py of the reference can arrive at (shared_optim.make_optimizer):
    SharedRMSprop (shared_optim.py:43-87) against its update rule in float64; the per-worker torch.optim.Adam / RMSprop
    (train.py:45-49) against torch.optim itself on a twin module."""
    import argparse
    from active_tracking_rl_amd.shared_optim import SharedRMSprop, make_optimizer
    torch.manual_seed(1)
    lin = torch.nn.Linear(7, 5)
    opt = make_optimizer(lin.parameters(), argparse.Namespace(optimizer="RMSprop", shared_optimizer=True, lr=7e-4, amsgrad=True))
    assert isinstance(opt, SharedRMSprop) and opt.param_groups[0]["eps"] == 0.1
    p = opt.bucket.flat.double().clone()
    v = torch.zeros_like(p)
    for t in range(5):
        opt.zero_grad()
        (lin(torch.randn(4, 7)) ** 2).sum().backward()
        g = opt.bucket.grad.double().clone()
        opt.step()
        v = v * 0.99 + 0.01 * g * g
        p = p - 7e-4 * g / (v.sqrt() + 0.1)
        np.testing.assert_allclose(opt.bucket.flat.numpy(), p.float().numpy(), rtol=1e-5, atol=1e-7)
    for name, ref_cls in (("Adam", torch.optim.Adam), ("RMSprop", torch.optim.RMSprop)):
        torch.manual_seed(2)
        a, b = torch.nn.Linear(6, 3), torch.nn.Linear(6, 3)
        b.load_state_dict(a.state_dict())
        oa = make_optimizer(a.parameters(), argparse.Namespace(optimizer=name, shared_optimizer=False, lr=1e-3, amsgrad=True))
        ob = ref_cls(b.parameters(), lr=1e-3)
        for t in range(6):
            x = torch.randn(5, 6)
            oa.zero_grad(); ob.zero_grad()
            (a(x) ** 2).sum().backward(); (b(x) ** 2).sum().backward()
            oa.step(); ob.step()
            for pa, pb in zip(a.parameters(), b.parameters()):
                np.testing.assert_allclose(pa.detach().numpy(), pb.detach().numpy(), rtol=2e-5, atol=1e-7)
    with pytest.raises(ValueError):
        make_optimizer(lin.parameters(), argparse.Namespace(optimizer="SGD", shared_optimizer=True, lr=1e-3, amsgrad=True))
