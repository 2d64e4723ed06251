"""Agent.loss (active_tracking_rl_amd/player_util.py) vs the golden outputs of the REFERENCE Agent.optimize
(player_util.py:108-161) on synthetic rollout buffers (tests/golden/loss.npz): n-step return, GAE, entropy and
aux-reward terms, for done / not-done rollouts and training modes -1 / 0 / 1. Tolerance 1e-5 relative (fp32)."""
import os

import numpy as np
import torch

from conftest import GOLDEN
from active_tracking_rl_amd.environment import _spaces
from active_tracking_rl_amd.player_util import Agent
from active_tracking_rl_amd.train import default_args


class _FakeVecEnv(object):
    num_envs = 1

    def __init__(self):
        self.observation_space, self.action_space = _spaces()


def _agent(aux, n=1):
    env = _FakeVecEnv()
    env.num_envs = n
    args = default_args(aux=aux, num_envs=n)
    ag = Agent(None, env, args, None, torch.device("cpu"))
    ag.w_entropy_target = 0.2
    return ag


def _fill(ag, g, p, leaf, n=1, done_rows=None):
    T = g[p + "values"].shape[0]
    rep = lambda a: torch.from_numpy(a).unsqueeze(0).expand(n, *a.shape).clone()
    ag.values = [rep(v) * leaf for v in g[p + "values"]]
    ag.log_probs = [rep(v) * leaf for v in g[p + "log_probs"]]
    ag.entropies = [rep(v) * leaf for v in g[p + "entropies"]]
    ag.rewards = [rep(v) for v in g[p + "rewards"]]
    ag.preds = [rep(v[0]) * leaf for v in g[p + "preds"]]
    ag.dones = [torch.zeros(n, dtype=torch.uint8) for _ in range(T)]
    if bool(g[p + "done"]):
        ag.dones[-1] = torch.ones(n, dtype=torch.uint8)
    boot = rep(g[p + "boot"])
    ag.model = lambda inp: (boot.clone(), None, None, None, None, None)
    ag.state, ag.hxs, ag.cxs = None, None, None


def test_loss_matches_reference_optimize():
    g = np.load(os.path.join(GOLDEN, "loss.npz"))
    for c in range(int(g["count"])):
        p = "c%d/" % c
        ag = _agent(str(g[p + "aux"]))
        leaf = torch.ones(1, requires_grad=True)
        _fill(ag, g, p, leaf)
        loss, pl, vl, en, pr = ag.loss(int(g[p + "mode"]))
        tol = dict(rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(pl[0].detach().numpy(), g[p + "policy_loss"], **tol)
        np.testing.assert_allclose(vl[0].detach().numpy(), g[p + "value_loss"], **tol)
        np.testing.assert_allclose(en[0].detach().numpy(), g[p + "entropy_sum"], **tol)
        np.testing.assert_allclose(pr[0].detach().numpy().reshape(-1), g[p + "pred_loss"].reshape(-1), **tol)
        loss.backward()
        np.testing.assert_allclose(leaf.grad.numpy(), g[p + "dloss_dleaf"], rtol=2e-5, atol=2e-5)


def test_batched_loss_is_mean_of_per_env_losses_with_mid_rollout_boundaries():
    """An env that finishes in the MIDDLE of the 20-step window contributes exactly the sum of the reference losses
    of its two segments (first: done rollout; second: bootstrapped rollout)."""
    g = np.load(os.path.join(GOLDEN, "loss.npz"))
    rs = np.random.RandomState(0)
    T, cut = 9, 4
    mk = lambda *s: rs.randn(*s).astype(np.float32)
    vals, lps, ents = mk(T, 2, 1), -np.abs(mk(T, 2, 1)), np.abs(mk(T, 2, 1))
    rews, preds, boot = rs.uniform(-1, 1, (T, 2, 1)).astype(np.float32), mk(T, 1), mk(2, 1)

    def run(sl, done_last, bootv):
        ag = _agent("reward")
        ag.values = [torch.from_numpy(v)[None] for v in vals[sl]]
        ag.log_probs = [torch.from_numpy(v)[None] for v in lps[sl]]
        ag.entropies = [torch.from_numpy(v)[None] for v in ents[sl]]
        ag.rewards = [torch.from_numpy(v)[None] for v in rews[sl]]
        ag.preds = [torch.from_numpy(v)[None] for v in preds[sl]]
        n = len(ag.values)
        ag.dones = [torch.zeros(1, dtype=torch.uint8) for _ in range(n)]
        if done_last is not None:
            for i in done_last:
                ag.dones[i] = torch.ones(1, dtype=torch.uint8)
        b = torch.from_numpy(bootv)[None]
        ag.model = lambda inp: (b.clone(), None, None, None, None, None)
        return ag.loss(-1)

    whole = run(slice(0, T), [cut - 1], boot)
    first = run(slice(0, cut), [cut - 1], np.zeros_like(boot))
    second = run(slice(cut, T), None, boot)
    for k in range(1, 5):
        np.testing.assert_allclose(whole[k].numpy(), (first[k] + second[k]).numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(float(whole[0]), float(first[0] + second[0]), rtol=1e-5)


def test_recompute_path_equals_per_step_autograd_path():
    """Actor/learner split (action_rollout + loss_recompute, time-batched) vs the reference-shaped per-step path
    (action_train + loss): same loss, same gradients, on the same env stream and the same sampled actions."""
    from test_distributed_cpu import FakeVecEnv
    from active_tracking_rl_amd.model import build_model
    from active_tracking_rl_amd.train import rollout
    saved = torch.Tensor.multinomial
    torch.Tensor.multinomial = lambda self, n, *a, **k: self.argmax(1, keepdim=True)
    try:
        for net, aux in (("tat-maze-lstm", "reward"), ("maze-lstm", "none")):
            outs = []
            for fast in (False, True):
                args = default_args(network=net, aux=aux, num_envs=5, num_steps=7)
                torch.manual_seed(4)
                env = FakeVecEnv(range(5))
                model = build_model(env.observation_space, env.action_space, args, torch.device("cpu"))
                ag = Agent(model, env, args, None, torch.device("cpu"))
                ag.reset()
                for it in range(2):                       # second rollout starts from a non-zero LSTM state
                    rollout(ag, args.num_steps, fast=fast)
                    loss, pl, vl, en, pr = (ag.loss_recompute if fast else ag.loss)(-1)
                    model.zero_grad()
                    loss.backward()
                    grads = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
                    outs.append((float(loss), pl.detach(), vl.detach(), en.detach(), pr.detach(), grads.clone()))
                    ag.clear_actions()
                    model.cache_dense(False)
            for a, b in zip(outs[:2], outs[2:]):
                assert abs(a[0] - b[0]) <= 1e-5 * max(1.0, abs(a[0]))
                for x, y in zip(a[1:5], b[1:5]):
                    np.testing.assert_allclose(x.numpy(), y.numpy(), rtol=1e-4, atol=1e-5)
                np.testing.assert_allclose(a[5].numpy(), b[5].numpy(), rtol=1e-3, atol=2e-6)
    finally:
        torch.Tensor.multinomial = saved
