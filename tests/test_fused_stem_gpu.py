"""-m gpu numerics test of the fused HIP conv stem (csrc/stem_hip.hip) against the plain PyTorch fp32 reference of
the same op (F.conv2d + ReLU, autograd for the gradients). Tolerances: forward 1e-5 abs/rel; parameter gradients
2e-4 relative to the gradient's max (fp32 sums over up to 10^5 frames in a different order)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref(x, w1, b1, w2, b2):
    a = F.relu(F.conv2d(x.view(-1, 1, 13, 13), w1, b1, stride=2, padding=1))
    return F.relu(F.conv2d(a, w2, b2, stride=2, padding=1)).reshape(x.shape[0], -1)


@pytest.mark.parametrize("M", [1, 3, 64, 1000, 12288])
def test_fused_stem_matches_conv2d(M):
    from active_tracking_rl_amd import fused
    torch.manual_seed(M)
    dev = "cuda"
    conv1 = torch.nn.Conv2d(1, 16, 3, 2, 1).to(dev)
    conv2 = torch.nn.Conv2d(16, 32, 3, 2, 1).to(dev)
    with torch.no_grad():
        conv1.weight.mul_(2.0); conv2.weight.mul_(3.0); conv1.bias.normal_(0, 0.2); conv2.bias.normal_(0, 0.2)
    x = torch.tensor(np.random.RandomState(M).choice([0, 1, 2, 4], size=(M, 169)).astype(np.float32), device=dev)
    y = fused.stem(x, conv1, conv2)
    yr = _ref(x, conv1.weight, conv1.bias, conv2.weight, conv2.bias)
    assert y.shape == (M, 512)
    torch.testing.assert_close(y, yr, rtol=1e-5, atol=1e-5)
    g = torch.randn_like(y)
    params = [conv1.weight, conv1.bias, conv2.weight, conv2.bias]
    got = torch.autograd.grad((y * g).sum(), params)
    want = torch.autograd.grad((yr * g).sum(), params)
    for a, b, name in zip(got, want, ("dw1", "db1", "dw2", "db2")):
        scale = float(b.abs().max()) + 1e-6
        assert float((a - b).abs().max()) <= 2e-4 * scale, (name, M, float((a - b).abs().max()), scale)


def test_model_with_fused_stem_equals_gemm_stem():
    from active_tracking_rl_amd.environment import _spaces
    from active_tracking_rl_amd.model import CNN_maze, build_model
    from active_tracking_rl_amd.train import default_args
    obs, act = _spaces()
    args = default_args()
    torch.manual_seed(0)
    m = build_model(obs, act, args, torch.device("cuda")).cuda()
    states = torch.randint(0, 5, (32, 2, 1, 1, 13, 13), device="cuda").float()
    hx = torch.zeros(32, 2, 128, device="cuda"); cx = torch.zeros_like(hx)
    outs = []
    for fused_on in (True, False):
        CNN_maze.use_fused = fused_on
        with torch.no_grad():
            v, a, e, lp, (h, c), rp = m((states, (hx, cx)), True)
        outs.append((v, e, lp, h, c, rp))
    CNN_maze.use_fused = True
    for p, q in zip(*outs):
        torch.testing.assert_close(p, q, rtol=1e-4, atol=1e-5)


def test_fused_action_sampler_draws_from_the_softmax_distribution():
    """atr_sample_actions: empirical action frequencies match softmax(W h + b) (chi-square), draws differ call to
    call (device-side counter) and across rows."""
    from active_tracking_rl_amd import fused
    torch.manual_seed(0)
    lin = torch.nn.Linear(128, 4).cuda()
    with torch.no_grad():
        lin.weight.mul_(8.0); lin.bias.normal_()
    hrow = torch.randn(1, 128, device="cuda")
    n = 200000
    h = hrow.expand(n, 128).contiguous()
    sampler = fused.ActionSampler(torch.device("cuda"), seed=123)
    a = sampler(h, lin)
    assert a.dtype == torch.int64 and int(a.min()) >= 0 and int(a.max()) <= 3
    p = torch.softmax(lin(hrow), 1)[0].detach().double().cpu().numpy()
    counts = np.bincount(a.cpu().numpy(), minlength=4).astype(np.float64)
    chi2 = (((counts - n * p) ** 2) / (n * p + 1e-9)).sum()
    assert chi2 < 30.0, (chi2, counts / n, p)            # dof 3: 30 is p ~ 1e-6
    b = sampler(h, lin)
    assert not torch.equal(a, b) and int(sampler.counter.item()) == 2
    # different rows, different distributions: per-row probability of the sampled action is plausible on average
    h2 = torch.randn(50000, 128, device="cuda")
    a2 = sampler(h2, lin)
    p2 = torch.softmax(lin(h2), 1).detach()
    mean_p = float(p2.gather(1, a2.unsqueeze(1)).mean())
    expect = float((p2 * p2).sum(1).mean())
    assert abs(mean_p - expect) < 0.01, (mean_p, expect)
