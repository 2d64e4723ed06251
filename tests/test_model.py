"""Batched policy restatement (active_tracking_rl_amd/model.py) vs
  * the golden outputs of the REFERENCE A3C_Dueling.forward(test=True) on deterministic weights/inputs
    (tests/golden/model.npz, made by make_golden.py::model_fixture) — tolerance 2e-5 abs (fp32 GEMM order);
  * a plain PyTorch fp32 F.conv2d evaluation of the same stem (numerics test of the Toeplitz-GEMM convs);
  * the state-dict contract (names, shapes, parameter counts) of SURVEY.md §8a row M."""
import os

import numpy as np
import torch

from conftest import GOLDEN
from active_tracking_rl_amd.environment import _spaces
from active_tracking_rl_amd.model import CNN_maze, build_model
from active_tracking_rl_amd.train import default_args


def det_weights(shape, k):
    n = int(np.prod(shape))
    fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else int(shape[0])
    w = np.sin(np.arange(n, dtype=np.float64) * 0.7391 + 0.1 * k) / np.sqrt(max(fan_in, 1))
    return w.astype(np.float32).reshape(shape)


def _model(net):
    obs, act = _spaces()
    args = default_args(network=net, aux="reward" if "tat" in net else "none")
    torch.manual_seed(0)
    return build_model(obs, act, args, torch.device("cpu")), args


def test_state_dict_contract_and_reference_outputs():
    g = np.load(os.path.join(GOLDEN, "model.npz"))
    for net, n_params in (("tat-maze-lstm", 801291), ("maze-lstm", 668810)):
        m, _ = _model(net)
        sd = m.state_dict()
        keys = sorted(sd.keys())
        assert keys == [str(k) for k in g[net + "/keys"]]
        assert [str(tuple(sd[k].shape)) for k in keys] == [str(s) for s in g[net + "/shapes"]]
        assert sum(v.numel() for v in sd.values()) == n_params == int(g[net + "/n_params"])
        for k, name in enumerate(keys):
            sd[name].copy_(torch.from_numpy(det_weights(tuple(sd[name].shape), k)))
        m.eval()
        states = torch.from_numpy(g[net + "/states"])
        hx, cx = torch.from_numpy(g[net + "/hx"]), torch.from_numpy(g[net + "/cx"])
        with torch.no_grad():
            v, a, e, lp, (h, c), rp = m((states, (hx, cx)), True)                 # batched layout, N = 6
        tol = dict(atol=2e-5, rtol=1e-5)
        np.testing.assert_allclose(v.numpy(), g[net + "/values"], **tol)
        np.testing.assert_allclose(e.numpy(), g[net + "/entropies"], **tol)
        np.testing.assert_allclose(lp.numpy(), g[net + "/log_probs"], **tol)
        np.testing.assert_allclose(h.numpy(), g[net + "/hx_out"], **tol)
        np.testing.assert_allclose(c.numpy(), g[net + "/cx_out"], **tol)
        assert np.array_equal(torch.stack(a, 1).numpy(), g[net + "/actions"])
        if "tat" in net:
            np.testing.assert_allclose(rp.numpy().reshape(-1), g[net + "/r_pred"].reshape(-1), **tol)
        else:
            assert rp == 0                                                         # model.py:248
        # the reference's own one-env layout is accepted too and returns the reference's shapes
        with torch.no_grad():
            v1, a1, e1, lp1, (h1, c1), _ = m((states[2], (hx[2], cx[2])), True)
        assert v1.shape == (2, 1) and e1.shape == (2, 1) and h1.shape == (2, 128)
        assert [int(x) for x in a1] == g[net + "/actions"][2].tolist()
        np.testing.assert_allclose(v1.numpy(), g[net + "/values"][2], **tol)


def test_toeplitz_gemm_stem_matches_conv2d_forward_and_backward():
    torch.manual_seed(1)
    for frames in (1, 2):
        enc = CNN_maze((1, 13, 13), frames)
        enc.conv1.bias.data.normal_(); enc.conv2.bias.data.normal_()
        x = torch.randint(0, 5, (9, frames, 1, 13, 13)).float()
        a, b = enc(x), enc.forward_conv2d(x)
        np.testing.assert_allclose(a.detach().numpy(), b.detach().numpy(), atol=2e-6, rtol=1e-5)
        ga = torch.autograd.grad((a ** 2).sum(), list(enc.parameters()))
        gb = torch.autograd.grad((b ** 2).sum(), list(enc.parameters()))
        for u, v in zip(ga, gb):
            assert (u - v).abs().max() <= 1e-5 * max(1.0, v.abs().max())
        enc.cache_dense(True)                      # cached expansion gives the same result
        np.testing.assert_allclose(enc(x).detach().numpy(), a.detach().numpy(), atol=0, rtol=0)
        enc.cache_dense(False)


def test_sampling_branch_shapes_and_init():
    m, _ = _model("tat-maze-lstm")
    n = 5
    states = torch.randint(0, 5, (n, 2, 1, 1, 13, 13)).float()
    hx = torch.zeros(n, 2, 128); cx = torch.zeros(n, 2, 128)
    v, a, e, lp, (h, c), rp = m((states, (hx, cx)))
    assert v.shape == (n, 2, 1) and e.shape == (n, 2, 1) and lp.shape == (n, 2, 1) and rp.shape == (n, 1)
    assert a[0].shape == (n,) and a[0].dtype == torch.int64 and h.shape == (n, 2, 128)
    # weights_init re-initialises every Conv/Linear last (model.py:130,187): zero biases, LSTM biases zero
    for name, p in m.named_parameters():
        if name.endswith("bias") or "bias_" in name:
            assert float(p.abs().max()) == 0.0, name
