"""Fused policy-side HIP kernels (include/atr_policy.h) as autograd functions.

stem(x, conv1, conv2): CNN_maze's conv(1->16,k3,s2,p1)+ReLU+conv(16->32,k3,s2,p1)+ReLU on [M,169] frames in one
launch forward and three launches backward (csrc/stem_hip.hip). Used on CUDA/ROCm tensors; on CPU tensors the
model evaluates the same network with plain PyTorch ops (model.CNN_maze.forward_dense / forward_conv2d)."""
import ctypes as C

import torch

from . import vec_env

_lib = None


def lib():
    global _lib
    if _lib is None:
        L = vec_env.load_library()
        vp, ll = C.c_void_p, C.c_longlong
        L.atr_stem_forward.restype = C.c_int
        L.atr_stem_forward.argtypes = [vp, ll, vp, vp, vp, vp, vp, ll, vp]
        L.atr_stem_workspace_floats.restype = ll
        L.atr_stem_workspace_floats.argtypes = [ll]
        L.atr_stem_backward.restype = C.c_int
        L.atr_stem_backward.argtypes = [vp, ll] + [vp] * 10 + [ll, vp]
        L.atr_sample_actions.restype = C.c_int
        L.atr_sample_actions.argtypes = [vp, vp, vp, vp, vp, C.c_ulonglong, C.c_int, C.c_int, C.c_int, vp]
        _lib = L
    return _lib


def _p(t):
    return C.c_void_p(t.data_ptr())


def _stream(t):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


class _Stem(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        if x.stride(1) != 1 or x.stride(0) < 169:      # rows must be contiguous; the row stride is free
            x = x.contiguous()
        w1c, b1c, w2c, b2c = w1.contiguous(), b1.contiguous(), w2.contiguous(), b2.contiguous()
        M = x.shape[0]
        y = torch.empty((M, 512), dtype=torch.float32, device=x.device)
        rc = lib().atr_stem_forward(_p(x), x.stride(0), _p(w1c), _p(b1c), _p(w2c), _p(b2c), _p(y), M, _stream(x))
        if rc != 0:
            raise RuntimeError("atr_stem_forward failed (%d)" % rc)
        ctx.save_for_backward(x, y, w1c, b1c, w2c)
        ctx.shapes = (w1.shape, w2.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, w1, b1, w2 = ctx.saved_tensors
        dy = dy.contiguous()
        M = x.shape[0]
        L = lib()
        ws = torch.empty(L.atr_stem_workspace_floats(M), dtype=torch.float32, device=x.device)
        dw1, db1 = torch.empty(144, device=x.device), torch.empty(16, device=x.device)
        dw2, db2 = torch.empty(4608, device=x.device), torch.empty(32, device=x.device)
        rc = L.atr_stem_backward(_p(x), x.stride(0), _p(y), _p(dy), _p(w1), _p(b1), _p(w2), _p(dw1), _p(db1), _p(dw2),
                                 _p(db2), _p(ws), M, _stream(x))
        if rc != 0:
            raise RuntimeError("atr_stem_backward failed (%d)" % rc)
        return None, dw1.view(ctx.shapes[0]), db1, dw2.view(ctx.shapes[1]), db2


def rows169(x):
    """View x [..., 13, 13]-shaped frames as [M, 169] rows WITHOUT copying when the frames are evenly strided (e.g.
    one agent's slice of the env's obs tensor, or of the stacked rollout buffer); falls back to reshape (copy)."""
    m = x.numel() // 169
    lead = [(sz, st) for sz, st in zip(x.shape[:-2], x.stride()[:-2]) if sz != 1]
    if x.stride(-1) == 1 and x.stride(-2) == 13:
        ok, stride = True, 169
        if lead:
            stride = lead[-1][1]
            for (sz, st), (sz2, st2) in zip(lead[:-1], lead[1:]):
                ok = ok and st == sz2 * st2
        if ok and stride >= 169:
            return x.as_strided((m, 169), (stride, 1), x.storage_offset())
    return x.reshape(m, 169)


def stem(x, conv1, conv2):
    """x: frames [..., 13, 13] float32 on the GPU (any evenly strided view) -> [M, 512]."""
    return _Stem.apply(rows169(x), conv1.weight, conv1.bias, conv2.weight, conv2.bias)


class ActionSampler(object):
    """Fused actor head for the rollout: action ~ Categorical(softmax(W h + b)) in one launch (csrc/policy_hip.hip).
    Holds the device-side stream counter (advanced by every call, hipGraph-safe) and the Philox seed."""

    def __init__(self, device, seed=None):
        self.counter = torch.zeros(1, dtype=torch.int64, device=device)
        self.seed = int(torch.initial_seed() if seed is None else seed) & 0xFFFFFFFFFFFFFFFF

    @torch.no_grad()
    def __call__(self, h, linear):
        h = h.contiguous()
        n, R = h.shape
        A = linear.weight.shape[0]
        actions = torch.empty(n, dtype=torch.int64, device=h.device)
        rc = lib().atr_sample_actions(_p(h), _p(linear.weight), _p(linear.bias), _p(actions), _p(self.counter),
                                      self.seed, n, R, A, _stream(h))
        if rc != 0:
            raise RuntimeError("atr_sample_actions failed (%d)" % rc)
        return actions
