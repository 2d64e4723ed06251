"""ctypes + autograd bindings of the policy-side HIP kernels (include/atr_policy.h). GPU tensors only — on CPU tensors the
model evaluates the same network with plain PyTorch ops (model.CNN_maze.forward_dense_stem / forward_conv2d, nn.LSTMCell).

  stem / stem_into / stem_into2 / stem_cached   CNN_maze's conv(1->16,k3,s2,p1)+ReLU+conv(16->32,k3,s2,p1)+ReLU on
                                                13x13 frames: one launch forward, two backward (csrc/stem_hip.hip)
  lstm_cell / lstm_cell_into / lstm_cell_act_into, lstm_sequence(_cached)
                                                masked LSTMCell step(s), the actor's cell + head + draw in one launch,
                                                the whole recurrence as one autograd node (csrc/lstm_hip.hip)
  ActionSampler                                 actor head + categorical draw (csrc/policy_hip.hip)
  gae_returns, heads_values, heads_loss         returns / GAE, critic values, heads + A3C loss terms with analytic
                                                gradients (csrc/lstm_hip.hip, csrc/heads_hip.hip)
  gemm_tn                                       weight-gradient GEMMs x1^T x2 (+ bias gradient, row mask) on the f32
                                                matrix cores (csrc/gemm_tn_hip.hip)
  linear_relu_cached, lstm_sequence_cached      cached-forward autograd nodes of the rollout cache (model.RolloutCache)"""
import ctypes as C

import torch

from . import vec_env

_lib = None


class ActStepArgs(C.Structure):
    """atr_act_step of include/atr_policy.h."""
    _fields_ = [("ig", C.c_void_p * 2), ("hg", C.c_void_p * 2), ("bias", C.c_void_p * 2), ("c_prev", C.c_void_p * 2),
                ("h_out", C.c_void_p * 2), ("c_out", C.c_void_p * 2), ("acts", C.c_void_p * 2),
                ("actor_w", C.c_void_p * 2), ("actor_b", C.c_void_p * 2), ("emb", C.c_void_p), ("done_prev", C.c_void_p),
                ("actions_out", C.c_void_p), ("counter", C.c_void_p), ("seed", C.c_ulonglong), ("ordinal", C.c_uint),
                ("A", C.c_int), ("N", C.c_int), ("R", C.c_int), ("hm_out", C.c_void_p * 2), ("hm_ld", C.c_longlong)]


class CoopStepArgs(C.Structure):
    """atr_coop_step of include/atr_policy.h."""
    _fields_ = [("y", C.c_void_p * 2), ("fc_w", C.c_void_p * 2), ("fc_b", C.c_void_p * 2), ("w_cat", C.c_void_p * 2),
                ("ldy", C.c_longlong * 2), ("kfc", C.c_int * 2), ("fh", C.c_void_p), ("gates", C.c_void_p),
                ("fh_pstride", C.c_longlong), ("fh_ld", C.c_longlong), ("F", C.c_int), ("workgroups", C.c_int),
                ("probe", C.c_void_p)]


class LinearArgs(C.Structure):
    """atr_linear_args of include/atr_policy.h."""
    _fields_ = [("a", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("c", C.c_void_p), ("lda", C.c_longlong),
                ("ldw", C.c_longlong), ("ldc", C.c_longlong), ("stride_a", C.c_longlong), ("stride_w", C.c_longlong),
                ("stride_c", C.c_longlong), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("batch", C.c_int),
                ("relu", C.c_int), ("workspace", C.c_void_p), ("workspace_bytes", C.c_longlong)]


class GateCellArgs(C.Structure):
    """atr_gate_cell_args of include/atr_policy.h."""
    _fields_ = [("a", C.c_void_p * 2), ("w", C.c_void_p * 2), ("bias", C.c_void_p * 2), ("pre", C.c_void_p * 2),
                ("c_prev", C.c_void_p * 2), ("h_out", C.c_void_p * 2), ("c_out", C.c_void_p * 2), ("done_prev", C.c_void_p),
                ("lda", C.c_longlong), ("cell", C.c_int * 2), ("N", C.c_int), ("K", C.c_int), ("R", C.c_int), ("probe", C.c_void_p)]


class PairLinearArgs(C.Structure):
    """atr_pair_linear_args of include/atr_policy.h."""
    _fields_ = [("a1", C.c_void_p * 2), ("w1", C.c_void_p * 2), ("a2", C.c_void_p * 2), ("w2", C.c_void_p * 2),
                ("bias", C.c_void_p * 2), ("c", C.c_void_p * 2), ("lda1", C.c_longlong * 2), ("lda2", C.c_longlong * 2),
                ("ldc", C.c_longlong * 2), ("k1", C.c_int * 2), ("k2", C.c_int * 2), ("done", C.c_void_p),
                ("M", C.c_int), ("N", C.c_int), ("relu", C.c_int)]


class HeadsLossArgs(C.Structure):
    """atr_heads_loss_args of include/atr_policy.h."""
    _fields_ = [("h", C.c_void_p), ("actions", C.c_void_p), ("act_n", C.c_longlong), ("act_tstride", C.c_longlong),
                ("ret", C.c_void_p), ("gae", C.c_void_p), ("val", C.c_void_p), ("stride", C.c_int), ("off", C.c_int),
                ("r_aux", C.c_void_p), ("aux_stride", C.c_int), ("aux_off", C.c_int), ("wa", C.c_void_p), ("ba", C.c_void_p),
                ("wc", C.c_void_p), ("waux", C.c_void_p), ("baux", C.c_void_p), ("scale", C.c_float),
                ("scale_aux", C.c_float), ("w_ent", C.c_float), ("dh", C.c_void_p), ("grads_and_sums", C.c_void_p),
                ("workspace", C.c_void_p), ("stats_out", C.c_void_p), ("rows", C.c_longlong), ("R", C.c_int), ("A", C.c_int)]


class GemmTnProblem(C.Structure):
    """atr_gemm_tn_problem of include/atr_policy.h."""
    _fields_ = [("x1", C.c_void_p), ("x2", C.c_void_p), ("c", C.c_void_p), ("row_scale", C.c_void_p),
                ("row_scale_shift", C.c_longlong), ("colsum0", C.c_void_p), ("colsum1", C.c_void_p), ("M", C.c_int),
                ("N", C.c_int), ("ld1", C.c_longlong), ("ld2", C.c_longlong)]


def lib():
    global _lib
    if _lib is None:
        L = vec_env.load_library()
        vp, ll = C.c_void_p, C.c_longlong
        L.atr_stem_forward.restype = C.c_int
        L.atr_stem_forward.argtypes = [vp, ll, vp, vp, vp, vp, vp, ll, vp]
        L.atr_stem_forward2.restype = C.c_int
        L.atr_stem_forward2.argtypes = [vp, ll, vp, vp, vp, vp, vp, ll] * 2 + [vp]
        L.atr_stem_workspace_floats.restype = ll
        L.atr_stem_workspace_floats.argtypes = [ll]
        L.atr_stem_backward.restype = C.c_int
        L.atr_stem_backward.argtypes = [vp, ll] + [vp] * 10 + [ll, vp]
        for name in ("atr_stem_forward", "atr_stem_forward2", "atr_stem_backward"):      # the u8-frame twins
            f = getattr(L, name + "_u8")
            f.restype, f.argtypes = C.c_int, getattr(L, name).argtypes
        L.atr_sample_actions.restype = C.c_int
        L.atr_sample_actions.argtypes = [vp, vp, vp, vp, vp, C.c_ulonglong, C.c_uint, C.c_int, C.c_int, C.c_int, C.c_int, vp]
        i32 = C.c_int
        L.atr_lstm_cell_forward.restype = i32
        L.atr_lstm_cell_forward.argtypes = [vp, vp, vp, vp, ll, vp, vp, vp, ll, vp, ll, vp, ll, i32, i32, i32, vp]
        L.atr_lstm_cell_forward_act.restype = i32
        L.atr_lstm_cell_forward_act.argtypes = [vp] * 11 + [i32, vp, vp, C.c_ulonglong, C.c_uint, i32, i32, vp]
        L.atr_lstm_cell_forward_act1.restype = i32
        L.atr_lstm_cell_forward_act1.argtypes = [vp] * 12 + [i32, vp, vp, C.c_ulonglong, C.c_uint, i32, i32, vp]
        L.atr_lstm_cell_forward_act2.restype = i32
        L.atr_lstm_cell_forward_act2.argtypes = [vp] * 5 + [ll, vp, vp, ll, vp, ll, vp, ll] + [vp] * 4 + [i32, vp, vp,
                                                 C.c_ulonglong, C.c_uint, i32, i32, vp]
        L.atr_embed_add.restype = i32
        L.atr_embed_add.argtypes = [vp, vp, vp, vp, ll, ll, ll, vp, ll, i32, i32, vp]
        L.atr_embed_add_ld.restype = i32
        L.atr_embed_add_ld.argtypes = [vp, ll, vp, vp, vp, ll, ll, ll, vp, ll, i32, i32, vp]
        L.atr_relu_backward_ld.restype = i32
        L.atr_relu_backward_ld.argtypes = [vp, vp, ll, vp, ll, i32, vp]
        L.atr_lt_init.restype = i32
        L.atr_lt_init.argtypes = [C.c_char_p]
        L.atr_linear.restype = i32
        L.atr_linear.argtypes = [C.POINTER(LinearArgs), vp]
        L.atr_linear_plan_info.restype = i32
        L.atr_linear_plan_info.argtypes = [C.POINTER(LinearArgs), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32),
                                           C.POINTER(C.c_float), C.POINTER(i32), C.POINTER(i32)]
        L.atr_linear_set_choice.restype = i32
        L.atr_linear_set_choice.argtypes = [C.POINTER(LinearArgs), i32, i32]
        L.atr_lt_library_info.restype = i32
        L.atr_lt_library_info.argtypes = [C.POINTER(i32), C.POINTER(i32), C.c_char_p, i32]
        L.atr_linear_kernel_name.restype = i32
        L.atr_linear_kernel_name.argtypes = [C.POINTER(LinearArgs), C.c_char_p, i32]
        L.atr_lt_last_error.restype = C.c_char_p
        L.atr_embed_grad_workspace_floats.restype = ll
        L.atr_embed_grad_workspace_floats.argtypes = [ll, i32, i32]
        L.atr_embed_grad.restype = i32
        L.atr_embed_grad.argtypes = [vp, vp, ll, ll, ll, vp, vp, vp, ll, i32, i32, vp]
        L.atr_lstm_bptt.restype = i32
        L.atr_lstm_bptt.argtypes = [vp, vp, vp, vp, ll, vp, ll, vp, vp, vp, ll, vp, vp, i32, i32, i32, i32, vp]
        L.atr_gate_cell.restype = i32
        L.atr_gate_cell.argtypes = [C.POINTER(GateCellArgs), vp]
        L.atr_gate_cell_workgroups.restype = i32
        L.atr_gate_cell_workgroups.argtypes = [i32]
        L.atr_lstm_bptt_pre2.restype = i32
        L.atr_lstm_bptt_pre2.argtypes = [vp, vp, vp, vp, ll, vp, vp, vp, i32, i32, vp, ll, vp, ll, vp, vp, vp, ll, vp, vp, vp,
                                         i32, i32, i32, i32, vp]
        L.atr_lstm_bptt_act_sums_floats.restype = ll
        L.atr_lstm_bptt_act_sums_floats.argtypes = [i32]
        L.atr_embed_fold.restype = i32
        L.atr_embed_fold.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp]
        L.atr_lstm_bptt_pre.restype = i32
        L.atr_lstm_bptt_pre.argtypes = [vp, vp, vp, vp, ll, vp, vp, vp, i32, i32, vp, ll, vp, ll, vp, vp, vp, ll, vp, vp,
                                        i32, i32, i32, i32, vp]
        L.atr_pair_linear.restype = i32
        L.atr_pair_linear.argtypes = [C.POINTER(PairLinearArgs), vp]
        L.atr_act_env_step.restype = i32
        L.atr_act_env_step.argtypes = [vp, C.POINTER(ActStepArgs), vp, i32, vp, vp, vp]
        L.atr_coop_env_step.restype = i32
        L.atr_coop_env_step.argtypes = [vp, C.POINTER(ActStepArgs), C.POINTER(CoopStepArgs), vp, i32, vp, vp, vp]
        L.atr_actor_step.restype = i32
        L.atr_actor_step.argtypes = [vp] * 12 + [i32, i32, i32, vp]
        L.atr_lstm_cell_backward.restype = i32
        L.atr_lstm_cell_backward.argtypes = [vp, ll, vp, vp, vp, vp, vp, ll, vp, ll, vp, ll, vp, ll, i32, i32, i32, i32, vp]
        L.atr_heads_values.restype = i32
        L.atr_heads_values.argtypes = [vp, vp, vp, vp, ll, i32, i32, i32, vp]
        L.atr_heads_values2.restype = i32
        L.atr_heads_values2.argtypes = [vp, vp, vp, i32, vp, vp, vp, i32, vp, ll, i32, i32, vp]
        L.atr_heads_loss_multi.restype = i32
        L.atr_heads_loss_multi.argtypes = [C.POINTER(HeadsLossArgs), i32, C.c_float, vp]
        L.atr_heads_workspace_floats.restype = ll
        L.atr_heads_workspace_floats.argtypes = [ll, i32, i32]
        L.atr_heads_loss.restype = i32
        L.atr_heads_loss.argtypes = [vp, vp, vp, vp, vp, i32, i32, vp, i32, i32, vp, vp, vp, vp, vp, C.c_float, C.c_float,
                                     C.c_float, vp, vp, vp, ll, i32, i32, vp]
        L.atr_gemm_tn_workspace_floats.restype = ll
        L.atr_gemm_tn_workspace_floats.argtypes = [ll, i32, i32]
        L.atr_gemm_tn.restype = i32
        L.atr_gemm_tn.argtypes = [vp, vp, vp, vp, ll, i32, i32, vp, vp, vp]
        L.atr_gemm_tn_grouped_workspace_floats.restype = ll
        L.atr_gemm_tn_grouped_workspace_floats.argtypes = [C.POINTER(GemmTnProblem), i32, ll]
        L.atr_gemm_tn_set_corun.restype = i32
        L.atr_gemm_tn_set_corun.argtypes = [i32]
        L.atr_gemm_tn_grouped.restype = i32
        L.atr_gemm_tn_grouped.argtypes = [C.POINTER(GemmTnProblem), i32, ll, vp, vp]
        L.atr_scatter_segments.restype = i32
        L.atr_scatter_segments.argtypes = [C.POINTER(C.c_void_p), C.POINTER(ll), C.POINTER(i32), i32, vp, vp]
        L.atr_gae_returns.restype = i32
        L.atr_gae_returns.argtypes = [vp, vp, vp, C.c_float, C.c_float, vp, vp, i32, i32, i32, vp]
        L.atr_rollout_begin.restype = i32
        L.atr_rollout_begin.argtypes = [vp, vp, vp, vp, ll, vp, vp, ll, i32, i32, i32, vp]
        L.atr_rollout_end.restype = i32
        L.atr_rollout_end.argtypes = [vp, vp, ll, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
        L.atr_rollout_end2.restype = i32
        L.atr_rollout_end2.argtypes = [vp, vp, ll, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, ll, vp, vp]
        L.atr_adam_step.restype = i32
        L.atr_adam_step.argtypes = [vp] * 7 + [C.c_double] * 5 + [i32, ll, vp]
        L.atr_rmsprop_step.restype = i32
        L.atr_rmsprop_step.argtypes = [vp] * 3 + [C.c_double] * 4 + [ll, vp]
        _lib = L
    return _lib


def _p(t):
    return C.c_void_p(t.data_ptr())


def _stem_fn(name, x):
    """The stem entry point for x's element type: float32 frames, or the env's u8 observations (decoded inside conv1)."""
    if x.dtype == torch.uint8:
        return getattr(lib(), name + "_u8")
    if x.dtype != torch.float32:
        raise TypeError("stem frames must be float32 or uint8, got %s" % x.dtype)
    return getattr(lib(), name)


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
        rc = _stem_fn("atr_stem_forward", x)(_p(x), x.stride(0), _p(w1c), _p(b1c), _p(w2c), _p(b2c), _p(y), M, _stream(x))
        if rc != 0:
            raise RuntimeError("atr_stem_forward failed (%d)" % rc)
        ctx.save_for_backward(x, y, w1c, b1c, w2c)
        ctx.shapes = (w1.shape, w2.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, w1, b1, w2 = ctx.saved_tensors
        return (None,) + _stem_backward(x, y, dy, w1, b1, w2, ctx.shapes)


def _stem_backward(x, y, dy, w1, b1, w2, shapes):
    dy = dy.contiguous()
    M = x.shape[0]
    L = lib()
    ws = torch.empty(L.atr_stem_workspace_floats(M), dtype=torch.float32, device=x.device)
    dw1, db1 = torch.empty(144, device=x.device), torch.empty(16, device=x.device)
    dw2, db2 = torch.empty(4608, device=x.device), torch.empty(32, device=x.device)
    rc = _stem_fn("atr_stem_backward", x)(_p(x), x.stride(0), _p(y), _p(dy), _p(w1), _p(b1), _p(w2), _p(dw1), _p(db1),
                                          _p(dw2), _p(db2), _p(ws), M, _stream(x))
    if rc != 0:
        raise RuntimeError("atr_stem_backward failed (%d)" % rc)
    return dw1.view(shapes[0]), db1, dw2.view(shapes[1]), db2


class _StemCached(torch.autograd.Function):
    """The stem as an autograd node whose forward was already evaluated (by stem_into during the rollout, with the
    same weights): forward hands back the stored output, backward is the ordinary stem backward."""

    @staticmethod
    def forward(ctx, x, y, w1, b1, w2, b2):
        w1c, b1c, w2c = w1.contiguous(), b1.contiguous(), w2.contiguous()
        ctx.save_for_backward(x, y, w1c, b1c, w2c)
        ctx.shapes = (w1.shape, w2.shape)
        return y.view_as(y)

    @staticmethod
    def backward(ctx, dy):
        x, y, w1, b1, w2 = ctx.saved_tensors
        return (None, None) + _stem_backward(x, y, dy, w1, b1, w2, ctx.shapes)


@torch.no_grad()
def stem_into(x, conv1, conv2, out):
    """No-grad stem forward of frames x [..., 13, 13] written into out [M, 512] (a slot of the rollout cache)."""
    x = rows169(x)
    if x.stride(1) != 1 or x.stride(0) < 169:
        x = x.contiguous()
    M = x.shape[0]
    assert out.is_contiguous() and out.numel() == M * 512
    rc = _stem_fn("atr_stem_forward", x)(_p(x), x.stride(0), _p(conv1.weight), _p(conv1.bias), _p(conv2.weight),
                                         _p(conv2.bias), _p(out), M, _stream(x))
    if rc != 0:
        raise RuntimeError("atr_stem_forward failed (%d)" % rc)
    return out


@torch.no_grad()
def stem_into2(xa, enc_a, out_a, xb, enc_b, out_b):
    """stem_into for two encoders (different weights) in one launch: the rollout step's tracker and target stems."""
    xs = []
    for x in (xa, xb):
        x = rows169(x)
        if x.stride(1) != 1 or x.stride(0) < 169:
            x = x.contiguous()
        xs.append(x)
    args = []
    for x, enc, out in ((xs[0], enc_a, out_a), (xs[1], enc_b, out_b)):
        assert out.is_contiguous() and out.numel() == x.shape[0] * 512
        args += [_p(x), x.stride(0), _p(enc.conv1.weight), _p(enc.conv1.bias), _p(enc.conv2.weight), _p(enc.conv2.bias),
                 _p(out), x.shape[0]]
    assert xs[0].dtype == xs[1].dtype
    rc = _stem_fn("atr_stem_forward2", xs[0])(*(args + [_stream(xs[0])]))
    if rc != 0:
        raise RuntimeError("atr_stem_forward2 failed (%d)" % rc)
    return out_a, out_b


def stem_cached(x, y, conv1, conv2):
    x = rows169(x)
    if x.stride(1) != 1 or x.stride(0) < 169:
        x = x.contiguous()
    return _StemCached.apply(x, y, conv1.weight, conv1.bias, conv2.weight, conv2.bias)


class _LinearReluCached(torch.autograd.Function):
    """relu(x W^T + b) whose value f is already known (computed in the rollout with the same weights): forward
    returns f, backward is the usual one (dx, dW, db through the ReLU mask f > 0)."""

    @staticmethod
    def forward(ctx, x, w, b, f):
        ctx.save_for_backward(x, w, f, b)
        return f.view_as(f)

    @staticmethod
    def backward(ctx, df):
        x, w, f, b = ctx.saved_tensors
        if f.is_cuda and not f.is_contiguous() and f.dim() == 2 and f.stride(1) == 1 and f.stride(0) % 4 == 0 and f.shape[1] % 4 == 0:
            dpre = relu_backward_ld(df.contiguous(), f)     # (f is a column block of the rollout's [features | k h] rows)
        else:
            dpre = torch.ops.aten.threshold_backward(df.contiguous(), f, 0.0)
        r = _deferred.add(dpre, x, w, biases=(b,)) if _deferred is not None else None
        if r is not None:                  # joins the backward pass's grouped weight-gradient launch
            dw, (db,) = r
        else:
            dw, db = gemm_tn(dpre, x, colsum=True)
        return dpre @ w, dw, db, None


def linear_relu_cached(x, linear, f):
    return _LinearReluCached.apply(x, linear.weight, linear.bias, f)


@torch.no_grad()
def relu_backward_ld(df, f):
    """df * (f > 0) for a dense df [rows, C] and an activation f [rows, C] whose rows are f.stride(0) floats apart."""
    rows, Cc = df.shape
    out = torch.empty_like(df)
    rc = lib().atr_relu_backward_ld(_p(df), _p(f), f.stride(0), _p(out), rows, Cc, _stream(df))
    if rc != 0:
        raise RuntimeError("atr_relu_backward_ld failed (%d)" % rc)
    return out


_lt_ready = False


def _lt_init():
    """Hand csrc/lt_gemm.cpp the libhipblaslt.so PyTorch itself uses (it resolves the entry points from that copy)."""
    global _lt_ready
    if _lt_ready:
        return
    import os
    cands = [os.path.join(os.path.dirname(torch.__file__), "lib", "libhipblaslt.so")]
    try:
        for ln in open("/proc/self/maps"):
            if "libhipblaslt" in ln:
                cands.insert(0, ln.split()[-1])
                break
    except OSError:
        pass
    L = lib()
    for path in cands:
        if os.path.exists(path) and L.atr_lt_init(path.encode()) == 0:
            _lt_ready = True
            return
    raise RuntimeError("atr_lt_init failed: %s" % L.atr_lt_last_error().decode())


_lt_state = None


def lt_available():
    """Whether atr_linear can be used (PyTorch's libhipblaslt.so found and initialised). False only on an unexpected
    runtime: the model then keeps round 3's rollout step (library GEMMs through torch + two gate tensors) — slower, still the
    HIP path — and says so once on stderr."""
    global _lt_state
    if _lt_state is None:
        try:
            _lt_init()
            _lt_state = True
        except (RuntimeError, OSError) as ex:
            import sys
            print("active_tracking_rl_amd: hipBLASLt direct path unavailable (%s); using the torch GEMM path" % (ex,), file=sys.stderr)
            _lt_state = False
    return _lt_state


def _linear_args(a, w, out, bias, relu, workspace):
    g = LinearArgs()
    if a.dim() == 2:
        a, w, out = a.unsqueeze(0), w.unsqueeze(0), out.unsqueeze(0)
    B, M, K = a.shape
    N = w.shape[1]
    assert w.shape == (B, N, K) and out.shape == (B, M, N)
    for t in (a, w, out):
        assert t.is_cuda and t.dtype == torch.float32 and t.stride(2) == 1 and t.data_ptr() % 16 == 0
    g.a, g.w, g.c = a.data_ptr(), w.data_ptr(), out.data_ptr()
    g.lda, g.ldw, g.ldc = a.stride(1), w.stride(1), out.stride(1)
    g.stride_a, g.stride_w, g.stride_c = a.stride(0), w.stride(0), out.stride(0)
    g.bias = bias.data_ptr() if bias is not None else None
    assert bias is None or (B == 1 and bias.is_contiguous() and bias.numel() == N)
    g.M, g.N, g.K, g.batch, g.relu = M, N, K, B, 1 if relu else 0
    if workspace is not None:
        g.workspace, g.workspace_bytes = workspace.data_ptr(), workspace.numel() * workspace.element_size()
    return g


LT_TUNING_FILE = __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)),
                                            "lt_tuning_gfx950.json")
LT_SOURCES = {0: "none", 1: "timed", 2: "recorded", 3: "first-usable", 4: "refused", 5: "refused-timed"}
_lt_choices, _lt_seen, _lt_status = None, {}, "not loaded"


def _lt_key(g):
    # everything that selects a kernel, including the workspace limit handed to the heuristic (another limit = another list)
    return "M%d_N%d_K%d_b%d_lda%d_ldw%d_ldc%d_sa%d_sw%d_sc%d_relu%d_bias%d_ws%d" % (
        g.M, g.N, g.K, g.batch, g.lda, g.ldw, g.ldc, g.stride_a, g.stride_w, g.stride_c, g.relu, 1 if g.bias else 0,
        g.workspace_bytes if g.workspace else 0)


def _lt_family(key):
    """The problem key without the output's batch stride. A product must not change with the PLACE its batches are written to
    (the gate GEMM writes a [2, N, 512] scratch tensor or — since round 5 — slot t of the rollout's [2, T, N, 512] store: the
    same products), but hipBLASLt's heuristic takes the stride into account and a timing run can crown another tile shape,
    whose fp32 sums differ in the last bit: two runs of one seed then part ways at the first sampled action. So the kernel is
    chosen per FAMILY: every output stride runs the choice recorded (or first timed) for the family."""
    import re
    return re.sub(r"_sc\d+", "", key)


def _lt_family_choices():
    """{family: (candidate index, solution index)}: of a family's records the one with the largest output stride (the rollout
    store at the bench's rollout length — the default path)."""
    fam = {}
    import re
    for k, v in lt_choices().items():
        sc = int(re.search(r"_sc(\d+)", k).group(1)) if re.search(r"_sc(\d+)", k) else 0
        f = _lt_family(k)
        if f not in fam or sc > fam[f][0]:
            fam[f] = (sc, v)
    return {f: v for f, (sc, v) in fam.items()}


_lt_family_first = {}       # families without a record: the choice the first problem of the family ended up with


def lt_library():
    """{'version', 'git', 'header_version'} of the hipBLASLt build csrc/lt_gemm.cpp loaded (PyTorch's copy) and of the header it
    was compiled against."""
    _lt_init()
    v, hv, rev = C.c_int(0), C.c_int(0), C.create_string_buffer(256)
    if lib().atr_lt_library_info(C.byref(v), C.byref(hv), rev, 256) != 0:
        raise RuntimeError("atr_lt_library_info failed")
    return dict(version=v.value, git=rev.value.decode(errors="replace"), header_version=hv.value)


def lt_choices():
    """{problem key: (candidate index, solution index)} recorded by tools/tune_lt.py (lt_tuning_gfx950.json next to this file;
    ATR_LT_TUNING=0 ignores it): the kernel choices of csrc/lt_gemm.cpp are then the same in every run — what
    tunableop_gfx950.csv is for torch's GEMMs — and nothing is timed at first use. The record is only used when it was made
    with the hipBLASLt build that is loaded now (version + revision, as TunableOp's validators do): a position in the
    heuristic's list means nothing on another build. On a mismatch the file is ignored, every problem is timed at first use
    (outside capture) and lt_tuning_status() says so; within a matching build csrc/lt_gemm.cpp still checks each record's
    solution index against the list it gets."""
    global _lt_choices, _lt_status
    if _lt_choices is None:
        import json
        import os
        _lt_choices = {}
        if os.environ.get("ATR_LT_TUNING", "1") == "0":
            _lt_status = "timed at first use (ATR_LT_TUNING=0)"
        elif not os.path.exists(LT_TUNING_FILE):
            _lt_status = "timed at first use (no tuning file)"
        else:
            try:
                rec = json.load(open(LT_TUNING_FILE))
                have = lt_library()
                want = rec.get("hipblaslt") or {}
                if want.get("version") != have["version"] or want.get("git") != have["git"]:
                    _lt_status = ("timed at first use (tuning file made with hipBLASLt %s %s, loaded %s %s)"
                                  % (want.get("version"), want.get("git"), have["version"], have["git"]))
                else:
                    for k, v in rec.get("choices", {}).items():
                        idx, sol = int(v["index"]), int(v.get("solution", -1))
                        if idx >= 0:
                            _lt_choices[k] = (idx, sol)
                    _lt_status = "recorded (hipBLASLt %s %s, %d problems)" % (have["version"], have["git"], len(_lt_choices))
            except (ValueError, OSError, KeyError, TypeError, AttributeError) as ex:
                _lt_choices = {}
                _lt_status = "timed at first use (unreadable tuning file: %s)" % (ex,)
    return _lt_choices


def lt_tuning_status():
    """One line for the bench record: where the direct-hipBLASLt kernel choices of this process come from, and how many of the
    problems seen so far run a recorded / timed / refused choice."""
    if _lt_state is False:
        return "torch GEMM path (direct hipBLASLt unavailable)"
    if _lt_choices is None:
        return "not used"
    counts = {}
    for k, g in _lt_seen.items():
        info = _plan_info(g)
        if info is not None:
            counts[LT_SOURCES.get(info["source"], "?")] = counts.get(LT_SOURCES.get(info["source"], "?"), 0) + 1
    tail = ", ".join("%d %s" % (n, k) for k, n in sorted(counts.items()))
    return _lt_status + ("; in use: " + tail if tail else "")


def _plan_info(g):
    cand, ch, tu, us, sol, src = C.c_int(0), C.c_int(0), C.c_int(0), C.c_float(0), C.c_int(-1), C.c_int(0)
    if lib().atr_linear_plan_info(C.byref(g), C.byref(cand), C.byref(ch), C.byref(tu), C.byref(us), C.byref(sol), C.byref(src)) != 0:
        return None
    return dict(candidates=cand.value, chosen=ch.value, tuned=bool(tu.value), best_us=us.value, solution=sol.value,
                source=src.value)


def lt_chosen():
    """{problem key: plan info + kernel name} of every problem this process has run and timed (for tools/tune_lt.py)."""
    out = {}
    for k, g in _lt_seen.items():
        info = _plan_info(g)
        if info is not None and info["tuned"]:
            name = C.create_string_buffer(512)
            lib().atr_linear_kernel_name(C.byref(g), name, 512)
            info["kernel"] = name.value.decode(errors="replace")
            out[k] = info
    return out


@torch.no_grad()
def linear_lt(a, w, out, bias=None, relu=False, workspace=None):
    """out = act(a @ w.T + bias) through hipBLASLt called directly (atr_linear, csrc/lt_gemm.cpp): a [M, K], w [N, K] (nn.Linear
    layout), out [M, N] — or a batch of them as 3-D tensors (no bias then). Rows may be strided (a column block of wider rows:
    the fc output inside the rollout's [features | k h] rows). workspace: a uint8 device tensor private to the calling chain
    of launches (two streams must not share one); None = the library's own."""
    _lt_init()
    g = _linear_args(a, w, out, bias, relu, workspace)
    L = lib()
    key = _lt_key(g)
    fresh = key not in _lt_seen
    if fresh:
        _lt_seen[key] = g
        fam = _lt_family(key)
        rec = _lt_family_choices().get(fam) or _lt_family_first.get(fam)
        if rec is not None:
            L.atr_linear_set_choice(C.byref(g), rec[0], rec[1])
    if L.atr_linear(C.byref(g), _stream(a)) != 0:
        raise RuntimeError("atr_linear failed: %s" % L.atr_lt_last_error().decode())
    if fresh and rec is None:
        info = _plan_info(g)
        # (only a TIMED choice is handed to the family's other strides: a first-usable plan made inside a stream capture is
        # re-timed later and may change — its siblings would keep the untimed kernel and the family would split again)
        if info is not None and info["chosen"] >= 0 and info["solution"] >= 0 and info["tuned"]:
            _lt_family_first[fam] = (info["chosen"], info["solution"])
    return out


def linear_lt_info(a, w, out, bias=None, relu=False, workspace=None):
    """(candidates, chosen index, tuned?, best us) of the kernel choice for this problem (after its first linear_lt call)."""
    return _plan_info(_linear_args(a, w, out, bias, relu, workspace))


def linear_lt_set_choice(a, w, out, index, solution=-1, bias=None, relu=False, workspace=None):
    """Pre-select candidate `index` (and, with solution >= 0, insist that it is that solution) for this problem: what
    lt_choices() does from the tuning file, by hand (tests, A/B runs)."""
    _lt_init()
    g = _linear_args(a, w, out, bias, relu, workspace)
    _lt_seen.setdefault(_lt_key(g), g)
    if lib().atr_linear_set_choice(C.byref(g), int(index), int(solution)) != 0:
        raise RuntimeError("atr_linear_set_choice failed: %s" % lib().atr_lt_last_error().decode())


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
    Holds the device-side stream counter (hipGraph-safe) and the Philox seed. Stand-alone calls advance the counter
    themselves; inside begin_block() ... the counter is advanced once and the calls are told apart by their ordinal
    (saves one tiny launch per call in the rollout)."""

    def __init__(self, device, seed=None):
        self.counter = torch.zeros(1, dtype=torch.int64, device=device)
        self.seed = int(torch.initial_seed() if seed is None else seed) & 0xFFFFFFFFFFFFFFFF
        self._ordinal = None

    @torch.no_grad()
    def begin_block(self, launch=True):
        """Advance the counter once (in stream order); the following calls use ordinals 1, 2, ... until end_block.
        launch=False: the caller's next launch bumps the counter itself (fused.rollout_begin(counter=...))."""
        if not launch:
            self._ordinal = 0
            return
        dummy = self.counter
        rc = lib().atr_sample_actions(_p(dummy), _p(dummy), _p(dummy), _p(dummy), _p(self.counter), self.seed, 0, 1,
                                      0, 4, 1, _stream(self.counter))
        if rc != 0:
            raise RuntimeError("atr_sample_actions failed (%d)" % rc)
        self._ordinal = 0

    def end_block(self):
        if self._ordinal is not None:
            self._last = self._ordinal
        self._ordinal = None

    def reopen_block(self):
        """Continue the ordinals of the block that end_block() closed last (the learner's bootstrap step draws once more
        after the rollout, under the same counter value)."""
        self._ordinal = getattr(self, "_last", None)

    @torch.no_grad()
    def __call__(self, h, linear, out=None):
        h = h.contiguous()
        n, R = h.shape
        A = linear.weight.shape[0]
        actions = torch.empty(n, dtype=torch.int64, device=h.device) if out is None else out
        assert actions.is_contiguous() and actions.dtype == torch.int64 and actions.numel() == n
        if self._ordinal is None:
            ordinal, bump = 0, 1
        else:
            self._ordinal += 1
            ordinal, bump = self._ordinal, 0
        rc = lib().atr_sample_actions(_p(h), _p(linear.weight), _p(linear.bias), _p(actions), _p(self.counter),
                                      self.seed, ordinal, bump, n, R, A, _stream(h))
        if rc != 0:
            raise RuntimeError("atr_sample_actions failed (%d)" % rc)
        return actions


def _pn(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def lstm_cell(ig, hg, c_prev, keep=None, done=None):
    """One no-grad LSTMCell step for the rollout (csrc/lstm_hip.hip): ig [N,4R] = x W_ih^T + b_ih + b_hh, hg [N,4R] =
    h_prev W_hh^T with h_prev NOT masked; the previous step's episode-boundary mask (keep [N] float, or done [N] uint8)
    is applied inside. Returns (h, c) [N,R]."""
    N, R = c_prev.shape
    h, c = torch.empty_like(c_prev), torch.empty_like(c_prev)
    rc = lib().atr_lstm_cell_forward(_p(ig), None, _p(hg), _p(c_prev), 0, _pn(keep), _pn(done), _p(h), 0, _p(c), 0,
                                     None, 0, 1, N, R, _stream(ig))
    if rc != 0:
        raise RuntimeError("atr_lstm_cell_forward failed (%d)" % rc)
    return h, c


class _LstmSeq(torch.autograd.Function):
    """Two (or one) independent LSTMCells over T time-major steps with the per-step episode mask, as ONE autograd
    node: per step one batched hidden GEMM + one fused cell launch forward, one fused cell-backward launch + one
    batched GEMM backward; the weight gradient of W_hh is a single GEMM over all T*N rows at the end.
    ig0/ig1 [T*N,4R] (input projections incl. both biases), whh [P,R,4R] (= W_hh^T per player), h0/c0 [P,N,R]
    (already masked), keep [T,N] float. Returns per player h_seq [T,N,R] (the un-masked step outputs, what the heads
    read) and, last, the un-masked final cell state [P,N,R] (not differentiable)."""

    @staticmethod
    def forward(ctx, ig0, ig1, whh, h0, c0, keep):
        L = lib()
        P, N, R = h0.shape
        T = keep.shape[0]
        dev = h0.device
        ig0, whh, keep = ig0.contiguous(), whh.contiguous(), keep.contiguous()
        ig1 = ig1.contiguous() if ig1 is not None else None
        h_all = torch.empty((P, T + 1, N, R), dtype=torch.float32, device=dev)   # slot 0 = h0, slot t+1 = h_t
        c_all = torch.empty((P, T + 1, N, R), dtype=torch.float32, device=dev)
        acts = torch.empty((P, T, N, 4 * R), dtype=torch.float32, device=dev)
        h_all[:, 0].copy_(h0)
        c_all[:, 0].copy_(c0)
        hg = torch.empty((P, N, 4 * R), dtype=torch.float32, device=dev)
        st = _stream(h0)
        ps, pa, step, astep = (T + 1) * N * R, T * N * 4 * R, N * R * 4, N * 4 * R * 4   # strides: floats / bytes
        for t in range(T):
            torch.bmm(h_all[:, t], whh, out=hg)
            off = t * N * 4 * R * 4
            rc = L.atr_lstm_cell_forward(
                C.c_void_p(ig0.data_ptr() + off), C.c_void_p(ig1.data_ptr() + off) if ig1 is not None else None,
                _p(hg), C.c_void_p(c_all.data_ptr() + t * step), ps, _p(keep[t - 1]) if t else None, None,
                C.c_void_p(h_all.data_ptr() + (t + 1) * step), ps, C.c_void_p(c_all.data_ptr() + (t + 1) * step), ps,
                C.c_void_p(acts.data_ptr() + t * astep), pa, P, N, R, st)
            if rc != 0:
                raise RuntimeError("atr_lstm_cell_forward failed (%d)" % rc)
        ctx.save_for_backward(whh, keep, h_all, c_all, acts)
        ctx.two = ig1 is not None
        c_last = c_all[:, T].clone()                                             # un-masked final cell state
        ctx.mark_non_differentiable(c_last)
        return tuple(h_all[p, 1:] for p in range(P)) + (c_last,)                 # per player [T,N,R], contiguous

    @staticmethod
    def backward(ctx, *dhs):
        whh, keep, h_all, c_all, acts = ctx.saved_tensors
        dG, dhn, dcc, dwhh = _lstm_bptt(whh, keep, h_all, c_all, acts, dhs[:-1])
        return dG[0], (dG[1] if ctx.two else None), dwhh, dhn, dcc, None


use_fused_bptt = True   # the whole recurrence backward as ONE launch (csrc/bptt_hip.hip) instead of 2 launches per step


def activate_preacts(pre, info):
    """The activated gates (i, f, g, o) [P, T, N, 4R] from stored pre-activations (`info`: see _lstm_bptt) with tensor ops — what
    the fused BPTT kernel recomputes in registers; only the per-step fallback recurrence needs them materialised."""
    P, T, N, R4 = pre.shape
    R = R4 // 4
    x = pre + torch.stack(list(info["bias"]), 0).view(P, 1, 1, R4)
    if info.get("emb") is not None and 0 <= info["emb_player"] < P:
        x[info["emb_player"]] = info["emb"][info["act"]] + x[info["emb_player"]]
    x = x.view(P, T, N, 4, R)
    return torch.stack([torch.sigmoid(x[..., 0, :]), torch.sigmoid(x[..., 1, :]), torch.tanh(x[..., 2, :]),
                        torch.sigmoid(x[..., 3, :])], -2).reshape(P, T, N, R4).contiguous()


def _lstm_bptt(whh, keep, h_all, c_all, acts, dhs, whh_nn=None, want_dwhh=True, pre=None):
    """Back-propagation through time over stored activations: dhs = per-player dL/dh_seq [T,N,R] (None = zero).
    Returns dG [P, T*N, 4R] (= dL/d ig), dL/dh0, dL/dc0 [P,N,R] and dL/dW_hh^T [P,R,4R]. whh_nn: per-player weight_hh [4R,R]
    (nn layout; made from whh [P,R,4R] when absent).
    pre: None, or dict(bias=[per-player b_ih + b_hh [4R]], emb=[n_act, 4R] or None, emb_player, act=[T, N] int64 view of the
    tracker's actions) — `acts` then holds the rollout's gate PRE-activations without bias (model._act_step's one-GEMM path keeps
    the GEMM's output instead of the activated gates) and the activations are recomputed inside the kernel."""
    L = lib()
    P, T1, N, R = h_all.shape
    T = T1 - 1
    dev = h_all.device
    fused_ok = (use_fused_bptt and R == 128 and h_all.is_cuda and acts.is_contiguous() and c_all.is_contiguous()
                and keep.is_contiguous())
    if pre is not None and not fused_ok:
        acts, pre = activate_preacts(acts, pre), None
    dG = torch.empty((P, T, N, 4 * R), dtype=torch.float32, device=dev)
    dhn = torch.empty((P, N, R), dtype=torch.float32, device=dev)
    dcc = torch.empty((P, N, R), dtype=torch.float32, device=dev)
    st = _stream(h_all)
    ps, pa, step, astep = (T + 1) * N * R, T * N * 4 * R, N * R * 4, N * 4 * R * 4
    if fused_ok:
        if whh_nn is None:
            wt = whh.transpose(1, 2).contiguous()
            whh_nn = [wt[p] for p in range(P)]
        whh_nn = [w if w.is_contiguous() else w.contiguous() for w in whh_nn]
        dh_c = [d.contiguous() if d is not None else None for d in dhs]
        if pre is not None:
            bias = [b.contiguous() for b in pre["bias"]]
            emb = pre.get("emb")
            ep = int(pre.get("emb_player", -1))
            if emb is None or not (0 <= ep < P):
                emb, ep, act, n_act, act_ts = None, -1, None, 0, 0
            else:
                emb, act = emb.contiguous(), pre["act"]
                assert act.dtype == torch.int64 and act.shape == (T, N) and act.stride(1) == 1
                n_act, act_ts = emb.shape[0], act.stride(0)
            sums = None
            if pre.get("want_sums") and emb is not None and n_act == 4:
                # per row tile of the embedding's player: column sums of dG by the row's tracker action (see embed_fold)
                sums = torch.empty(L.atr_lstm_bptt_act_sums_floats(N), dtype=torch.float32, device=dev)
            pre["act_sums"] = sums
            rc = L.atr_lstm_bptt_pre2(_pn(dh_c[0]), _pn(dh_c[1]) if P > 1 else None, _p(keep), _p(acts), acts.stride(0),
                                      _p(bias[0]), _p(bias[1]) if P > 1 else None, _pn(emb), ep, n_act, _pn(act), act_ts,
                                      _p(c_all), ps, _p(whh_nn[0]), _p(whh_nn[1]) if P > 1 else None, _p(dG), pa, _p(dhn),
                                      _p(dcc), _pn(sums), P, T, N, R, st)
        else:
            rc = L.atr_lstm_bptt(_pn(dh_c[0]), _pn(dh_c[1]) if P > 1 else None, _p(keep), _p(acts), pa, _p(c_all), ps,
                                 _p(whh_nn[0]), _p(whh_nn[1]) if P > 1 else None, _p(dG), pa, _p(dhn), _p(dcc), P, T, N, R, st)
        if rc != 0:
            raise RuntimeError("atr_lstm_bptt failed (%d)" % rc)
    else:
        dhs = [torch.zeros((T, N, R), dtype=torch.float32, device=dev) if d is None else d.contiguous() for d in dhs]
        pd = (dhs[1].data_ptr() - dhs[0].data_ptr()) // 4 if P > 1 else 0          # player stride between the two grads
        whh_t = whh.transpose(1, 2)                                               # [P,4R,R]
        for t in range(T - 1, -1, -1):
            rc = L.atr_lstm_cell_backward(
                C.c_void_p(dhs[0].data_ptr() + t * step), pd, _p(dhn), _p(dcc), _p(keep[t]),
                _p(keep[t - 1]) if t else None, C.c_void_p(acts.data_ptr() + t * astep), pa,
                C.c_void_p(c_all.data_ptr() + (t + 1) * step), ps, C.c_void_p(c_all.data_ptr() + t * step), ps,
                C.c_void_p(dG.data_ptr() + t * astep), pa, 1 if t < T - 1 else 0, P, N, R, st)
            if rc != 0:
                raise RuntimeError("atr_lstm_cell_backward failed (%d)" % rc)
            torch.bmm(dG[:, t], whh_t, out=dhn)                                   # gradient into h_{t-1}
    if not want_dwhh:          # (the caller registers dW_hh with the grouped weight-gradient launch)
        return dG.view(P, T * N, 4 * R), dhn, dcc, None
    # W_hh: sum_t (k_{t-1} h_{t-1})^T dG_t as one GEMM per player over all T*N rows
    kprev = torch.cat([torch.ones_like(keep[:1]), keep[:-1]], 0)              # [T, N]: mask on h_{t-1}
    dG = dG.view(P, T * N, 4 * R)
    if use_gemm_tn and R % 128 == 0 and T * N >= 4096:
        # the mask is applied to h's rows on their way into the GEMM kernel's LDS tiles
        dwhh = torch.stack([gemm_tn(h_all[p, :T].reshape(T * N, R), dG[p], row_scale=kprev.reshape(T * N))
                            for p in range(P)], 0)
    else:
        hm = (h_all[:, :T] * kprev.view(1, T, N, 1)).view(P, T * N, R)
        dwhh = torch.bmm(hm.transpose(1, 2), dG)
    return dG, dhn, dcc, dwhh


class _LstmSeqCached(torch.autograd.Function):
    """The input projection + masked recurrence of both players as an autograd node whose forward was already
    evaluated step by step during the rollout (lstm_cell_into, same weights): h_all/c_all/acts hold every step.
    Inputs: per-player features [T*N,F] and LSTMCell parameters; output: per-player h_seq [T,N,R]."""

    @staticmethod
    def forward(ctx, keep, h_all, c_all, acts, need, *fw):
        P = h_all.shape[0]
        feats, wih, whh_l = fw[:P], fw[P:2 * P], fw[2 * P:3 * P]
        # ([P, R, 4R] transposed copy: only the per-step fallback recurrence reads it — the fused BPTT kernel takes weight_hh as is)
        fused_path = use_fused_bptt and h_all.shape[-1] == 128 and h_all.is_cuda
        whh = h_all.new_empty(0) if fused_path else torch.stack([w.t() for w in whh_l], 0).contiguous()
        ctx.save_for_backward(keep.contiguous(), h_all, c_all, acts, whh, *feats, *wih, *whh_l, *fw[3 * P:5 * P], *fw[5 * P:])
        ctx.P = P
        ctx.fold = len(fw) == 5 * P + 2           # fc_action_tracker's weight and bias ride along: the embedding is folded
        need, ctx.hm, ctx.pre = need              # (lstm_sequence_cached packs the non-tensor arguments together)
        ctx.need = tuple(bool(x) for x in need) if need is not None else (True,) * P
        return tuple(h_all[p, 1:] for p in range(P))

    @staticmethod
    def backward(ctx, *dhs):
        P = ctx.P
        keep, h_all, c_all, acts, whh = ctx.saved_tensors[:5]
        feats, wih = ctx.saved_tensors[5:5 + P], ctx.saved_tensors[5 + P:5 + 2 * P]
        whh_nn = ctx.saved_tensors[5 + 2 * P:5 + 3 * P]          # weight_hh [4R, R] as nn.LSTMCell holds it
        bih, bhh = ctx.saved_tensors[5 + 3 * P:5 + 4 * P], ctx.saved_tensors[5 + 4 * P:5 + 5 * P]
        fa_w, fa_b = (ctx.saved_tensors[5 + 5 * P], ctx.saved_tensors[5 + 5 * P + 1]) if ctx.fold else (None, None)
        dfa = (None, None)
        dfeat, dwih, db, dwhh_l = [None] * P, [None] * P, [None] * P, [None] * P
        db2 = None
        q = _deferred
        T, N, R = h_all.shape[1] - 1, h_all.shape[2], h_all.shape[3]
        if whh.numel() == 0 and not (use_fused_bptt and R == 128 and acts.is_contiguous() and c_all.is_contiguous()
                                     and keep.is_contiguous()):
            whh = torch.stack([w.t() for w in whh_nn], 0).contiguous()
        if all(ctx.need):
            groups = [list(range(P))]
        else:       # a player the loss does not train (train-mode 0 / 1): none of its recurrence is back-propagated
            groups = [[p] for p in range(P) if ctx.need[p]]
        for grp in groups:
            a, b = grp[0], grp[-1] + 1
            defer = q is not None and use_fused_bptt and R == 128 and T * N >= 4096
            pre = None
            if ctx.pre is not None:          # (this group's slice of the stored pre-activations' side information)
                pre = dict(ctx.pre, bias=list(ctx.pre["bias"][a:b]), emb_player=ctx.pre.get("emb_player", -1) - a)
                pre["want_sums"] = ctx.fold and a <= ctx.pre.get("emb_player", -1) < b
            dG, _, _, dwhh = _lstm_bptt(whh[a:b], keep, h_all[a:b], c_all[a:b], acts[a:b], dhs[a:b], whh_nn=list(whh_nn[a:b]),
                                        want_dwhh=not defer, pre=pre)
            fold_p = ctx.pre.get("emb_player", -1) if (ctx.fold and pre is not None and pre.get("want_sums")) else -1
            if fold_p >= 0 and pre.get("act_sums") is None:
                raise RuntimeError("folded tracker-action embedding: the BPTT launch did not return the by-action sums of dG")
            for i, p in enumerate(grp):
                dfeat[p] = dG[i] @ wih[p]
                if p == fold_p:
                    # (feats[p] are the RAW fc features: dW_ih's product misses S^T E, added by embed_fold once the product is there)
                    fold_args = (pre["act_sums"], fa_w, fa_b, wih[p])
                    dfa = (torch.empty_like(fa_w), torch.empty_like(fa_b))
                if defer:
                    # both products of this player contract dG: registered with the grouped launch. dW_hh^T = dG^T (k h):
                    # the mask on h_{t-1} is keep[t-1] = the keep array shifted by one step of N rows, applied to dG's rows
                    r1 = q.add(dG[i], feats[p], wih[p], biases=(bih[p], bhh[p]))
                    if r1 is None:
                        r2 = None
                    elif ctx.hm is not None:       # (k h rows stored by the rollout: nothing to mask)
                        r2 = q.add(dG[i], ctx.hm[p], whh_nn[p])
                    else:
                        r2 = q.add(dG[i], h_all[p, :T].reshape(T * N, R), whh_nn[p], row_scale=keep, shift=N)
                    if r1 is not None and r2 is not None:
                        dwih[p], (dbi, dbh) = r1
                        dwhh_l[p] = r2[0]
                        db[p] = dbi
                        db2 = db2 if db2 is not None else [None] * P
                        db2[p] = dbh
                        if p == fold_p:          # (the slice is filled by the grouped launch at flush(): the fold follows it)
                            q.after.append(lambda fa_=fold_args, dw_=dwih[p], out_=dfa: embed_fold(*fa_, dw_, *out_))
                        continue
                    if r1 is not None:         # (cannot happen for R = 128; keep the queue consistent if it ever does)
                        raise RuntimeError("grouped weight gradients: dW_hh could not join the group after dW_ih did")
                    kprev = torch.cat([torch.ones_like(keep[:1]), keep[:-1]], 0).reshape(T * N)
                    dwhh_l[p] = gemm_tn(h_all[p, :T].reshape(T * N, R), dG[i], row_scale=kprev).t()
                    dwih[p], db[p] = gemm_tn(dG[i], feats[p], colsum=True)
                    if p == fold_p:
                        dwih[p] = dwih[p].contiguous()
                        embed_fold(*fold_args, dwih[p], *dfa)
                    continue
                dwih[p], db[p] = gemm_tn(dG[i], feats[p], colsum=True)
                if p == fold_p:
                    dwih[p] = dwih[p].contiguous()
                    embed_fold(*fold_args, dwih[p], *dfa)
                dwhh_l[p] = dwhh[i].t()
        if ctx.fold and ctx.pre is not None and ctx.need[ctx.pre.get("emb_player", 0)] and dfa[0] is None:
            raise RuntimeError("folded tracker-action embedding: no group of the backward pass produced its gradients")
        db_hh = tuple(db2[p] if (db2 is not None and db2[p] is not None) else db[p] for p in range(P))
        out = (None, None, None, None, None) + tuple(dfeat) + tuple(dwih) + tuple(dwhh_l) + tuple(db) + db_hh
        return out + (tuple(dfa) if ctx.fold else ())


@torch.no_grad()
def embed_fold(act_sums, fa_w, fa_b, wih, dwih, dfa_w, dfa_b):
    """The tracker-action embedding's share of the target's backward pass from the by-action column sums of dG (atr_embed_fold):
    dwih [4R, C] += S^T E in place, dfa_w [C, 4] / dfa_b [C] = the gradients of fc_action_tracker. act_sums: what
    atr_lstm_bptt_pre2 left per row tile."""
    J, Cc = wih.shape
    assert fa_w.shape == (Cc, 4) and fa_w.is_contiguous() and fa_b.is_contiguous() and wih.is_contiguous() and dwih.is_contiguous()
    assert dwih.shape == (J, Cc) and act_sums.numel() % (4 * J) == 0
    S = torch.empty((4, J), dtype=torch.float32, device=wih.device)
    rc = lib().atr_embed_fold(_p(act_sums), act_sums.numel() // (4 * J), _p(fa_w), _p(fa_b), _p(wih), _p(dwih), _p(dfa_w), _p(dfa_b),
                              _p(S), J, Cc, _stream(wih))
    if rc != 0:
        raise RuntimeError("atr_embed_fold failed (%d)" % rc)


def lstm_sequence_cached(lstms, feats, keep, h_all, c_all, acts, need=None, hm=None, pre=None, fold=None):
    """feats: per-player [T*N, F] (with grad); lstms: the nn.LSTMCells; stored activations from the rollout.
    need: per player, whether anything upstream of its hidden sequence is trained (None = all).
    hm: per player the MASKED previous hidden rows k_{t-1} h_{t-1} as [T*N, R] (row-strided views are fine) when the rollout
    stored them (model._act_step's one-GEMM path) — dW_hh then needs no row factors.
    pre: `acts` holds gate pre-activations without bias instead of activated gates (see _lstm_bptt)."""
    P = len(lstms)
    args = list(feats) + [l.weight_ih for l in lstms] + [l.weight_hh for l in lstms] + \
        [l.bias_ih for l in lstms] + [l.bias_hh for l in lstms]
    if fold is not None:     # fold = the tracker-aware player's fc_action_tracker: feats[pre['emb_player']] are then its RAW fc
        args += [fold.weight, fold.bias]     # features (no f + E[a] materialised), the embedding's gradients come from this node
    return _LstmSeqCached.apply(keep, h_all, c_all, acts, (need, hm, pre), *args)


def embed_fold_ok(pre, h_all, c_all, keep, fa):
    """Whether lstm_sequence_cached can fold the tracker-action embedding (the fused BPTT launch over stored pre-activations,
    the four-move action table, parameters as the kernels read them)."""
    return bool(fold_embedding and pre is not None and pre.get("emb") is not None and use_fused_bptt and h_all.is_cuda
                and h_all.shape[-1] == 128 and c_all.is_contiguous() and keep.is_contiguous()
                and fa.weight.shape[1] == 4 and fa.weight.is_contiguous() and fa.weight.data_ptr() % 16 == 0)


fold_embedding = __import__("os").environ.get("ATR_FOLD_EMBEDDING", "1") != "0"


@torch.no_grad()
def lstm_cell_into(ig, hg, c_prev, done, h_out, c_out, acts):
    """The rollout's LSTM step for one player, writing h, c and the activated gates into rollout-cache slots."""
    N, R = c_prev.shape
    rc = lib().atr_lstm_cell_forward(_p(ig), None, _p(hg), _p(c_prev), 0, None, _pn(done), _p(h_out), 0, _p(c_out), 0,
                                     _p(acts), 0, 1, N, R, _stream(ig))
    if rc != 0:
        raise RuntimeError("atr_lstm_cell_forward failed (%d)" % rc)


@torch.no_grad()
def lstm_cell_act2_into(ig, hg, biases, c_prev, done, h_out, c_out, acts, sampler, actors, actions_out):
    """Both players' actor step after the GEMMs in ONE launch (players independent of each other's action): ig / hg
    [2,N,4R] (ig without bias), biases = (b0, b1) [4R] each, c_prev / h_out / c_out [2,N,R] and acts [2,N,4R] views with
    a player stride, actions_out [2,N] int64 contiguous. Only valid inside sampler.begin_block(); consumes two ordinals."""
    P, N, R = c_prev.shape
    assert P == 2 and sampler._ordinal is not None and ig.is_contiguous() and hg.is_contiguous() and actions_out.is_contiguous()
    for t in (c_prev, h_out, c_out, acts):
        assert t.stride(-1) == 1 and t.stride(-2) == t.shape[-1]
    ordinal = sampler._ordinal + 1
    sampler._ordinal += 2
    rc = lib().atr_lstm_cell_forward_act2(
        _p(ig), _p(hg), _p(biases[0]), _p(biases[1]), _p(c_prev), c_prev.stride(0), _pn(done), _p(h_out), h_out.stride(0),
        _p(c_out), c_out.stride(0), _pn(acts), acts.stride(0) if acts is not None else 0, _p(actors[0].weight),
        _p(actors[0].bias), _p(actors[1].weight), _p(actors[1].bias), actors[0].weight.shape[0], _p(actions_out),
        _p(sampler.counter), sampler.seed, ordinal, N, R, _stream(ig))
    if rc != 0:
        raise RuntimeError("atr_lstm_cell_forward_act2 failed (%d)" % rc)
    return actions_out


@torch.no_grad()
def lstm_cell_act_into(ig, hg, c_prev, done, h_out, c_out, acts, sampler, actor, actions_out, emb=None, act_in=None,
                       bias=None):
    """The actor's whole per-player step after the two GEMMs in ONE launch: masked LSTM cell (+ optional
    emb[act_in] added to the gates) written into the rollout-cache slots, then actor head + categorical draw on the
    fresh hidden row into actions_out (int64 [N]). bias [4R]: added to the pre-activations here (ig from a bias-free
    GEMM). Only valid inside sampler.begin_block()."""
    N, R = c_prev.shape
    assert sampler._ordinal is not None
    sampler._ordinal += 1
    if bias is not None:
        rc = lib().atr_lstm_cell_forward_act1(_p(ig), _p(hg), _p(bias), _p(c_prev), _pn(done), _p(h_out), _p(c_out),
                                              _pn(acts), _pn(emb), _pn(act_in), _p(actor.weight), _p(actor.bias),
                                              actor.weight.shape[0], _p(actions_out), _p(sampler.counter), sampler.seed,
                                              sampler._ordinal, N, R, _stream(ig))
    else:
        rc = lib().atr_lstm_cell_forward_act(_p(ig), _p(hg), _p(c_prev), _pn(done), _p(h_out), _p(c_out), _pn(acts),
                                             _pn(emb), _pn(act_in), _p(actor.weight), _p(actor.bias), actor.weight.shape[0],
                                             _p(actions_out), _p(sampler.counter), sampler.seed, sampler._ordinal, N, R,
                                             _stream(ig))
    if rc != 0:
        raise RuntimeError("atr_lstm_cell_forward_act failed (%d)" % rc)
    return actions_out


def _rows(t):
    """(pointer, row stride) of a 2-D float32 tensor whose rows are contiguous."""
    assert t.dim() == 2 and t.stride(1) == 1 and t.dtype == torch.float32 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0
    return t.data_ptr(), t.stride(0)


@torch.no_grad()
def pair_linear(a1, w1, out, bias=None, a2=None, w2=None, done=None, relu=False):
    """out[p] = act(a1[p] @ w1[p].T [+ (k a2[p]) @ w2[p].T] + bias[p]) for the two players in ONE launch
    (atr_pair_linear, csrc/pair_gemm_hip.hip) — made for small row counts. a1[p] [M, k1p], w1[p] [N, k1p] (nn.Linear
    layout, contiguous), a2 / w2 optional second term, done [M] uint8: k = (done == 0) scales a2's rows; out[p] [M, N]."""
    g = PairLinearArgs()
    M, N = a1[0].shape[0], w1[0].shape[0]
    for p in range(2):
        assert a1[p].shape[0] == M and w1[p].shape[0] == N and w1[p].is_contiguous() and w1[p].shape[1] == a1[p].shape[1]
        g.a1[p], g.lda1[p] = _rows(a1[p])
        g.w1[p], g.k1[p] = w1[p].data_ptr(), a1[p].shape[1]
        if a2 is not None:
            assert w2[p].is_contiguous() and w2[p].shape == (N, a2[p].shape[1]) and a2[p].shape[0] == M
            g.a2[p], g.lda2[p] = _rows(a2[p])
            g.w2[p], g.k2[p] = w2[p].data_ptr(), a2[p].shape[1]
        g.bias[p] = bias[p].data_ptr() if bias is not None and bias[p] is not None else None
        assert out[p].shape == (M, N)
        g.c[p], g.ldc[p] = _rows(out[p])
    g.done = done.data_ptr() if done is not None else None
    g.M, g.N, g.relu = M, N, 1 if relu else 0
    rc = lib().atr_pair_linear(C.byref(g), _stream(a1[0]))
    if rc != 0:
        raise RuntimeError("atr_pair_linear failed (%d)" % rc)
    return out


@torch.no_grad()
def act_env_step(env_core, ig, hg, biases, c_prev, done, h_out, c_out, acts, sampler, actors, actions_out, emb=None,
                 env_out=None, hm_out=None):
    """The END of a rollout step as ONE launch (atr_act_env_step, csrc/track2d_hip.hip k_act_step): both players' cells +
    actor heads + draws (tracker first; emb [A,4R] adds emb[a_tracker] to the target's pre-activations) and, with
    env_core (a vec_env.VecTrack2D) and env_out = (obs [N,2,13,13] u8 | f32, rew [N,2], done [N] u8), the env step with
    those actions. ig / hg: per-player [N,4R] contiguous (hg[p] None: ig[p] is the whole pre-activation); biases (b0, b1)
    [4R] or None; c_prev / h_out / c_out per-player [N,R]; acts per-player [N,4R] or None; actions_out int64 [2,N].
    hm_out (with the env step only): per-player [N,R] views with a common row stride — they receive the fresh hidden rows
    zeroed where this step's done flag is set (what the next step's LSTMCell GEMM reads).
    Same results as lstm_cell_act_into x 2 + env.step. Only valid inside sampler.begin_block(); consumes two ordinals."""
    N, R = h_out[0].shape
    assert sampler._ordinal is not None and actions_out.is_contiguous() and actions_out.shape == (2, N)
    a = ActStepArgs()
    for p in range(2):
        for t in (ig[p], c_prev[p], h_out[p], c_out[p]):
            assert t is None or (t.is_contiguous() and t.dtype == torch.float32)
        # (ig[0] None: the tracker's cell already ran as the epilogue of the step's gate product, fused.gate_cell)
        a.ig[p] = ig[p].data_ptr() if ig[p] is not None else None
        a.hg[p] = hg[p].data_ptr() if hg is not None and hg[p] is not None else None
        a.bias[p] = biases[p].data_ptr() if biases is not None and biases[p] is not None else None
        a.c_prev[p] = c_prev[p].data_ptr() if c_prev[p] is not None else None
        a.h_out[p], a.c_out[p] = h_out[p].data_ptr(), (c_out[p].data_ptr() if c_out[p] is not None else None)
        a.acts[p] = acts[p].data_ptr() if acts is not None and acts[p] is not None else None
        assert hg is None or hg[p] is None or hg[p].is_contiguous()
        assert acts is None or acts[p] is None or acts[p].is_contiguous()
        a.actor_w[p], a.actor_b[p] = actors[p].weight.data_ptr(), actors[p].bias.data_ptr()
    a.emb = emb.data_ptr() if emb is not None else None
    a.done_prev = done.data_ptr() if done is not None else None
    a.actions_out, a.counter, a.seed = actions_out.data_ptr(), sampler.counter.data_ptr(), sampler.seed
    a.ordinal = sampler._ordinal + 1
    sampler._ordinal += 2
    a.A, a.N, a.R = actors[0].weight.shape[0], N, R
    if hm_out is not None:
        assert env_core is not None and hm_out[0].shape == (N, R) and hm_out[1].shape == (N, R)
        assert hm_out[0].stride(1) == 1 and hm_out[1].stride(1) == 1 and hm_out[0].stride(0) == hm_out[1].stride(0)
        a.hm_out[0], a.hm_out[1], a.hm_ld = hm_out[0].data_ptr(), hm_out[1].data_ptr(), hm_out[0].stride(0)
    L = lib()
    if env_core is None:
        rc = L.atr_act_env_step(None, C.byref(a), None, 0, None, None, _stream(h_out[0]))
    else:
        obs, rew, done_out = env_out
        assert obs.is_contiguous() and rew.is_contiguous() and done_out.is_contiguous() and obs.dtype in (torch.uint8, torch.float32)
        rc = L.atr_act_env_step(env_core.h, C.byref(a), _p(obs), 1 if obs.dtype == torch.uint8 else 0, _p(rew), _p(done_out),
                                _stream(h_out[0]))
    if rc != 0:
        raise RuntimeError("atr_act_env_step failed (%d): %s" % (rc, L.t2d_last_error().decode()))
    return actions_out


@torch.no_grad()
def gate_cell(fh, w_cat, biases, pre, c_prev, done, h_out, c_out, cell=(True, False), probe=None):
    """The step's LSTMCell product with the cell as its epilogue (atr_gate_cell, csrc/gate_cell_hip.hip): fh [2, N, K] rows
    [features | k h_prev] (a common row stride), w_cat [2, 4R, K] contiguous, pre [2, N, 4R] with contiguous [N, 4R] slabs
    (receives the product without bias), and for players with cell[p]: c_prev / h_out / c_out per-player [N, R] contiguous, biases per-player [4R]."""
    P, N, K = fh.shape
    a = GateCellArgs()
    assert P == 2 and fh.stride(2) == 1 and w_cat.is_contiguous() and w_cat.shape[0] == 2 and w_cat.shape[2] == K
    assert pre is None or (pre.shape == (2, N, w_cat.shape[1]) and pre[0].is_contiguous() and pre[1].is_contiguous())
    for p in range(2):
        a.a[p], a.w[p] = fh[p].data_ptr(), w_cat[p].data_ptr()
        a.pre[p] = pre[p].data_ptr() if pre is not None else None
        a.cell[p] = 1 if cell[p] else 0
        if cell[p]:
            for t in (c_prev[p], h_out[p], c_out[p], biases[p]):
                assert t.is_contiguous() and t.dtype == torch.float32
            a.bias[p], a.c_prev[p] = biases[p].data_ptr(), c_prev[p].data_ptr()
            a.h_out[p], a.c_out[p] = h_out[p].data_ptr(), c_out[p].data_ptr()
    a.done_prev = done.data_ptr() if done is not None else None
    a.lda, a.N, a.K, a.R = fh.stride(1), N, K, w_cat.shape[1] // 4
    a.probe = probe.data_ptr() if probe is not None else None
    rc = lib().atr_gate_cell(C.byref(a), _stream(fh))
    if rc != 0:
        raise RuntimeError("atr_gate_cell failed (%d)" % rc)


_stream_cus = {}


def register_stream_cus(stream, n_cus):
    """A stream created with a CU mask (train.cu_masked_stream): kernels launched on it see only n_cus compute units. Launches
    whose workgroups wait for each other (coop_env_step) size their grid by it."""
    _stream_cus[int(stream.cuda_stream)] = int(n_cus)


def stream_cus(device, stream=None):
    """Compute units available to `stream` (default: the current one): its registered CU mask, else the whole device."""
    st = torch.cuda.current_stream(device) if stream is None else stream
    n = _stream_cus.get(int(st.cuda_stream))
    return n if n is not None else int(torch.cuda.get_device_properties(device).multi_processor_count)


COOP_MAX_UNITS, COOP_WAVES = 16, 8       # (csrc/coop_gemm.h: kCoopMaxUnits, kCoopWaves)
COOP_PROBE = None                        # an int64 device tensor [workgroups, 8]: the kernel's phase stamps (tools/coop_step_timeline.py)


def coop_step_supported(N, F, R, workgroups):
    """The shape limits of atr_coop_env_step: every XCD takes N / 8 envs in whole 16-row tiles, a workgroup at most
    COOP_MAX_UNITS tiles per layer and COOP_WAVES env pairs."""
    if R != 128 or F <= 0 or F % 32 or N % 128 or workgroups < 8 or workgroups % 8:
        return False
    P, Rx = workgroups // 8, N // 8
    units_a, units_b = 2 * (Rx // 16) * (F // 32), 2 * (Rx // 16) * (4 * R // 32)
    return (-(-units_a // P) <= COOP_MAX_UNITS and -(-units_b // P) <= COOP_MAX_UNITS and -(-(Rx // 2) // P) <= COOP_WAVES)


@torch.no_grad()
def coop_env_step(env_core, ys, fcs, fh_t, w_cat, gates, biases, c_prev, done, h_out, c_out, acts, sampler, actors, actions_out,
                  emb, env_out, hm_out, workgroups):
    """A rollout step after the stem as ONE launch (atr_coop_env_step, csrc/track2d_hip.hip k_coop_step): fc + ReLU of both
    encoders (ys[p] [N, K_p] stem outputs, fcs[p] nn.Linear) into the feature columns of fh_t [2, N, F + R], the LSTMCell GEMM
    of both players over those rows (w_cat [2, 4R, F + R]) into `gates` [2, N, 4R], then what act_env_step does (cells, heads,
    draws, env step; biases = b_ih + b_hh per player; hm_out: the next step's hidden columns). Every XCD owns an eighth of the
    envs end to end; `workgroups` = the CUs the launch's stream may use (all of them must be free to be resident at once)."""
    N, R = c_prev[0].shape
    Fd = fcs[0].weight.shape[0]
    assert sampler._ordinal is not None and actions_out.is_contiguous() and actions_out.shape == (2, N)
    assert fh_t.shape == (2, N, Fd + R) and fh_t.stride(2) == 1 and fh_t.stride(1) == Fd + R
    assert w_cat.is_contiguous() and w_cat.shape == (2, 4 * R, Fd + R) and gates.is_contiguous() and gates.shape == (2, N, 4 * R)
    a, k = ActStepArgs(), CoopStepArgs()
    for p in range(2):
        for t in (c_prev[p], h_out[p], c_out[p], biases[p], ys[p]):
            assert t.is_contiguous() and t.dtype == torch.float32
        assert ys[p].shape == (N, fcs[p].weight.shape[1]) and fcs[p].weight.is_contiguous() and fcs[p].weight.shape[0] == Fd
        a.ig[p], a.hg[p], a.bias[p] = None, None, biases[p].data_ptr()
        a.c_prev[p], a.h_out[p], a.c_out[p] = c_prev[p].data_ptr(), h_out[p].data_ptr(), c_out[p].data_ptr()
        a.acts[p] = acts[p].data_ptr() if acts is not None and acts[p] is not None else None
        assert acts is None or acts[p] is None or acts[p].is_contiguous()
        a.actor_w[p], a.actor_b[p] = actors[p].weight.data_ptr(), actors[p].bias.data_ptr()
        k.y[p], k.ldy[p], k.kfc[p] = ys[p].data_ptr(), ys[p].stride(0), ys[p].shape[1]
        k.fc_w[p], k.fc_b[p], k.w_cat[p] = fcs[p].weight.data_ptr(), fcs[p].bias.data_ptr(), w_cat[p].data_ptr()
    a.emb = emb.data_ptr() if emb is not None else None
    a.done_prev = done.data_ptr() if done is not None else None
    a.actions_out, a.counter, a.seed = actions_out.data_ptr(), sampler.counter.data_ptr(), sampler.seed
    a.ordinal = sampler._ordinal + 1
    sampler._ordinal += 2
    a.A, a.N, a.R = actors[0].weight.shape[0], N, R
    assert hm_out[0].shape == (N, R) and hm_out[1].shape == (N, R)
    assert hm_out[0].stride(1) == 1 and hm_out[1].stride(1) == 1 and hm_out[0].stride(0) == hm_out[1].stride(0)
    a.hm_out[0], a.hm_out[1], a.hm_ld = hm_out[0].data_ptr(), hm_out[1].data_ptr(), hm_out[0].stride(0)
    k.fh, k.gates, k.fh_pstride, k.fh_ld, k.F, k.workgroups = fh_t.data_ptr(), gates.data_ptr(), fh_t.stride(0), Fd + R, Fd, int(workgroups)
    k.probe = COOP_PROBE.data_ptr() if COOP_PROBE is not None else None
    obs, rew, done_out = env_out
    assert obs.is_contiguous() and rew.is_contiguous() and done_out.is_contiguous() and obs.dtype in (torch.uint8, torch.float32)
    L = lib()
    rc = L.atr_coop_env_step(env_core.h, C.byref(a), C.byref(k), _p(obs), 1 if obs.dtype == torch.uint8 else 0, _p(rew), _p(done_out),
                             _stream(gates))
    if rc != 0:
        raise RuntimeError("atr_coop_env_step failed (%d): %s" % (rc, L.t2d_last_error().decode()))
    return actions_out


def actor_step_supported(F, R):
    return F == 256 and R == 128


@torch.no_grad()
def actor_step_into(f, h_prev, c_prev, done, lstm, bias, h_out, c_out, acts, emb=None, act_in=None):
    """The actor's LSTMCell step as ONE MFMA kernel (csrc/actor_step_hip.hip): gates = f W_ih^T + (k h_prev) W_hh^T + bias
    [+ emb[act_in]] run straight into the cell; h', c' and the activated gates are written into the given (rollout
    cache) slots. f [N,256], h_prev / c_prev [N,128] contiguous; k = (done == 0)."""
    N, F = f.shape
    R = h_prev.shape[1]
    assert f.is_contiguous() and h_prev.is_contiguous() and c_prev.is_contiguous() and h_out.is_contiguous() and c_out.is_contiguous()
    rc = lib().atr_actor_step(_p(f), _p(h_prev), _p(c_prev), _pn(done), _p(lstm.weight_ih), _p(lstm.weight_hh), _p(bias),
                              _pn(emb), _pn(act_in), _p(h_out), _p(c_out), _pn(acts), N, F, R, _stream(f))
    if rc != 0:
        raise RuntimeError("atr_actor_step failed (%d)" % rc)
    return h_out, c_out


def lstm_sequence(ig0, ig1, whh, h0, c0, keep):
    return _LstmSeq.apply(ig0, ig1, whh, h0, c0, keep)


@torch.no_grad()
def gae_returns(rewards, values, notdone, gamma, tau):
    """rewards [T,N,A,1], values [T+1,N,A,1] (row T = bootstrap), notdone [T,N] float -> (returns, gae) [T,N,A,1]."""
    T, N, A = rewards.shape[0], rewards.shape[1], rewards.shape[2]
    rewards, values, notdone = rewards.contiguous(), values.contiguous(), notdone.contiguous()
    ret, gae = torch.empty_like(rewards), torch.empty_like(rewards)
    rc = lib().atr_gae_returns(_p(rewards), _p(values), _p(notdone), float(gamma), float(tau), _p(ret), _p(gae),
                               T, N, A, _stream(rewards))
    if rc != 0:
        raise RuntimeError("atr_gae_returns failed (%d)" % rc)
    return ret, gae


@torch.no_grad()
def heads_values(h, critic, values, off):
    """values[..., off] = critic(h) for h [rows, R] (csrc/heads_hip.hip); values: contiguous [..., A] float32 whose
    leading dims flatten to >= rows."""
    h = h.contiguous()
    rows, R = h.shape
    A = values.shape[-2] if values.shape[-1] == 1 else values.shape[-1]
    rc = lib().atr_heads_values(_p(h), _p(critic.weight), _p(critic.bias), _p(values), rows, R, A, off, _stream(h))
    if rc != 0:
        raise RuntimeError("atr_heads_values failed (%d)" % rc)


@torch.no_grad()
def heads_values2(hs, critics, values):
    """heads_values for both players in one launch: values[..., p] = critics[p](hs[p]) for p = 0, 1."""
    h0, h1 = hs[0].contiguous(), hs[1].contiguous()
    rows, R = h0.shape
    assert h1.shape == h0.shape
    A = values.shape[-2] if values.shape[-1] == 1 else values.shape[-1]
    rc = lib().atr_heads_values2(_p(h0), _p(critics[0].weight), _p(critics[0].bias), 0, _p(h1), _p(critics[1].weight),
                                 _p(critics[1].bias), 1, _p(values), rows, R, A, _stream(h0))
    if rc != 0:
        raise RuntimeError("atr_heads_values2 failed (%d)" % rc)


def _act_layout(actions, rows):
    """(tensor whose data_ptr is the first action, act_n, act_tstride) for actions given as a flat [rows] vector or as a
    [T, N] view whose rows are contiguous (one player's column of the rollout's [T, players, N] store): read in place."""
    if actions.dim() == 2 and actions.stride(1) == 1 and actions.shape[0] * actions.shape[1] == rows:
        return actions, actions.shape[1], actions.stride(0)
    a = actions.reshape(rows).contiguous()
    return a, rows, 0


class _HeadsLossPair(torch.autograd.Function):
    """Both players' heads + A3C loss terms over all stored steps as ONE autograd node and one launch + one reduction launch
    (atr_heads_loss_multi): forward already produces dL/dh and the head-parameter gradients for coefficient 1; backward hands
    them out (the two outputs must enter the objective with coefficient 1 each — Agent._loss_fused_heads differentiates
    their sum). Inputs per player: h, actor w/b, critic w/b, aux w/b (None). Outputs: (term0, term1, stats [2, 4])."""

    @staticmethod
    def forward(ctx, cfg, *t):
        L = lib()
        arr = (HeadsLossArgs * 2)()
        hs, keep = [], []
        for p in range(2):
            h, wa, ba, wc, bc, waux, baux = t[7 * p:7 * p + 7]
            c = cfg[p]
            h = h.contiguous()
            rows, R = h.shape
            A = wa.shape[0]
            rec = (A + 2) * R + (A + 2) + 4
            dh = torch.empty_like(h)
            gs = torch.empty(rec + 1, dtype=torch.float32, device=h.device)
            ws = torch.empty(L.atr_heads_workspace_floats(rows, R, A), dtype=torch.float32, device=h.device)
            acts, act_n, act_ts = _act_layout(c["actions"], rows)
            ret = c["ret"]
            a = arr[p]
            a.h, a.actions, a.act_n, a.act_tstride = h.data_ptr(), acts.data_ptr(), act_n, act_ts
            a.ret, a.gae, a.val = ret.data_ptr(), c["gae"].data_ptr(), c["val"].data_ptr()
            a.stride = ret.shape[-2] if ret.shape[-1] == 1 else ret.shape[-1]
            a.off = c["off"]
            r_aux = c.get("r_aux")
            a.r_aux = r_aux.data_ptr() if r_aux is not None else None
            a.aux_stride = (r_aux.shape[-2] if r_aux.shape[-1] == 1 else r_aux.shape[-1]) if r_aux is not None else 0
            a.aux_off = c.get("aux_off", 0)
            a.wa, a.ba, a.wc = wa.data_ptr(), ba.data_ptr(), wc.data_ptr()
            a.waux = waux.data_ptr() if waux is not None else None
            a.baux = baux.data_ptr() if baux is not None else None
            a.scale, a.scale_aux, a.w_ent = float(c["scale"]), float(c["scale_aux"]), float(c["w_ent"])
            a.dh, a.grads_and_sums, a.workspace = dh.data_ptr(), gs.data_ptr(), ws.data_ptr()
            a.rows, a.R, a.A = rows, R, A
            hs.append((dh, gs, A, R, waux is not None, rec))
            keep += [h, acts, ws]
        stats = torch.empty((2, 4), dtype=torch.float32, device=hs[0][0].device)
        arr[0].stats_out, arr[1].stats_out = stats.data_ptr(), stats.data_ptr() + 16
        rc = L.atr_heads_loss_multi(arr, 2, float(cfg[0]["stats_scale"]), _stream(hs[0][0]))
        if rc != 0:
            raise RuntimeError("atr_heads_loss_multi failed (%d)" % rc)
        ctx.save_for_backward(hs[0][0], hs[0][1], hs[1][0], hs[1][1])
        ctx.dims = [(x[2], x[3], x[4]) for x in hs]
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(stats)
        return hs[0][1][hs[0][5]], hs[1][1][hs[1][5]], stats

    @staticmethod
    def backward(ctx, g0, g1, gstats):
        sv = ctx.saved_tensors
        out = [None]
        for p in range(2):
            dh, gs = sv[2 * p], sv[2 * p + 1]
            A, R, has_aux = ctx.dims[p]
            if (g0, g1)[p] is None:                 # this player's term is not part of the differentiated objective
                out += [None] * 7
                continue
            o = (A + 2) * R
            dwa, dwc, dwx = gs[:A * R].view(A, R), gs[A * R:(A + 1) * R].view(1, R), gs[(A + 1) * R:o].view(1, R)
            dba, dbc, dbx = gs[o:o + A], gs[o + A:o + A + 1], gs[o + A + 1:o + A + 2]
            out += [dh, dwa, dba, dwc, dbc, dwx if has_aux else None, dbx if has_aux else None]
        return tuple(out)


def heads_loss_pair(hs, actors, critics, auxes, cfg):
    """-> (term0, term1, stats [2, 4]); cfg: per player dict(actions, ret, gae, val, off, r_aux, aux_off, scale, scale_aux, w_ent,
    stats_scale). The terms must be summed into the objective with coefficient 1 (their gradients are handed out as they are)."""
    t = []
    for p in range(2):
        aux = auxes[p]
        t += [hs[p], actors[p].weight, actors[p].bias, critics[p].weight, critics[p].bias,
              aux.weight if aux is not None else None, aux.bias if aux is not None else None]
    return _HeadsLossPair.apply(cfg, *t)


class _HeadsLoss(torch.autograd.Function):
    """One player's heads + A3C loss terms over all stored steps as ONE autograd node (csrc/heads_hip.hip): forward
    launches the fused kernel, which already produces dL/dh and the head-parameter gradients; backward hands them
    out. The output must enter the objective with coefficient 1 (Agent.loss_recompute adds the players' terms)."""

    @staticmethod
    def forward(ctx, h, wa, ba, wc, bc, waux, baux, actions, ret, gae, val, off, r_aux, aux_off, scale, scale_aux, w_ent,
                unit_coeff=False):
        L = lib()
        h = h.contiguous()
        rows, R = h.shape
        A = wa.shape[0]
        stride = ret.shape[-2] if ret.shape[-1] == 1 else ret.shape[-1]
        rec = (A + 2) * R + (A + 2) + 4
        dh = torch.empty_like(h)
        gs = torch.empty(rec + 1, dtype=torch.float32, device=h.device)
        ws = torch.empty(L.atr_heads_workspace_floats(rows, R, A), dtype=torch.float32, device=h.device)
        actions = actions.contiguous()
        aux_stride = 0
        if r_aux is not None:
            aux_stride = r_aux.shape[-2] if r_aux.shape[-1] == 1 else r_aux.shape[-1]
        rc = L.atr_heads_loss(_p(h), _p(actions), _p(ret), _p(gae), _p(val), stride, off, _pn(r_aux), aux_stride, aux_off,
                              _p(wa), _p(ba), _p(wc), _pn(waux), _pn(baux), float(scale), float(scale_aux), float(w_ent),
                              _p(dh), _p(gs), _p(ws), rows, R, A, _stream(h))
        if rc != 0:
            raise RuntimeError("atr_heads_loss failed (%d)" % rc)
        ctx.save_for_backward(dh, gs)
        ctx.dims = (A, R, waux is not None)
        ctx.unit_coeff = bool(unit_coeff)
        stats = gs[rec - 4:rec]
        ctx.mark_non_differentiable(stats)
        return gs[rec], stats

    @staticmethod
    def backward(ctx, gloss, gstats):
        dh, gs = ctx.saved_tensors
        A, R, has_aux = ctx.dims
        o = (A + 2) * R
        if not ctx.unit_coeff:          # the kernel's gradients are those of coefficient 1: apply the incoming one
            dh, gs = dh * gloss, gs * gloss
        dwa, dwc, dwx = gs[:A * R].view(A, R), gs[A * R:(A + 1) * R].view(1, R), gs[(A + 1) * R:o].view(1, R)
        dba, dbc, dbx = gs[o:o + A], gs[o + A:o + A + 1], gs[o + A + 1:o + A + 2]
        return (dh, dwa, dba, dwc, dbc, dwx if has_aux else None, dbx if has_aux else None) + (None,) * 11


def heads_loss(h, actor, critic, aux, actions, ret, gae, val, off, r_aux, aux_off, scale, scale_aux, w_ent,
               unit_coeff=False):
    """-> (objective contribution (0-dim, differentiable w.r.t. h and the head parameters),
           stats [4] = unscaled sums: policy term, value term, entropy, |aux error|).
    unit_coeff=True promises that the contribution enters the differentiated objective with coefficient exactly 1
    (Agent._loss_fused_heads adds the players' terms and differentiates the sum): backward then hands out the
    kernel's gradients as they are instead of scaling them by the incoming gradient."""
    return _HeadsLoss.apply(h, actor.weight, actor.bias, critic.weight, critic.bias,
                            aux.weight if aux is not None else None, aux.bias if aux is not None else None,
                            actions, ret, gae, val, off, r_aux, aux_off, scale, scale_aux, w_ent, unit_coeff)


class _EmbedAdd(torch.autograd.Function):
    """f + fc_action_tracker(one_hot(actions)) as a row gather (csrc/driver_hip.hip): forward one launch, backward the
    per-action column sums of the incoming gradient (two launches); the gradient w.r.t. f is the incoming one."""

    @staticmethod
    def forward(ctx, f, w, b, actions):
        if not (f.stride(1) == 1 and f.stride(0) >= f.shape[1] and f.stride(0) % 4 == 0):
            f = f.contiguous()       # (rows may be strided: the features as a column block of the rollout's [features | k h] rows)
        rows, Cc = f.shape
        A = w.shape[1]
        wc, bc = w.contiguous(), b.contiguous()
        out = torch.empty((rows, Cc), dtype=f.dtype, device=f.device)
        acts, act_n, act_ts = _act_layout(actions, rows)
        rc = lib().atr_embed_add_ld(_p(f), f.stride(0), _p(wc), _p(bc), _p(acts), 1, act_n, act_ts, _p(out), rows, Cc, A,
                                    _stream(f))
        if rc != 0:
            raise RuntimeError("atr_embed_add failed (%d)" % rc)
        ctx.save_for_backward(acts)
        ctx.dims = (Cc, A, act_n, act_ts)
        return out

    @staticmethod
    def backward(ctx, dout):
        (acts,) = ctx.saved_tensors
        Cc, A, act_n, act_ts = ctx.dims
        dout = dout.contiguous()
        rows = dout.shape[0]
        L = lib()
        ws = torch.empty(L.atr_embed_grad_workspace_floats(rows, Cc, A), dtype=torch.float32, device=dout.device)
        dw = torch.empty((Cc, A), dtype=torch.float32, device=dout.device)
        db = torch.empty(Cc, dtype=torch.float32, device=dout.device)
        rc = L.atr_embed_grad(_p(dout), _p(acts), 1, act_n, act_ts, _p(dw), _p(db), _p(ws), rows, Cc, A, _stream(dout))
        if rc != 0:
            raise RuntimeError("atr_embed_grad failed (%d)" % rc)
        return dout, dw, db, None


def embed_add(f, linear, actions):
    """f [rows, C] + linear(one_hot(actions, A)) for linear = nn.Linear(A, C); actions int64: a flat [rows] vector, or a [T, N]
    view with contiguous rows (one player's column of the rollout's [T, players, N] store, read in place)."""
    assert actions.dtype == torch.int64 and f.dim() == 2 and actions.numel() == f.shape[0]
    return _EmbedAdd.apply(f, linear.weight, linear.bias, actions)


use_gemm_tn = True
use_grouped_dw = True       # the learner's weight-gradient GEMMs of one backward pass as ONE grouped launch (DeferredWeightGrads)
_deferred = None            # the open DeferredWeightGrads of the backward pass in flight, if any


class DeferredWeightGrads(object):
    """The weight-gradient GEMMs of one backward pass, collected and launched together (atr_gemm_tn_grouped).

    Each cached-forward autograd node used to launch its own split-K GEMM, reduction and bias column-sum as it ran: 16
    launches for six products that share K = T*N rows, the small ones split into 128 K-slices to fill the chip on their own.
    Opened around torch.autograd.grad by Agent.compute_grads, this object lets those nodes only REGISTER their product
    (operands, the parameter it belongs to) and hands them the parameter's slice of the flat gradient bucket as the gradient
    they return — filled when flush() launches all products as one grouped GEMM + one reduction that writes straight into
    those slices (so FlatParams.set_grads has nothing to copy for them). Operands are kept alive until the flush."""

    MAX = 8

    def __init__(self, bucket):
        self.views = {}
        for prm, v in zip(bucket.params, bucket.grad_views()):
            self.views[prm.data_ptr()] = v
        self.problems, self.K, self.registered = [], None, set()
        self.after = []           # launches that need the group's outputs in place (run at the end of flush(), in order)

    def view(self, prm):
        return self.views.get(prm.data_ptr()) if prm is not None else None

    def add(self, x1, x2, weight, biases=(), row_scale=None, shift=0):
        """Register dW = x1^T x2 -> `weight`'s gradient slice, colsum(x1) -> the slices of `biases` (<= 2). Returns
        (dW view, [bias views]) or None when this product cannot join the group (shape class, parameter not in the bucket, a
        different K, group full): the caller then computes it on the spot."""
        K, M = x1.shape
        N = x2.shape[1]
        dst = self.view(weight)
        bv = [self.view(b) for b in biases]
        if (dst is None or any(v is None for v in bv) or len(bv) > 2 or len(self.problems) >= self.MAX
                or (self.K is not None and K != self.K) or K < 4096 or M % 128 or N % 128
                or not (x1.is_cuda and x1.dtype == torch.float32 and x2.dtype == torch.float32
                        and x1.is_contiguous() and x2.stride(1) == 1 and x2.stride(0) >= N and x2.stride(0) % 4 == 0
                        and x2.data_ptr() % 16 == 0)
                or tuple(dst.shape) != (M, N) or (row_scale is not None and not row_scale.is_contiguous())):
            return None
        self.K = K
        self.problems.append((x1, x2, dst, row_scale, int(shift), bv, M, N))
        self.registered.update(v.data_ptr() for v in [dst] + bv)
        return dst, bv

    def check(self, grads):
        """The gradients autograd handed back for the registered parameters must BE the bucket slices the nodes returned (a
        parameter feeding two nodes would make autograd sum into a fresh tensor before the slice is filled)."""
        got = set(g.data_ptr() for g in grads if g is not None)
        if not self.registered <= got:
            raise RuntimeError("grouped weight gradients: autograd did not return the registered bucket slices as they are "
                               "(a parameter with more than one gradient contribution?); set fused.use_grouped_dw = False")

    @torch.no_grad()
    def flush(self):
        if not self.problems:
            return
        n = len(self.problems)
        arr = (GemmTnProblem * n)()
        for q, (x1, x2, dst, rs, shift, bv, M, N) in enumerate(self.problems):
            arr[q].x1, arr[q].x2, arr[q].c = x1.data_ptr(), x2.data_ptr(), dst.data_ptr()
            arr[q].row_scale = rs.data_ptr() if rs is not None else None
            arr[q].row_scale_shift = shift
            arr[q].colsum0 = bv[0].data_ptr() if len(bv) > 0 else None
            arr[q].colsum1 = bv[1].data_ptr() if len(bv) > 1 else None
            arr[q].M, arr[q].N = M, N
            arr[q].ld1, arr[q].ld2 = M, x2.stride(0)
        L = lib()
        x0 = self.problems[0][0]
        ws = torch.empty(L.atr_gemm_tn_grouped_workspace_floats(arr, n, self.K), dtype=torch.float32, device=x0.device)
        rc = L.atr_gemm_tn_grouped(arr, n, self.K, _p(ws), _stream(x0))
        if rc != 0:
            raise RuntimeError("atr_gemm_tn_grouped failed (%d)" % rc)
        self.problems, self.K = [], None
        for fn in self.after:
            fn()
        self.after = []


class deferred_weight_grads(object):
    """with deferred_weight_grads(bucket) as q: grads = torch.autograd.grad(...); q.flush()"""

    def __init__(self, bucket):
        self.bucket = bucket

    def __enter__(self):
        global _deferred
        self.q = DeferredWeightGrads(self.bucket) if (use_grouped_dw and use_gemm_tn and self.bucket.grad.is_cuda) else None
        _deferred = self.q
        return self.q

    def __exit__(self, *exc):
        global _deferred
        _deferred = None
        return False



class gemm_tn_corun(object):
    """`with fused.gemm_tn_corun(True):` — the weight-gradient launches issued (or captured) inside are planned for ONE workgroup
    per CU (atr_gemm_tn_set_corun, csrc/gemm_tn_hip.hip): ~1.4x the kernel's own time, but a chain of short kernels on another
    stream keeps running beside it. The pipelined schedule captures its learner graphs like this (train.PipelinedIteration)."""

    def __init__(self, on=True):
        self.on, self.was = bool(on), None

    def __enter__(self):
        self.was = lib().atr_gemm_tn_set_corun(1 if self.on else 0)
        return self

    def __exit__(self, *exc):
        lib().atr_gemm_tn_set_corun(self.was)
        return False


@torch.no_grad()
def gemm_tn(x1, x2, row_scale=None, colsum=False):
    """x1.t() @ x2 for tall row-major x1 [K,M], x2 [K,N] — the weight-gradient GEMMs (csrc/gemm_tn_hip.hip) when the
    shape fits the kernel (CUDA fp32, contiguous, M and N multiples of 128), otherwise the library GEMM.
    row_scale [K]: x1's rows are multiplied by it first; colsum=True: also return (scaled x1).sum(0) — the bias
    gradient that goes with the weight gradient — from the same pass."""
    K, M = x1.shape
    N = x2.shape[1]
    if (use_gemm_tn and x1.is_cuda and x1.dtype == torch.float32 and x2.dtype == torch.float32 and M % 128 == 0
            and N % 128 == 0 and K >= 4096 and x1.is_contiguous() and x2.is_contiguous()):
        L = lib()
        ws = torch.empty(L.atr_gemm_tn_workspace_floats(K, M, N), dtype=torch.float32, device=x1.device)
        c = torch.empty((M, N), dtype=torch.float32, device=x1.device)
        cs = torch.empty(M, dtype=torch.float32, device=x1.device) if colsum else None
        rs = row_scale.contiguous() if row_scale is not None else None
        rc = L.atr_gemm_tn(_p(x1), _p(x2), _p(c), _p(ws), K, M, N, _pn(rs), _pn(cs), _stream(x1))
        if rc != 0:
            raise RuntimeError("atr_gemm_tn failed (%d)" % rc)
        return (c, cs) if colsum else c
    if row_scale is not None:
        x1 = x1 * row_scale.unsqueeze(1)
    c = x1.t() @ x2
    return (c, x1.sum(0)) if colsum else c


class RolloutConsts(C.Structure):
    """atr_rollout_consts of include/atr_policy.h."""
    _fields_ = [("w_ih", C.c_void_p * 2), ("w_hh", C.c_void_p * 2), ("b_ih", C.c_void_p * 2), ("b_hh", C.c_void_p * 2),
                ("bsum", C.c_void_p), ("w_cat", C.c_void_p), ("fa_w", C.c_void_p), ("fa_b", C.c_void_p), ("emb_ih", C.c_void_p),
                ("counter", C.c_void_p), ("fh0", C.c_void_p), ("fh_pstride", C.c_longlong), ("fh_ld", C.c_longlong),
                ("F", C.c_int), ("A_act", C.c_int)]


def rollout_consts_ok(consts):
    """Whether atr_rollout_begin2 can make these constants (contiguous, 16-byte aligned parameters)."""
    ts = []
    for l in consts["lstm"]:
        ts += [l.weight_ih, l.weight_hh, l.bias_ih, l.bias_hh]
    if "fa" in consts:
        ts += [consts["fa"].weight, consts["fa"].bias]
    return all(t.is_cuda and t.is_contiguous() and t.dtype == torch.float32 and t.data_ptr() % 16 == 0 for t in ts)


@torch.no_grad()
def rollout_begin(hxs, cxs, h_all, c_all, obs_src=None, obs_dst=None, consts=None, counter=None):
    """hxs/cxs [N,A,R] -> h_all/c_all[:, 0] ([A,T+1,N,R] rollout stores); optionally obs_src -> obs_dst (same bytes).
    consts (model.new_cache(defer_consts=True).consts) + counter (the action sampler's): the per-rollout constants of the actor
    and the counter bump in the same launch (atr_rollout_begin2)."""
    N, A, R = hxs.shape
    assert hxs.is_contiguous() and cxs.is_contiguous() and h_all.is_contiguous() and c_all.is_contiguous()
    assert h_all.shape[0] == A and h_all.shape[2] == N and h_all.shape[3] == R and hxs.dtype == torch.float32
    nbytes = 0
    if obs_src is not None:
        assert obs_src.is_contiguous() and obs_dst.is_contiguous()
        nbytes = obs_src.numel() * obs_src.element_size()
        assert nbytes == obs_dst.numel() * obs_dst.element_size() and nbytes % 4 == 0
    if consts is None:
        rc = lib().atr_rollout_begin(_p(hxs), _p(cxs), _p(h_all), _p(c_all), h_all.stride(0), _pn(obs_src), _pn(obs_dst), nbytes,
                                     N, A, R, _stream(hxs))
        if rc != 0:
            raise RuntimeError("atr_rollout_begin failed (%d)" % rc)
        return
    k = RolloutConsts()
    for p, l in enumerate(consts["lstm"]):
        k.w_ih[p], k.w_hh[p] = l.weight_ih.data_ptr(), l.weight_hh.data_ptr()
        k.b_ih[p], k.b_hh[p] = l.bias_ih.data_ptr(), l.bias_hh.data_ptr()
    k.bsum = consts["bsum"].data_ptr()
    k.F = int(consts["F"])
    if "w_cat" in consts:
        k.w_cat = consts["w_cat"].data_ptr()
        fh = consts["fh_all"]
        k.fh0, k.fh_pstride, k.fh_ld = fh.data_ptr(), fh.stride(0), fh.stride(2)
    if "emb_ih" in consts:
        fa = consts["fa"]
        k.fa_w, k.fa_b, k.emb_ih, k.A_act = fa.weight.data_ptr(), fa.bias.data_ptr(), consts["emb_ih"].data_ptr(), fa.weight.shape[1]
    if counter is not None:
        k.counter = counter.data_ptr()
    L = lib()
    L.atr_rollout_begin2.restype = C.c_int
    L.atr_rollout_begin2.argtypes = [C.c_void_p] * 4 + [C.c_longlong, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_int,
                                                        C.c_int, C.POINTER(RolloutConsts), C.c_void_p]
    rc = L.atr_rollout_begin2(_p(hxs), _p(cxs), _p(h_all), _p(c_all), h_all.stride(0), _pn(obs_src), _pn(obs_dst), nbytes,
                              N, A, R, C.byref(k), _stream(hxs))
    if rc != 0:
        raise RuntimeError("atr_rollout_begin2 failed (%d)" % rc)


@torch.no_grad()
def rollout_end(h_all, c_all, dones, hxs, cxs, eps_len, keep, obs_src=None, obs_dst=None, done_dst=None):
    """Slot T of h_all/c_all [A,T+1,N,R], masked by dones[T-1], -> hxs/cxs [N,A,R] (in place); eps_len [N] int32 advanced
    over the rollout's dones [T,N] uint8 (in place); keep [T,N] float32 = (dones == 0). Optionally, in the same launch:
    obs_src -> obs_dst (same bytes: the observation the next rollout starts from) and dones[T-1] -> done_dst [N] uint8."""
    A, T1, N, R = h_all.shape
    T = T1 - 1
    assert dones.shape == (T, N) and dones.dtype == torch.uint8 and dones.is_contiguous()
    assert hxs.shape == (N, A, R) and hxs.is_contiguous() and cxs.is_contiguous() and keep.is_contiguous()
    assert eps_len.dtype == torch.int32 and eps_len.is_contiguous() and keep.shape == (T, N)
    nbytes = 0
    if obs_src is not None:
        assert obs_src.is_contiguous() and obs_dst.is_contiguous() and obs_src.dtype == obs_dst.dtype
        nbytes = obs_src.numel() * obs_src.element_size()
        assert nbytes == obs_dst.numel() * obs_dst.element_size() and nbytes % 4 == 0
    if done_dst is not None:
        assert done_dst.dtype == torch.uint8 and done_dst.is_contiguous() and done_dst.numel() == N
    rc = lib().atr_rollout_end2(_p(h_all[0, T]), _p(c_all[0, T]), h_all.stride(0), _p(dones), _p(hxs), _p(cxs), _p(eps_len),
                                _p(keep), T, N, A, R, _pn(obs_src), _pn(obs_dst), nbytes, _pn(done_dst), _stream(hxs))
    if rc != 0:
        raise RuntimeError("atr_rollout_end failed (%d)" % rc)


@torch.no_grad()
def adam_step(p, grad, exp_avg, exp_avg_sq, max_exp_avg_sq, scalars, step_size, lr, beta1, beta2, eps, weight_decay,
              torch_eps=False):
    """SharedAdam.step over the flat bucket (csrc/driver_hip.hip): scalars = [step, beta1^step, beta2^step] float64;
    step_size: two floats of scratch. torch_eps: torch.optim.Adam's placement of eps and the bias corrections."""
    rc = lib().atr_adam_step(_p(p), _p(grad), _p(exp_avg), _p(exp_avg_sq), _pn(max_exp_avg_sq), _p(scalars), _p(step_size),
                             float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), 1 if torch_eps else 0,
                             p.numel(), _stream(p))
    if rc != 0:
        raise RuntimeError("atr_adam_step failed (%d)" % rc)


@torch.no_grad()
def rmsprop_step(p, grad, square_avg, lr, alpha, eps, weight_decay):
    """SharedRMSprop.step (momentum 0, not centered) over the flat bucket (csrc/driver_hip.hip)."""
    rc = lib().atr_rmsprop_step(_p(p), _p(grad), _p(square_avg), float(lr), float(alpha), float(eps), float(weight_decay),
                                p.numel(), _stream(p))
    if rc != 0:
        raise RuntimeError("atr_rmsprop_step failed (%d)" % rc)
