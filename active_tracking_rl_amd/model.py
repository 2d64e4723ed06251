"""Batched restatement of the reference policy networks (stays PyTorch-ROCm, per the north star).

Mirrors model.py:12-265 and perception.py:68-92 of the reference with IDENTICAL module / parameter names and
shapes (state-dict contract), so checkpoints move both ways:
  player{0,1}.encoder.conv1/conv2/fc, .lstm (LSTMCell 256->128), .actor.actor_linear, .critic.critic_linear,
  and for `tat` heads player1.fc_action_tracker, player1.reward_aux.

The reference networks are structurally batch-1 (`x.view(1, -1)`, perception.py:89). Here every env is a row:
  states  [N, A=2, stack, C=1, 13, 13]   (or the reference layout [A, stack, C, 13, 13] for one env)
  hx, cx  [N, A, rnn_out]                 (or [A, rnn_out])
and `view(N, -1)` keeps the per-env feature order of the reference, including the
[tracker frame, target frame] concatenation the TAT target uses (model.py:255).
Only the discrete maze heads are built ('maze' encoders, 'lstm' core); the continuous / Unreal image heads
(CNN_simple, ICML, GRU) are outside the hot-path scope.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# ---- initialisers ---------------------------------------------------------------------------------------
# Same numerics and the same consumption of torch's global RNG as utils.py:30-33,47-72 of the reference (one
# randn / uniform_ / normal_ of the weight's shape per call, in the same order), so seeded inits agree.
def _glorot_bound(fan_in, fan_out):
    return float(np.sqrt(6.0 / (fan_in + fan_out)))


def norm_col_init(weights, std=1.0):
    """Gaussian rows rescaled to Euclidean norm `std` (utils.py:30-33)."""
    g = torch.randn(weights.size())
    return g * (std / g.pow(2).sum(1, keepdim=True).sqrt())


@torch.no_grad()
def weights_init(m):
    """Glorot-uniform weights, zero biases for every Conv* / *Linear* module (utils.py:47-62). For a conv kernel
    [O, I, kh, kw] the fans are I*kh*kw and O*kh*kw; for a linear layer [O, I] they are I and O."""
    kind = type(m).__name__
    if 'Conv' in kind:
        o, i, kh, kw = m.weight.shape
        bound = _glorot_bound(i * kh * kw, o * kh * kw)
    elif 'Linear' in kind:
        o, i = m.weight.shape
        bound = _glorot_bound(i, o)
    else:
        return
    m.weight.uniform_(-bound, bound)
    m.bias.zero_()


@torch.no_grad()
def weights_init_mlp(m):
    """Unit-norm Gaussian rows, zero bias, Linear modules only (utils.py:65-72)."""
    if 'Linear' not in type(m).__name__:
        return
    m.weight.normal_(0, 1)
    m.weight.mul_(m.weight.pow(2).sum(1, keepdim=True).sqrt().reciprocal())
    if m.bias is not None:
        m.bias.zero_()


def build_model(obs_space, action_space, args, device):
    """Same signature as the reference build_model (model.py:12-15)."""
    model = A3C_Dueling(obs_space, action_space, args, device)
    model.train()
    return model


def sample_action(logit, test=False):
    """Discrete branch of sample_action (model.py:41-50), batched: logit [N, n_actions].
    Returns action int64 [N] (stays on the device), entropy [N,1], log_prob [N,1] (train) or [N,n] (test)."""
    prob = F.softmax(logit, dim=1)
    log_prob = F.log_softmax(logit, dim=1)
    entropy = -(log_prob * prob).sum(1, keepdim=True)
    if test:
        action = prob.max(1)[1]
    else:
        action = prob.multinomial(1).squeeze(1)
        log_prob = log_prob.gather(1, action.unsqueeze(1))
    return action, entropy, log_prob


fused_lstm = True   # GPU tensors: recurrence through the fused HIP cell kernels (csrc/lstm_hip.hip); False = ATen ops


def lstm_sequence_fused(lstms, feats, h, c, keep):
    """lstm_sequence on the fused kernels: feats = per-player list of [T, N, F] (no stacking copy). One autograd node
    for the whole recurrence (fused.lstm_sequence); returns (h_seq as a per-player list of [T, N, R], final masked h
    [P, N, R], final masked c [P, N, R] (c not differentiable))."""
    from . import fused
    T, N = feats[0].shape[0], feats[0].shape[1]
    igs = [F.linear(f.reshape(T * N, -1), l.weight_ih, l.bias_ih + l.bias_hh) for f, l in zip(feats, lstms)]
    whh = torch.stack([l.weight_hh.t() for l in lstms], 0)
    outs = fused.lstm_sequence(igs[0], igs[1] if len(igs) > 1 else None, whh, h.contiguous(), c.contiguous(), keep)
    h_seq, k_last = list(outs[:-1]), keep[-1].view(1, N, 1)
    return h_seq, torch.stack([o[-1] for o in h_seq], 0) * k_last, outs[-1] * k_last


def lstm_sequence(lstms, feats, h, c, keep):
    """Run P independent nn.LSTMCells (same sizes, different weights — the two players) over time-major feature
    sequences in lock step, with the per-step episode-boundary mask applied AFTER each step (what Agent.action_train
    does). feats [T, P, N, F], h/c [P, N, R], keep [T, N] (0 where the env finished at that step).
    The input projections are one GEMM per player over all T*N rows (both biases folded in); per step the P hidden
    GEMMs are ONE bmm and the pointwise cell ONE fused kernel over P*N rows — the recurrence is launch-bound, so
    batching the players halves its cost. Returns h_seq [T, P, N, R] and the final masked (h, c) [P, N, R].
    feats may also be a per-player list of [T, N, F]: on the GPU that selects the fused HIP recurrence
    (lstm_sequence_fused; h_seq then comes back as a per-player list of [T, N, R]) without a stacking copy."""
    if isinstance(feats, (list, tuple)):
        f0 = feats[0]
        if f0.is_cuda and fused_lstm and len(feats) <= 2 and h.shape[-1] % 4 == 0 and f0.dtype == torch.float32:
            return lstm_sequence_fused(lstms, list(feats), h, c, keep)
        feats = torch.stack(list(feats), 1)
    T, P, N = feats.shape[0], feats.shape[1], feats.shape[2]
    R = h.shape[-1]
    ig = torch.stack([F.linear(feats[:, p].reshape(T * N, -1), l.weight_ih, l.bias_ih + l.bias_hh).view(T, N, -1)
                      for p, l in enumerate(lstms)], 1)                       # [T, P, N, 4R]
    # unbind (not ig[t]): its backward is one stack, not T zero-filled tensors added together
    igates = ig.reshape(T, P * N, 4 * R).unbind(0)
    whh = torch.stack([l.weight_hh.t() for l in lstms], 0)                     # [P, R, 4R]
    keepm = keep.unsqueeze(1).expand(T, P, N).reshape(T, P * N, 1).unbind(0)
    fused = feats.is_cuda and hasattr(torch.ops.aten, "_thnn_fused_lstm_cell")
    h, c = h.reshape(P * N, R), c.reshape(P * N, R)
    outs = []
    for t in range(T):
        hgates = torch.bmm(h.view(P, N, R), whh).view(P * N, 4 * R)
        if fused:
            h, c, _ = torch.ops.aten._thnn_fused_lstm_cell(igates[t], hgates, c)
        else:
            g = igates[t] + hgates
            i, f, gg, o = g.chunk(4, 1)
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
            h = torch.sigmoid(o) * torch.tanh(c)
        outs.append(h)
        h, c = h * keepm[t], c * keepm[t]
    return torch.stack(outs, 0).view(T, P, N, R), h.view(P, N, R), c.view(P, N, R)


def _player(h_seq, p):
    """Player p's [T, N, R] outputs from either form lstm_sequence returns."""
    return h_seq[p] if isinstance(h_seq, (list, tuple)) else h_seq[:, p]


def policy_stats(logit, action):
    """entropy and log-prob of given actions (the train branch of sample_action with the action fixed)."""
    prob = F.softmax(logit, dim=1)
    log_prob = F.log_softmax(logit, dim=1)
    entropy = -(log_prob * prob).sum(1, keepdim=True)
    return entropy, log_prob.gather(1, action.unsqueeze(1))


class ValueNet(nn.Module):
    def __init__(self, input_dim):
        super(ValueNet, self).__init__()
        self.critic_linear = nn.Linear(input_dim, 1)
        self.critic_linear.weight.data = norm_col_init(self.critic_linear.weight.data, 0.01)
        self.critic_linear.bias.data.fill_(0)

    def forward(self, x):
        return self.critic_linear(x)


class PolicyNet(nn.Module):
    def __init__(self, input_dim, action_space, head_name, device):
        super(PolicyNet, self).__init__()
        if 'continuous' in head_name:
            raise NotImplementedError("continuous heads belong to the Unreal envs (out of scope)")
        self.actor_linear = nn.Linear(input_dim, action_space.n)
        self.actor_linear.weight.data = norm_col_init(self.actor_linear.weight.data, 0.01)
        self.actor_linear.bias.data.fill_(0)

    def forward(self, x, test=False):
        return sample_action(self.actor_linear(x), test)


def _toeplitz_index(c_in, c_out, h_in, k, stride, pad):
    """Index table that expands a conv kernel [c_out, c_in, k, k] into the dense matrix of the same linear map
    on a fixed h_in x h_in input: rows (co, oh, ow), columns (ci, ih, iw); entries point into the flattened
    kernel, or to an appended zero slot where the tap falls outside the (zero-padded) input."""
    h_out = (h_in + 2 * pad - k) // stride + 1
    zero_slot = c_out * c_in * k * k
    idx = torch.full((c_out * h_out * h_out, c_in * h_in * h_in), zero_slot, dtype=torch.long)
    co = torch.arange(c_out).view(-1, 1, 1, 1, 1, 1)
    oh = torch.arange(h_out).view(1, -1, 1, 1, 1, 1)
    ow = torch.arange(h_out).view(1, 1, -1, 1, 1, 1)
    ci = torch.arange(c_in).view(1, 1, 1, -1, 1, 1)
    kh = torch.arange(k).view(1, 1, 1, 1, -1, 1)
    kw = torch.arange(k).view(1, 1, 1, 1, 1, -1)
    ih, iw = oh * stride - pad + kh, ow * stride - pad + kw
    ok = ((ih >= 0) & (ih < h_in) & (iw >= 0) & (iw < h_in)).expand(c_out, h_out, h_out, c_in, k, k)
    row = ((co * h_out + oh) * h_out + ow).expand_as(ok)
    col = ((ci * h_in + ih) * h_in + iw).expand_as(ok)
    src = (((co * c_in + ci) * k + kh) * k + kw).expand_as(ok)
    idx[row[ok], col[ok]] = src[ok]
    return idx, h_out


class _ToeplitzExpand(torch.autograd.Function):
    """dense = cat(w, 0)[idx]; the backward is a gather + sum over the (<= 49) places each tap is used, instead of
    autograd's generic index_put(accumulate) (a 3.8 ms sort-based kernel per call on MI355X)."""

    @staticmethod
    def forward(ctx, w, idx, inv):
        ctx.save_for_backward(inv)
        ctx.wshape = w.shape
        return torch.cat([w.reshape(-1), w.new_zeros(1)])[idx]

    @staticmethod
    def backward(ctx, g):
        (inv,) = ctx.saved_tensors
        gw = torch.cat([g.reshape(-1), g.new_zeros(1)])[inv].sum(1)
        return gw.view(ctx.wshape), None, None


def _inverse_index(idx, n_weights):
    """inv[j, :] = flat positions of idx that read weight j, padded with idx.numel() (-> the appended zero)."""
    flat = idx.reshape(-1)
    pos = torch.nonzero(flat < n_weights).squeeze(1)
    src = flat[pos]
    order = torch.argsort(src, stable=True)
    src, pos = src[order], pos[order]
    counts = torch.bincount(src, minlength=n_weights)
    width = int(counts.max().item())
    inv = torch.full((n_weights, width), flat.numel(), dtype=torch.long)
    start = torch.cumsum(counts, 0) - counts
    col = torch.arange(src.numel()) - start[src]
    inv[src, col] = pos
    return inv


class CNN_maze(nn.Module):
    """perception.py:68-92: conv(C->16,k3,s2,p1) 13->7, conv(16->32,k3,s2,p1) 7->4, fc 512*F->256, ReLUs.

    `stack_frames` frames per env are folded into the batch and unfolded into the feature axis, exactly what the
    reference's view(1, -1) does for one env. MI355X-first evaluation: on a fixed 13x13 input each conv is a small
    fixed linear map, so the two convs run as two plain fp32 GEMMs against Toeplitz-expanded weights (built from
    conv{1,2}.weight by one gather, differentiable) — no im2col, no per-sample MIOpen launches (MIOpen's GEMM path
    issues one Im2Col kernel PER SAMPLE: 393 216 launches for 100 steps of 4096 envs). Parameters keep the
    reference's names and shapes; outputs equal F.conv2d's up to fp32 summation order."""

    def __init__(self, obs_shape, stack_frames):
        super(CNN_maze, self).__init__()
        self.conv1 = nn.Conv2d(obs_shape[0], 16, 3, stride=2, padding=1)
        self.conv2 = nn.Conv2d(16, 32, 3, stride=2, padding=1)
        relu_gain = nn.init.calculate_gain('relu')
        self.conv1.weight.data.mul_(relu_gain)
        self.conv2.weight.data.mul_(relu_gain)
        assert obs_shape[1] == obs_shape[2], "square observations"
        # 13x13 crops ('Partial' ids): fused HIP stem on the GPU, Toeplitz GEMMs on the CPU. Whole-map observations
        # ('Full' ids, 81/82 wide) would need 10^8-entry expansion tables: they go through F.conv2d.
        self.small = obs_shape[1] <= 16
        h1 = (obs_shape[1] + 2 - 3) // 2 + 1
        h2 = (h1 + 2 - 3) // 2 + 1
        if self.small:
            idx1, _ = _toeplitz_index(obs_shape[0], 16, obs_shape[1], 3, 2, 1)
            idx2, _ = _toeplitz_index(16, 32, h1, 3, 2, 1)
            self.register_buffer("_idx1", idx1, persistent=False)
            self.register_buffer("_idx2", idx2, persistent=False)
            self.register_buffer("_inv1", _inverse_index(idx1, self.conv1.weight.numel()), persistent=False)
            self.register_buffer("_inv2", _inverse_index(idx2, self.conv2.weight.numel()), persistent=False)
        self._hw1, self._hw2 = h1 * h1, h2 * h2
        cnn_dim = 32 * self._hw2 * stack_frames
        self.fc = nn.Linear(cnn_dim, 256)
        self.outdim = 256
        self._dense = None
        self.apply(weights_init)
        self.train()

    def dense_weights(self):
        """(W1 [16*49, C*169], b1, W2 [32*16, 16*49], b2) — differentiable w.r.t. conv1/conv2 parameters."""
        def expand(conv, idx, inv, hw):
            return _ToeplitzExpand.apply(conv.weight, idx, inv), conv.bias.repeat_interleave(hw)
        W1, b1 = expand(self.conv1, self._idx1, self._inv1, self._hw1)
        W2, b2 = expand(self.conv2, self._idx2, self._inv2, self._hw2)
        return W1, b1, W2, b2

    def cache_dense(self, on=True):
        """Build the expanded weights once and reuse them for every forward until cache_dense(False): the rollout
        driver brackets the 20-step rollout + backward with it (weights only change at optimizer.step). Not needed
        (and skipped) when the fused HIP stem is in use."""
        fused_path = self.use_fused and self.conv1.weight.is_cuda and self.conv1.in_channels == 1
        self._dense = self.dense_weights() if (on and self.small and not fused_path) else None

    use_fused = True   # GPU tensors go through the fused HIP stem (csrc/stem_hip.hip); set False to force GEMMs

    def forward(self, x, fc=True):
        n, f = x.shape[0], x.shape[1]
        fused_ok = x.is_cuda and self.small and self.use_fused and self.conv1.in_channels == 1 and x.shape[-1] == 13
        if not x.is_floating_point() and not fused_ok:
            x = x.float()            # u8 observations: only the fused stem decodes them inside conv1
        if not self.small:
            return self.forward_conv2d(x, fc)
        if fused_ok:
            from . import fused
            x = fused.stem(x, self.conv1, self.conv2)       # strided views of the obs tensor are read in place
        else:
            x = self.forward_dense_stem(x.reshape(n * f, -1))
        x = x.reshape(n, -1)
        if fc:
            x = F.relu(self.fc(x))
        return x

    def forward_dense_stem(self, x):
        """The two convs as plain fp32 GEMMs against Toeplitz-expanded weights (CPU path and GPU cross-check)."""
        W1, b1, W2, b2 = self._dense if self._dense is not None else self.dense_weights()
        x = F.relu(F.linear(x, W1, b1))
        return F.relu(F.linear(x, W2, b2))

    def forward_conv2d(self, x, fc=True):
        """The same network through F.conv2d — the plain PyTorch fp32 reference used by the numerics tests."""
        n, f = x.shape[0], x.shape[1]
        x = x.reshape(n * f, x.shape[2], x.shape[3], x.shape[4])
        x = F.relu(self.conv1(x))
        x = F.relu(self.conv2(x))
        x = x.reshape(n, -1)
        if fc:
            x = F.relu(self.fc(x))
        return x


def _make_encoder(head_name, obs_space, stack_frames):
    if 'maze' not in head_name:
        raise NotImplementedError("only the 'maze' encoders (maze-lstm, tat-maze-lstm) are in scope; got %r" % head_name)
    return CNN_maze(obs_space, stack_frames)


class A3C(nn.Module):
    """Tracker network, model.py:102-145."""

    def __init__(self, obs_space, action_space, rnn_out=128, head_name='cnn_lstm', stack_frames=1, sub_task=False,
                 device=None):
        super(A3C, self).__init__()
        self.sub_task = sub_task
        self.head_name = head_name
        self.encoder = _make_encoder(head_name, obs_space, stack_frames)
        feature_dim = self.encoder.outdim
        if 'lstm' not in head_name:
            raise NotImplementedError("only LSTM cores are in scope")
        self.lstm = nn.LSTMCell(feature_dim, rnn_out)
        self.lstm.bias_ih.data.fill_(0)
        self.lstm.bias_hh.data.fill_(0)
        feature_dim = rnn_out
        self.actor = PolicyNet(feature_dim, action_space, head_name, device)
        self.critic = ValueNet(feature_dim)
        self.apply(weights_init)
        self.train()

    def forward(self, inputs, test=False):
        x, (hx, cx) = inputs
        feature = self.encoder(x)
        hx, cx = self.lstm(feature, (hx, cx))
        value = self.critic(hx)
        action, entropy, log_prob = self.actor(hx, test)
        return value, action, entropy, log_prob, (hx, cx)


    def sequence_features(self, x_seq):
        """Encoder over all T*N stored frames at once: x_seq [T, N, F, C, 13, 13] -> [T, N, 256]."""
        T, N = x_seq.shape[0], x_seq.shape[1]
        return self.encoder(x_seq.reshape(T * N, *x_seq.shape[2:])).view(T, N, -1)

    def sequence_heads(self, h_seq, actions):
        """Heads over all T*N hidden states at once: values, entropies, log-probs of the stored actions [T,N,1]."""
        T, N = h_seq.shape[0], h_seq.shape[1]
        flat = h_seq.reshape(T * N, -1)
        value = self.critic(flat)
        entropy, log_prob = policy_stats(self.actor.actor_linear(flat), actions.reshape(T * N))
        return value.view(T, N, 1), entropy.view(T, N, 1), log_prob.view(T, N, 1)

    def forward_sequence(self, x_seq, actions, h, c, keep):
        """Time-batched re-evaluation of T stored steps for this player alone (same math as T calls of forward())."""
        feats = self.sequence_features(x_seq)
        h_seq, h, c = lstm_sequence([self.lstm], [feats], h.unsqueeze(0), c.unsqueeze(0), keep)
        return self.sequence_heads(_player(h_seq, 0), actions) + ((h[0], c[0]),)


class TAT(nn.Module):
    """Tracker-aware target, model.py:148-209."""

    def __init__(self, obs_space, action_space, rnn_out=128, head_name='cnn_lstm', stack_frames=1,
                 dim_action_tracker=-1, device=None):
        super(TAT, self).__init__()
        self.sub_task = dim_action_tracker > 0
        self.head_name = head_name
        self.encoder = _make_encoder(head_name, obs_space, stack_frames)
        feature_dim = self.encoder.outdim
        if 'lstm' not in head_name:
            raise NotImplementedError("only LSTM cores are in scope")
        self.lstm = nn.LSTMCell(feature_dim, rnn_out)
        self.lstm.bias_ih.data.fill_(0)
        self.lstm.bias_hh.data.fill_(0)
        feature_dim = rnn_out
        self.actor = PolicyNet(feature_dim, action_space, head_name, device)
        self.critic = ValueNet(feature_dim)
        self.fc_action_tracker = nn.Linear(dim_action_tracker, self.encoder.outdim)
        weights_init_mlp(self.fc_action_tracker)
        if self.sub_task:
            self.reward_aux = nn.Linear(feature_dim, 1)
            self.reward_aux.weight.data = norm_col_init(self.reward_aux.weight.data, 0.01)
            self.reward_aux.bias.data.fill_(0)
        self.apply(weights_init)
        self.train()

    def forward(self, inputs, test=False):
        x, (hx, cx), action_tracker = inputs
        feature = self.encoder(x) + self.fc_action_tracker(action_tracker)
        hx, cx = self.lstm(feature, (hx, cx))
        value = self.critic(hx)
        action, entropy, log_prob = self.actor(hx, test)
        R_pred = self.reward_aux(hx) if self.sub_task else None
        return value, action, entropy, log_prob, (hx, cx), R_pred


    def sequence_features(self, x_seq, action_tracker):
        """x_seq [T, N, 2F, C, 13, 13] (tracker frames then target frames), action_tracker one-hot [T, N, n_act]."""
        T, N = x_seq.shape[0], x_seq.shape[1]
        feats = self.encoder(x_seq.reshape(T * N, *x_seq.shape[2:]))
        return (feats + self.fc_action_tracker(action_tracker.reshape(T * N, -1))).view(T, N, -1)

    def sequence_heads(self, h_seq, actions):
        T, N = h_seq.shape[0], h_seq.shape[1]
        flat = h_seq.reshape(T * N, -1)
        value = self.critic(flat)
        entropy, log_prob = policy_stats(self.actor.actor_linear(flat), actions.reshape(T * N))
        R_pred = self.reward_aux(flat).view(T, N, 1) if self.sub_task else None
        return value.view(T, N, 1), entropy.view(T, N, 1), log_prob.view(T, N, 1), R_pred

    def forward_sequence(self, x_seq, actions, action_tracker, h, c, keep):
        feats = self.sequence_features(x_seq, action_tracker)
        h_seq, h, c = lstm_sequence([self.lstm], [feats], h.unsqueeze(0), c.unsqueeze(0), keep)
        v, e, l, R_pred = self.sequence_heads(_player(h_seq, 0), actions)
        return v, e, l, (h[0], c[0]), R_pred


use_addmm_activation = hasattr(torch, "_addmm_activation")


def _addmm_relu(bias, x, w_t, out):
    """relu(x w_t + bias) into `out`: the GEMM's fused ReLU epilogue where this PyTorch build offers it."""
    if use_addmm_activation:
        return torch._addmm_activation(bias, x, w_t, out=out)
    return torch.addmm(bias, x, w_t, out=out).relu_()


class RolloutCache(object):
    """Forward activations of one rollout (see A3C_Dueling.new_cache). `lazy`: per-rollout constants built on first use (a
    launch each) — the small-shard step path never touches the stacked / transposed weight copies the batched-GEMM path reads."""

    def __init__(self):
        self.lazy = {}

    def __getattr__(self, name):
        lazy = self.__dict__.get("lazy")
        if lazy is not None and name in lazy:
            v = lazy.pop(name)()
            setattr(self, name, v)
            return v
        raise AttributeError(name)


class A3C_Dueling(nn.Module):
    """Two-player wrapper, model.py:212-265.

    Batched call:   states [N,2,stack,C,13,13], hx/cx [N,2,R]  ->
        values [N,2,1], [a_tracker [N], a_target [N]] (int64, on device), entropies [N,2,1],
        log_probs [N,2,1] (test: [N,2,n_actions]), (hx, cx) [N,2,R], R_pred [N,1] (int 0 when not tat)
    Reference-layout call (one env): states [2,stack,C,13,13], hx/cx [2,R] -> the reference's shapes:
        values [2,1], [0-d numpy ints], entropies [2,1], log_probs [2,1], (hx,cx) [2,R], R_pred [1,1].
    """

    def __init__(self, obs_space, action_space, args, device=None):
        super(A3C_Dueling, self).__init__()
        self.num_agents = len(obs_space)
        obs_shapes = [obs_space[i].shape for i in range(self.num_agents)]
        stack_frames = args.stack_frames
        rnn_out = args.rnn_out
        head_name = args.network
        self.single = args.single
        self.device = device
        if 'continuous' in head_name:
            raise NotImplementedError("continuous heads belong to the Unreal envs (out of scope)")
        self.continuous = False
        self.action_dim_tracker = action_space[0].n
        self.player0 = A3C(obs_shapes[0], action_space[0], rnn_out, head_name, stack_frames, device=device)
        if not self.single:
            if 'tat' in head_name:
                self.tat = True
                self.player1 = TAT(obs_shapes[1], action_space[1], rnn_out, head_name, stack_frames * 2,
                                   self.action_dim_tracker, device=device)
            else:
                self.tat = False
                self.player1 = A3C(obs_shapes[1], action_space[1], rnn_out, head_name, stack_frames, device=device)

    def cache_dense(self, on=True):
        """Expand the conv weights once per rollout (see CNN_maze.cache_dense)."""
        for m in self.modules():
            if isinstance(m, CNN_maze):
                m.cache_dense(on)

    def forward(self, inputs, test=False):
        states, (hx, cx) = inputs
        ref_layout = states.dim() == 5
        if ref_layout:
            states, hx, cx = states.unsqueeze(0), hx.unsqueeze(0), cx.unsqueeze(0)
        n = states.shape[0]
        value0, action_0, entropy_0, log_prob_0, (hx_0, cx_0) = self.player0(
            (states[:, 0], (hx[:, 0], cx[:, 0])), test)
        if self.single or states.shape[1] == 1:
            out = (value0.unsqueeze(1), [action_0], entropy_0.unsqueeze(1), log_prob_0.unsqueeze(1),
                   (hx_0.unsqueeze(1), cx_0.unsqueeze(1)), 0)
            return self._to_ref(out) if ref_layout else out
        R_pred = 0
        if self.tat:
            action2target = F.one_hot(action_0, self.action_dim_tracker).to(hx.dtype)       # model.py:253-254
            state_target = states.reshape(n, -1, states.shape[3], states.shape[4], states.shape[5])  # :255
            value1, action_1, entropy_1, log_prob_1, (hx1, cx1), R_pred = self.player1(
                (state_target, (hx[:, 1], cx[:, 1]), action2target), test)
        else:
            value1, action_1, entropy_1, log_prob_1, (hx1, cx1) = self.player1(
                (states[:, 1], (hx[:, 1], cx[:, 1])), test)
        out = (torch.stack([value0, value1], 1), [action_0, action_1], torch.stack([entropy_0, entropy_1], 1),
               torch.stack([log_prob_0, log_prob_1], 1), (torch.stack([hx_0, hx1], 1), torch.stack([cx_0, cx1], 1)),
               R_pred)
        return self._to_ref(out) if ref_layout else out

    fused_sampling = True   # GPU rollouts draw actions with the fused HIP head (csrc/policy_hip.hip)
    fused_actor_step = True  # ... and run the LSTMCell step as one MFMA kernel (csrc/actor_step_hip.hip)
    env_step_fused_seen = False   # (diagnostic: some step of this model ran its env step inside k_act_step)
    fused_env_step = True    # ... and end the step with ONE launch: both cells + heads + draws + the env step (k_act_step)
    pair_gemm_max_rows = int(__import__('os').environ.get('ATR_PAIR_GEMM_MAX_ROWS', '1024'))  # up to here: GEMM pairs as one launch
    # from cat_gemm_min_rows up: the LSTMCell's two GEMMs as ONE library product over [features | k h_prev] rows
    # (fused.linear_lt, K = F + R; measured in a replayed graph: 9.9 us against 13.4 for the pair kernel at 1024 rows, 7.3
    # against 7.7 at 512); the fc + ReLU pair stays one pair-kernel launch up to pair_gemm_max_rows
    cat_gate_gemm = __import__('os').environ.get('ATR_CAT_GATE_GEMM', '1') != '0'
    # coop_step (off by default: measured SLOWER than the four-launch step it replaces — 1.52 against 1.42 ms per synchronous
    # iteration at 512 envs, profiles/r05_coop_step_timeline_*.txt, DESIGN.md section 5 "The two-launch step"; ATR_COOP_STEP=1
    # turns it on): up to coop_max_rows envs everything between the stem and the next observation as ONE launch whose workgroups
    # cooperate per XCD (fused.coop_env_step, csrc/track2d_hip.hip k_coop_step): 2 launches per env step.
    # coop_workgroups: the grid size = the CUs of the stream the step is launched — or, for a captured step, REPLAYED — on
    # (None: the current stream's; train.PipelinedIteration sets it to its rollout stream's CU count before capturing)
    coop_step = __import__('os').environ.get('ATR_COOP_STEP', '0') != '0'
    # one-GEMM path: the rollout keeps the gate GEMM's OUTPUT per step (pre-activations without bias, written straight into
    # the rollout cache) instead of the activated gates — the step's last kernel writes 4R floats per row less (16.8 of its
    # 54 MB per launch at 4096 envs) and the learner's BPTT kernel recomputes the activations from them (fused.lstm_bptt pre mode)
    store_preacts = __import__('os').environ.get('ATR_STORE_PREACTS', '1') != '0'
    coop_max_rows = int(__import__('os').environ.get('ATR_COOP_MAX_ROWS', '1024'))
    coop_workgroups = None
    coop_step_seen = False
    cat_gemm_min_rows = int(__import__('os').environ.get('ATR_CAT_GEMM_MIN_ROWS', '768'))
    # atr_actor_step (both LSTMCell GEMMs + cell as one MFMA kernel per player, then two draw launches and the step launch)
    # is kept as an option: since k_act_step the GEMM pair + ONE fused cell/draw/env launch is faster at every batch size
    # measured (4096 rows: 6.15 vs 6.21 ms per iteration); ATR_MFMA_MIN_ROWS=3072 restores round 2's choice
    mfma_step_min_rows = int(__import__('os').environ.get('ATR_MFMA_MIN_ROWS', str(1 << 30)))
    # the one-GEMM step's product as csrc/gate_cell_hip.hip (f32 MFMA, the tracker's cell in its epilogue) instead of the library
    # product + both cells in k_act_step (round 6; ATR_GATE_CELL=1 turns it on — see DESIGN.md section 5 for where it stands)
    gate_cell_kernel = __import__('os').environ.get('ATR_GATE_CELL', '0') != '0'
    gate_cell_min_rows = int(__import__('os').environ.get('ATR_GATE_CELL_MIN_ROWS', '2048'))
    gate_cell_seen = False

    @torch.no_grad()
    def begin_act(self):
        """Per-rollout constants of act() (the weights do not change inside a rollout): b_ih + b_hh per player."""
        self._bsum = [l.bias_ih + l.bias_hh for l in (self.player0.lstm, self.player1.lstm)]

    def _act_cell(self, i, lstm, feat, h, c, done):
        """One LSTMCell step of the actor. (h, c) are the previous step's UN-masked outputs and `done` [N] uint8 that
        step's done flags (None: nothing pending): on the GPU the mask is applied inside the fused cell kernel
        (csrc/lstm_hip.hip); otherwise here, before nn.LSTMCell."""
        if feat.is_cuda and fused_lstm and feat.dtype == torch.float32 and h.shape[1] % 4 == 0:
            from . import fused
            bsum = getattr(self, "_bsum", None)
            b = bsum[i] if bsum is not None else lstm.bias_ih + lstm.bias_hh
            return fused.lstm_cell(torch.addmm(b, feat, lstm.weight_ih.t()), torch.mm(h, lstm.weight_hh.t()), c, done=done)
        if done is not None:
            k = (done == 0).to(h.dtype).unsqueeze(1)
            h, c = h * k, c * k
        return lstm(feat, (h, c))

    @torch.no_grad()
    def act(self, states, hs, cs, done=None):
        """Actor step of the fast path: sample both players' actions and advance their LSTM states, nothing else
        (values, entropies and log-probs are re-evaluated by forward_sequence). states [N,2,stack,C,13,13]; hs, cs:
        per-player lists of contiguous [N,R] tensors, the previous step's un-masked outputs; done [N] uint8: the
        previous step's done flags, whose LSTM reset is applied here (None: hs/cs are ready to use).
        Returns ([a_tracker, a_target], hs, cs)."""
        n = states.shape[0]
        p0, p1 = self.player0, self.player1
        if states.is_cuda and self.fused_sampling:
            if getattr(self, "_sampler", None) is None:
                from . import fused
                self._sampler = fused.ActionSampler(states.device)
            sample = self._sampler
        else:
            sample = lambda h, lin: F.softmax(lin(h), dim=1).multinomial(1).squeeze(1)
        h0, c0 = self._act_cell(0, p0.lstm, p0.encoder(states[:, 0]), hs[0], cs[0], done)
        a0 = sample(h0, p0.actor.actor_linear)
        if self.tat:
            x1 = states.reshape(n, -1, states.shape[3], states.shape[4], states.shape[5])
            fa = p1.fc_action_tracker
            feat = p1.encoder(x1) + fa.weight.t()[a0] + fa.bias          # fc_action_tracker(one_hot(a0))
        else:
            feat = p1.encoder(states[:, 1])
        h1, c1 = self._act_cell(1, p1.lstm, feat, hs[1], cs[1], done)
        a1 = sample(h1, p1.actor.actor_linear)
        return [a0, a1], [h0, h1], [c0, c1]

    # ---- rollout cache: the learner back-propagates through the forward pass the actor already evaluated -------
    def new_cache(self, num_steps, states, defer_consts=False, env_fused=False):
        """Storage for one rollout's forward activations (per player: stem outputs, fc features, LSTM gates / cell /
        hidden states for every step), filled by act_cached and consumed by forward_sequence_cached — the reference
        also evaluates the forward pass once, inside the rollout (player_util.py:46-73). None when the fused GPU path
        does not apply (CPU tensors, non-maze encoders, fused kernels switched off).
        env_fused: every step of this rollout will hand act_cached an env_out (the env step runs inside k_act_step, which
        then knows the step's done flags and writes the next step's masked hidden rows). Only then is the one-GEMM LSTMCell
        store [features | k h_prev] built: with a separate env.step (RPF / Nav-less 'Full' ids, --rescale, --stack-frames > 1,
        NumpyVecEnv) nobody would write those hidden columns."""
        p0, p1 = self.player0, self.player1
        ok = (states.is_cuda and states.dtype in (torch.float32, torch.uint8) and fused_lstm and not self.single
              and all(isinstance(p.encoder, CNN_maze) and p.encoder.small and p.encoder.use_fused
                      and p.encoder.conv1.in_channels == 1 for p in (p0, p1))
              and states.shape[-1] == 13 and states.shape[-2] == 13 and p0.lstm.hidden_size % 4 == 0
              and p0.lstm.hidden_size == p1.lstm.hidden_size)
        if not ok:
            return None
        T, N, dev = num_steps, states.shape[0], states.device
        R = p0.lstm.hidden_size
        stack = states.shape[2]
        frames = [stack, 2 * stack if self.tat else stack]
        c = RolloutCache()
        c.T, c.N, c.frames = T, N, frames
        c.y = [torch.empty((T, N * f, 512), device=dev) for f in frames]
        c.fh_all = None
        c.hm_written = 0
        same_f = p0.encoder.outdim == p1.encoder.outdim
        from . import fused as _fz
        coop = self._coop_ok(N, p0.encoder.outdim, R, dev) if (same_f and env_fused) else False
        if (same_f and env_fused and self.cat_gate_gemm and (N >= self.cat_gemm_min_rows or coop) and self._env_fused_static(N, R)
                and _fz.lt_available()
                and R == 128 and p0.lstm.weight_ih.shape == p1.lstm.weight_ih.shape and p0.encoder.outdim % 4 == 0):
            # From cat_gemm_min_rows up the LSTMCell's two GEMMs are ONE product over rows [features | k h_prev] (K = F + R):
            # slot t of this store holds step t's fc features (written by the fc GEMM with row stride F + R) next to the
            # previous step's hidden row, already zeroed where that step ended an episode (written by k_act_step); slot T is
            # the bootstrap step's. The learner reads the features in place (strided).
            Fd = p0.encoder.outdim
            c.fh_all = torch.empty((2, T + 1, N, Fd + R), device=dev)
            c.f_all = c.fh_all[:, :T, :, :Fd]
            c.f = [c.f_all[0], c.f_all[1]]
            if not defer_consts:
                c.lazy["w_cat"] = lambda: torch.stack([torch.cat([l.weight_ih, l.weight_hh], 1) for l in (p0.lstm, p1.lstm)], 0)
            if getattr(self, "_lt_ws", None) is None or self._lt_ws.device != dev:
                self._lt_ws = torch.empty(32 << 20, dtype=torch.uint8, device=dev)   # this model's chain of launches only
        elif same_f:      # one [2, T, N, F] store: a step's pair of rows is one strided batch
            c.f_all = torch.empty((2, T, N, p0.encoder.outdim), device=dev)
            c.f = [c.f_all[0], c.f_all[1]]
        else:
            c.f_all = None
            c.f = [torch.empty((T, N, p.encoder.outdim), device=dev) for p in (p0, p1)]
        c.feat1 = torch.empty((T, N, p1.encoder.outdim), device=dev) if self.tat else None
        c.pre_all = None
        # (the [features | k h] rows above — and with them this store in place of the activated gates — exist only when EVERY step
        # of the rollout is certain to take the one-GEMM branch of _act_step (_env_fused_static): the MFMA actor step and the
        # per-player fallback launches want contiguous feature rows and write acts[i])
        if c.fh_all is not None and self.store_preacts and fused_lstm and not coop:
            c.pre_all = torch.empty((2, T, N, 4 * R), device=dev)     # slot t = the gate GEMM's output of step t
            c.acts = None
        else:
            c.acts = torch.empty((2, T, N, 4 * R), device=dev)
        c.h_all = torch.empty((2, T + 1, N, R), device=dev)
        c.c_all = torch.empty((2, T + 1, N, R), device=dev)
        c.actions = torch.empty((T, 2, N), dtype=torch.int64, device=dev) if self.fused_sampling else None
        c.gates = torch.empty((2, N, 4 * R), device=dev) if (N <= self.pair_gemm_max_rows or c.fh_all is not None) else None   # scratch: pre-activations
        defer_consts = bool(defer_consts) and p0.lstm.weight_ih.shape == p1.lstm.weight_ih.shape and p0.encoder.outdim % 4 == 0
        c.consts = None
        if defer_consts:
            bs = torch.empty((2, 4 * R), device=dev)
            c.bsum = [bs[0], bs[1]]
            c.consts = dict(lstm=(p0.lstm, p1.lstm), bsum=bs, F=p0.encoder.outdim)
            if c.fh_all is not None:
                c.w_cat = torch.empty((2, 4 * R, p0.encoder.outdim + R), device=dev)
                c.consts.update(w_cat=c.w_cat, fh_all=c.fh_all)
        else:
            c.bsum = [l.bias_ih + l.bias_hh for l in (p0.lstm, p1.lstm)]
        c.lazy["whh_t"] = lambda: torch.stack([l.weight_hh.t() for l in (p0.lstm, p1.lstm)], 0)  # [2, R, 4R]
        if c.f_all is not None and p0.lstm.weight_ih.shape == p1.lstm.weight_ih.shape:
            c.lazy["wih_t"] = lambda: torch.stack([l.weight_ih.t() for l in (p0.lstm, p1.lstm)], 0)  # [2, F, 4R]: one bmm
            c.has_wih_t = True
        else:
            c.wih_t, c.has_wih_t = None, False
        if self.tat:
            fa = p1.fc_action_tracker
            if defer_consts and fa.weight.shape[1] <= 8:
                c.lazy["emb"] = lambda: fa.weight.t() + fa.bias     # (only the per-player fallback launches read it)
                c.emb_ih = torch.empty((fa.weight.shape[1], 4 * R), device=dev)
                c.consts.update(fa=fa, emb_ih=c.emb_ih)
            else:
                c.emb = fa.weight.t() + fa.bias                    # row a = fc_action_tracker(one_hot(a))
                c.emb_ih = c.emb @ p1.lstm.weight_ih.t()           # ... projected through W_ih: [n_act, 4R]
        return c

    def _mfma_step_static(self, n, R):
        """The batch-size / shape half of _act_step's choice of the per-player MFMA actor step (csrc/actor_step_hip.hip)."""
        from . import fused
        p0, p1 = self.player0, self.player1
        return bool(self.fused_actor_step and n >= self.mfma_step_min_rows and fused.actor_step_supported(p0.encoder.outdim, R)
                    and p1.encoder.outdim == p0.encoder.outdim)

    def _env_fused_static(self, n, R):
        """What new_cache can know about _act_step's `env_fused` branch before the first step (ONE predicate for both: a cache
        laid out for the one-GEMM step must never meet a step that takes another branch)."""
        p0, p1 = self.player0, self.player1
        return bool(not self._mfma_step_static(n, R) and self.fused_env_step and self.fused_sampling and R == 128
                    and p0.actor.actor_linear.weight.shape == p1.actor.actor_linear.weight.shape
                    and p0.actor.actor_linear.weight.shape[0] <= 8)

    def _coop_ok(self, N, Fd, R, dev):
        """Whether a rollout step of N envs takes the one-launch cooperative form (shape limits of atr_coop_env_step for the
        grid the launch will have)."""
        from . import fused
        if not (self.coop_step and self.fused_env_step and self.fused_sampling and N <= self.coop_max_rows):
            return False
        wg = self.coop_workgroups if self.coop_workgroups else fused.stream_cus(dev)
        return fused.coop_step_supported(N, Fd, R, wg)

    @torch.no_grad()
    def fill_consts(self, cache):
        """The deferred per-rollout constants of new_cache(defer_consts=True) with tensor ops (when the rollout's first launch
        does not make them)."""
        k = cache.consts
        if k is None:
            return
        l0, l1 = k["lstm"]
        torch.add(l0.bias_ih, l0.bias_hh, out=k["bsum"][0])
        torch.add(l1.bias_ih, l1.bias_hh, out=k["bsum"][1])
        if "w_cat" in k:
            F_ = k["F"]
            for p, l in enumerate((l0, l1)):
                k["w_cat"][p, :, :F_].copy_(l.weight_ih)
                k["w_cat"][p, :, F_:].copy_(l.weight_hh)
        if "emb_ih" in k:
            fa = k["fa"]
            torch.mm(fa.weight.t() + fa.bias, l1.weight_ih.t(), out=k["emb_ih"])
        cache.consts = None

    @torch.no_grad()
    def act_cached(self, states, cache, t, done=None, env_out=None):
        """act() for step t of a cached rollout: same sampling and state update, every intermediate written into the
        cache's slot t (LSTM state of step t lives in cache.h_all/c_all[:, t], the new one goes to slot t+1).
        env_out = (vec_env.VecTrack2D, obs slot, reward slot, done slot): where the kernels allow it the env step runs
        inside the step's last launch (fused.act_env_step); self.env_stepped tells the caller whether it did."""
        return self._act_step(states, cache, [cache.y[0][t], cache.y[1][t]], [cache.f[0][t], cache.f[1][t]],
                              cache.feat1[t] if cache.feat1 is not None else None,
                              cache.h_all[:, t], cache.c_all[:, t], cache.h_all[:, t + 1], cache.c_all[:, t + 1],
                              cache.acts[:, t] if cache.acts is not None else None,
                              cache.actions[t] if cache.actions is not None else None, done,
                              f_pair=cache.f_all[:, t] if cache.f_all is not None else None, env_out=env_out,
                              fh=(cache.fh_all[:, t], cache.fh_all[:, t + 1]) if cache.fh_all is not None else None,
                              gates=cache.pre_all[:, t] if getattr(cache, "pre_all", None) is not None else None)

    @torch.no_grad()
    def boot_values(self, states, cache, done, v_out):
        """The bootstrap forward of Agent.loss (player_util.py:109-117: one more model call on the state after the last
        step, of which only the values are used) with the rollout's fused kernels: one more actor step from the LSTM
        state of slot T into scratch buffers (the tracker's action is drawn, as the reference's forward draws it, because
        the tracker-aware target's features depend on it), then the critic heads. v_out [N, A, 1] float32 contiguous."""
        from . import fused
        b = getattr(cache, "boot", None)
        if b is None:
            b = cache.boot = RolloutCache()
            dev, N, R = states.device, cache.N, cache.h_all.shape[-1]
            b.y = [torch.empty_like(cache.y[i][0]) for i in range(2)]
            if cache.fh_all is not None:       # slot T of the [features | k h] store is the bootstrap step's
                b.f_all = cache.fh_all[:, cache.T, :, :cache.f_all.shape[-1]]
            else:
                b.f_all = torch.empty_like(cache.f_all[:, 0]) if cache.f_all is not None else None
            b.f = [b.f_all[0], b.f_all[1]] if b.f_all is not None else [torch.empty_like(cache.f[i][0]) for i in range(2)]
            b.feat1 = torch.empty_like(cache.feat1[0]) if cache.feat1 is not None else None
            b.h, b.c = torch.empty((2, N, R), device=dev), torch.empty((2, N, R), device=dev)
            b.acts = torch.empty((2, N, 4 * R), device=dev) if cache.acts is not None else None
            b.actions = torch.empty((2, N), dtype=torch.int64, device=dev) if cache.actions is not None else None
        T = cache.T
        self._act_step(states, cache, b.y, b.f, b.feat1, cache.h_all[:, T], cache.c_all[:, T], b.h, b.c, b.acts, b.actions,
                       done, f_pair=b.f_all, fh=(cache.fh_all[:, T], None) if cache.fh_all is not None else None)
        fused.heads_values2([b.h[0], b.h[1]], (self.player0.critic.critic_linear, self.player1.critic.critic_linear), v_out)
        return v_out

    def _act_step(self, states, cache, y, f_out, feat1, h_prev, c_prev, h_out, c_out, acts, actions, done, f_pair=None,
                  env_out=None, fh=None, gates=None):
        """One actor step of both players on explicit buffers: y / f_out per-player stem and fc outputs, h_prev / c_prev
        [2,N,R] (un-masked; `done` [N] uint8 of the previous step is applied inside), h_out / c_out [2,N,R], acts
        [2,N,4R] (activated gates), actions [2,N] int64 or None."""
        from . import fused
        n = states.shape[0]
        p0, p1 = self.player0, self.player1
        if getattr(self, "_sampler", None) is None:
            self._sampler = fused.ActionSampler(states.device)
        sample = self._sampler if self.fused_sampling else \
            (lambda h, lin, out=None: F.softmax(lin(h), dim=1).multinomial(1).squeeze(1))
        x_in = [states[:, 0], states.reshape(n, -1, states.shape[3], states.shape[4], states.shape[5])
                if self.tat else states[:, 1]]
        acts_out = []
        # both players' stems in one launch, both hidden GEMMs in one bmm (neither depends on the tracker's action)
        ys = fused.stem_into2(x_in[0], p0.encoder, y[0], x_in[1], p1.encoder, y[1])
        R = h_prev.shape[-1]
        # the whole LSTMCell step (both GEMMs + cell) as one MFMA kernel per player (csrc/actor_step_hip.hip), the draw as
        # a second small launch; else hidden GEMMs as one bmm + per-player input GEMM + fused cell/head/draw kernel
        # (only from 3072 rows up: one wave tile per SIMD of the chip needs 4096 rows; at 1024 rows its 22 us per call lose to
        # the library GEMMs + cell kernel, measured with tools/config_sweep.py)
        self.env_stepped = False
        mfma_step = (self._mfma_step_static(n, R) and actions is not None and self._sampler._ordinal is not None)
        env_fused = (not mfma_step and self._env_fused_static(n, R) and f_pair is not None and actions is not None
                     and self._sampler._ordinal is not None and getattr(cache, "has_wih_t", False)
                     and all(t.is_contiguous() for t in (c_prev[0], c_prev[1], h_out[0], h_out[1], c_out[0], c_out[1],
                                                         h_prev[0], h_prev[1]) + ((acts[0], acts[1]) if acts is not None else ())))
        if not env_fused and acts is None:
            raise RuntimeError("this rollout cache keeps the gate GEMM's output instead of the activated gates (store_preacts), "
                               "which only the one-GEMM step fills; the step at hand takes another branch")
        # fh = (this step's [2, N, F + R] rows of the [features | k h_prev] store, the next step's): one gate GEMM, K = F + R
        cat_gemm = env_fused and fh is not None and getattr(cache, "gates", None) is not None
        pair_gemm = env_fused and not cat_gemm and n <= self.pair_gemm_max_rows and getattr(cache, "gates", None) is not None
        hgs = None if (mfma_step or pair_gemm or cat_gemm) else torch.bmm(h_prev, cache.whh_t)
        one_launch = (actions is not None and self._sampler._ordinal is not None and R // 4 in (16, 32, 64)
                      and p0.actor.actor_linear.weight.shape[0] <= 8 and p1.actor.actor_linear.weight.shape[0] <= 8)
        # Below the MFMA-step threshold: both input projections as ONE batched GEMM on the pair's feature rows, then both
        # cells + heads + draws (tracker first, the tracker-aware target adds emb[a_tracker]) and — given env_out — the env
        # step itself as ONE launch (csrc/track2d_hip.hip k_act_step): stem, 2 x fc, 2 x bmm, act+env = 6 launches per step
        if env_fused:
            core = env_out[0] if env_out is not None else None
            if (cat_gemm and env_out is not None and fh[1] is not None and p0.encoder.outdim == p1.encoder.outdim
                    and self._coop_ok(n, p0.encoder.outdim, R, states.device)):
                # small shards: fc pair -> LSTMCell GEMM -> cells + heads + draws + env step as ONE launch whose workgroups
                # cooperate per XCD (csrc/track2d_hip.hip k_coop_step): stem + this = 2 launches per env step
                fh_t, fh_next = fh
                Fd = p0.encoder.outdim
                wg = self.coop_workgroups if self.coop_workgroups else fused.stream_cus(states.device)
                fused.coop_env_step(core, [ys[0].view(n, -1), ys[1].view(n, -1)], (p0.encoder.fc, p1.encoder.fc), fh_t, cache.w_cat,
                                    cache.gates, cache.bsum, c_prev, done, h_out, c_out, acts, self._sampler,
                                    (p0.actor.actor_linear, p1.actor.actor_linear), actions,
                                    cache.emb_ih if self.tat else None, env_out[1:], [fh_next[0][:, Fd:], fh_next[1][:, Fd:]], wg)
                cache.hm_written = getattr(cache, "hm_written", 0) + 1
                self.env_stepped = True
                self.env_step_fused_seen = True
                self.coop_step_seen = True
                return [actions[0], actions[1]]
            if pair_gemm:
                # small shards: each GEMM pair as ONE launch (csrc/pair_gemm_hip.hip) — fc + ReLU of both encoders, then both
                # LSTMCell GEMMs of both players straight to the gate pre-activations (mask and bias inside): 4 launches per step
                fused.pair_linear([ys[0].view(n, -1), ys[1].view(n, -1)], [p0.encoder.fc.weight, p1.encoder.fc.weight],
                                  [f_out[0], f_out[1]], bias=[p0.encoder.fc.bias, p1.encoder.fc.bias], relu=True)
                g = cache.gates
                fused.pair_linear([f_out[0], f_out[1]], [p0.lstm.weight_ih, p1.lstm.weight_ih], [g[0], g[1]], bias=cache.bsum,
                                  a2=[h_prev[0], h_prev[1]], w2=[p0.lstm.weight_hh, p1.lstm.weight_hh], done=done)
                ig, hg_, bs = g, None, None
                hm = None
            elif cat_gemm:
                # fc + ReLU straight into the feature columns of this step's rows (ldc = F + R), then both LSTMCell GEMMs of both
                # players as ONE batched product over those rows: 5 launches per step, one gate tensor (the bias is added in
                # k_act_step: the library has no per-batch bias). k_act_step writes the NEXT step's k h columns itself — it
                # knows this step's done flags — so the mask on h is already in the rows; `done` still masks c_prev.
                fh_t, fh_next = fh
                Fd = f_out[0].shape[-1]
                if fh_next is not None and env_out is None:
                    # (only the bootstrap step has no next slot; every other step's masked hidden row is written by k_act_step
                    # from the done flags of the env step it runs itself)
                    raise RuntimeError("one-GEMM LSTMCell rows need the env step inside k_act_step: new_cache(env_fused=True) "
                                       "was promised an env_out for every step of the rollout")
                if n <= self.pair_gemm_max_rows:      # (both encoders' fc + ReLU as one pair-kernel launch, row stride F + R)
                    fused.pair_linear([ys[0].view(n, -1), ys[1].view(n, -1)], [p0.encoder.fc.weight, p1.encoder.fc.weight],
                                      [fh_t[0][:, :Fd], fh_t[1][:, :Fd]], bias=[p0.encoder.fc.bias, p1.encoder.fc.bias], relu=True)
                else:
                    for i, p in enumerate((p0, p1)):
                        fused.linear_lt(ys[i].view(n, -1), p.encoder.fc.weight, fh_t[i][:, :Fd], bias=p.encoder.fc.bias,
                                        relu=True, workspace=self._lt_ws)
                # (gates: this step's slot of the rollout's pre-activation store — kept for the learner instead of the activated
                # gates — else the scratch tensor)
                g = gates if gates is not None else cache.gates
                if (self.gate_cell_kernel and gates is not None and acts is None and n >= self.gate_cell_min_rows and R == 128
                        and Fd % 32 == 0):
                    # the product as this build's own MFMA kernel with the TRACKER's cell as its epilogue (csrc/gate_cell_hip.hip):
                    # the tracker's gates are written once for the learner and never read back by the rollout; the tracker-aware
                    # target's cell still needs the tracker's draw, so it stays in k_act_step (ig[0] = None tells it so)
                    fused.gate_cell(fh_t, cache.w_cat, cache.bsum, g, c_prev, done, h_out, c_out, cell=(True, False))
                    ig = [None, g[1]]
                    self.gate_cell_seen = True
                else:
                    fused.linear_lt(fh_t, cache.w_cat, g, workspace=self._lt_ws)
                    ig = g
                hg_, bs = None, cache.bsum
                hm = [fh_next[0][:, Fd:], fh_next[1][:, Fd:]] if (fh_next is not None and env_out is not None) else None
            else:
                hm = None
                for i, p in enumerate((p0, p1)):
                    _addmm_relu(p.encoder.fc.bias, ys[i].view(n, -1), p.encoder.fc.weight.t(), f_out[i])
                ig, hg_, bs = torch.bmm(f_pair, cache.wih_t), hgs, cache.bsum
            fused.act_env_step(core, ig, hg_, bs, c_prev, done, h_out, c_out, acts, self._sampler,
                               (p0.actor.actor_linear, p1.actor.actor_linear), actions,
                               emb=cache.emb_ih if self.tat else None, env_out=env_out[1:] if env_out is not None else None,
                               hm_out=hm)
            if hm is not None:            # slot t + 1 of the [features | k h] rows now holds k_t h_t (the learner's dW_hh reads
                cache.hm_written = getattr(cache, "hm_written", 0) + 1        # them when every step of the rollout wrote one)
            self.env_stepped = env_out is not None
            self.env_step_fused_seen = self.env_step_fused_seen or self.env_stepped
            return [actions[0], actions[1]]
        # players that do not see each other's action (maze-lstm pairs): both input projections as ONE batched GEMM on
        # the pair's feature rows and both cells + heads + draws as ONE launch — 7 launches per env step instead of 9
        if (one_launch and not mfma_step and not self.tat and f_pair is not None and getattr(cache, "has_wih_t", False)
                and p0.actor.actor_linear.weight.shape == p1.actor.actor_linear.weight.shape):
            for i, p in enumerate((p0, p1)):
                _addmm_relu(p.encoder.fc.bias, ys[i].view(n, -1), p.encoder.fc.weight.t(), f_out[i])
            ig = torch.bmm(f_pair, cache.wih_t)
            fused.lstm_cell_act2_into(ig, hgs, cache.bsum, c_prev, done, h_out, c_out, acts, self._sampler,
                                      (p0.actor.actor_linear, p1.actor.actor_linear), actions)
            return [actions[0], actions[1]]
        # tracker-aware pair below the MFMA-step threshold: the target's cell needs the tracker's action, its input
        # projection does not — both projections still go out as one batched GEMM, ahead of the tracker's cell
        ig_pair = None
        if one_launch and not mfma_step and f_pair is not None and getattr(cache, "has_wih_t", False):
            for i, p in enumerate((p0, p1)):
                _addmm_relu(p.encoder.fc.bias, ys[i].view(n, -1), p.encoder.fc.weight.t(), f_out[i])
            ig_pair = torch.bmm(f_pair, cache.wih_t)
        for i, p in enumerate((p0, p1)):
            enc = p.encoder
            tat = i == 1 and self.tat
            if ig_pair is not None:
                acts_out.append(fused.lstm_cell_act_into(
                    ig_pair[i], hgs[i], c_prev[i], done, h_out[i], c_out[i], acts[i],
                    self._sampler, p.actor.actor_linear, actions[i],
                    emb=cache.emb_ih if tat else None, act_in=acts_out[0] if tat else None, bias=cache.bsum[i]))
                continue
            f = _addmm_relu(enc.fc.bias, ys[i].view(n, -1), enc.fc.weight.t(), f_out[i])
            if mfma_step:
                fused.actor_step_into(f, h_prev[i], c_prev[i], done, p.lstm, cache.bsum[i], h_out[i], c_out[i], acts[i],
                                      emb=cache.emb_ih if tat else None, act_in=acts_out[0] if tat else None)
                acts_out.append(sample(h_out[i], p.actor.actor_linear, out=actions[i]))
                continue
            if tat and not one_launch:
                f = torch.add(f, cache.emb[acts_out[0]], out=feat1)
            ig = torch.addmm(cache.bsum[i], f, p.lstm.weight_ih.t())
            if one_launch:   # cell (+ tracker-action embedding, projected through W_ih once per rollout) + actor head + draw
                acts_out.append(fused.lstm_cell_act_into(
                    ig, hgs[i], c_prev[i], done, h_out[i], c_out[i], acts[i],
                    self._sampler, p.actor.actor_linear, actions[i],
                    emb=cache.emb_ih if tat else None, act_in=acts_out[0] if tat else None))
                continue
            fused.lstm_cell_into(ig, hgs[i], c_prev[i], done, h_out[i], c_out[i], acts[i])
            if actions is not None:
                acts_out.append(sample(h_out[i], p.actor.actor_linear, out=actions[i]))
            else:
                acts_out.append(sample(h_out[i], p.actor.actor_linear))
        return acts_out

    def cached_hidden(self, cache, states_seq, actions_seq, keep, need=None):
        """The differentiable graph (stem -> fc -> [+ tracker-action embedding] -> LSTM) rebuilt around a cached
        rollout's stored activations: per-player hidden sequences [T, N, R] with autograd history, nothing evaluated
        forward."""
        from . import fused
        T, N = cache.T, cache.N
        p0, p1 = self.player0, self.player1
        x_in = [states_seq[:, :, 0],
                states_seq.reshape(T, N, -1, states_seq.shape[4], states_seq.shape[5], states_seq.shape[6])
                if self.tat else states_seq[:, :, 1]]
        feats = []
        # the rollout kept the gate GEMM's outputs: the BPTT kernel re-activates them (bias, the tracker-action embedding of
        # the tracker-aware target, then the cell's own sigmoid / tanh: the values the rollout computed, bit for bit)
        pre = None
        if getattr(cache, "pre_all", None) is not None:
            pre = dict(bias=[cache.bsum[0], cache.bsum[1]], emb=cache.emb_ih if self.tat else None, emb_player=1,
                       act=actions_seq[:, :, 0])
        fold = None
        if self.tat and (need is None or need[1]) and fused.embed_fold_ok(pre, cache.h_all, cache.c_all, keep, p1.fc_action_tracker):
            fold = p1.fc_action_tracker
        for i, p in enumerate((p0, p1)):
            enc = p.encoder
            y = fused.stem_cached(x_in[i], cache.y[i].view(-1, 512), enc.conv1, enc.conv2)
            f = fused.linear_relu_cached(y.view(T * N, -1), enc.fc, cache.f[i].view(T * N, -1))
            if i == 1 and self.tat:      # + fc_action_tracker(one_hot(a_tracker)) (model.py:193-194): a row gather
                if fold is not None or (need is not None and not need[1]):
                    pass         # (folded: the LSTM node contracts the raw features and makes the embedding's gradients from
                                 #  the by-action column sums of dG — no f + E[a] tensor, no gather of dL/df by action; or
                                 #  the target is not trained in this mode and nothing reads its features)
                elif p.fc_action_tracker.weight.shape[1] <= 8 and f.shape[1] % 4 == 0:
                    # ([T, N] view of the tracker's column of the [T, players, N] action store: read in place)
                    f = fused.embed_add(f, p.fc_action_tracker, actions_seq[:, :, 0])
                else:
                    a_tr = actions_seq[:, :, 0].reshape(T * N)
                    f = f + p.fc_action_tracker(F.one_hot(a_tr, self.action_dim_tracker).to(f.dtype))
            feats.append(f)
        # the masked hidden rows k_{t-1} h_{t-1} the rollout left next to the features (one-GEMM LSTMCell path): dW_hh contracts
        # them as they are — no mask to apply inside the weight-gradient kernel, whose operands then go straight into LDS
        hm = None
        if getattr(cache, "fh_all", None) is not None and getattr(cache, "hm_written", 0) >= T:
            Fd = cache.f_all.shape[-1]
            hm = [cache.fh_all[i, :T, :, Fd:].view(T * N, -1) for i in range(2)]
        acts = cache.pre_all if pre is not None else cache.acts
        return fused.lstm_sequence_cached([p0.lstm, p1.lstm], feats, keep, cache.h_all, cache.c_all, acts, need, hm=hm, pre=pre,
                                          fold=fold)

    def forward_sequence_cached(self, cache, states_seq, actions_seq, keep):
        """forward_sequence over a cached rollout: only the heads are evaluated forward; the backward pass is the
        same as forward_sequence's. Same return values."""
        p0, p1 = self.player0, self.player1
        h_seq = self.cached_hidden(cache, states_seq, actions_seq, keep)
        v0, e0, l0 = p0.sequence_heads(h_seq[0], actions_seq[:, :, 0])
        R_pred = 0
        if self.tat:
            v1, e1, l1, R_pred = p1.sequence_heads(h_seq[1], actions_seq[:, :, 1])
        else:
            v1, e1, l1 = p1.sequence_heads(h_seq[1], actions_seq[:, :, 1])
        return torch.stack([v0, v1], 2), torch.stack([e0, e1], 2), torch.stack([l0, l1], 2), R_pred

    def forward_sequence(self, states_seq, actions_seq, hx, cx, keep):
        """Re-evaluate T stored steps with gradients, time-batched (the learner half of the rollout driver):
        states_seq [T, N, 2, stack, C, 13, 13], actions_seq [T, N, 2] int64, hx/cx [N, 2, R] at the start of the
        rollout, keep [T, N] float (0 where the env finished at that step). Returns values [T,N,2,1], entropies
        [T,N,2,1], log_probs [T,N,2,1], R_pred [T,N,1] (0 when not tat). Numerically the same quantities as T calls
        of forward() with those actions; the encoder and all heads run as single GEMMs over T*N rows."""
        T, N = states_seq.shape[0], states_seq.shape[1]
        p0, p1 = self.player0, self.player1
        f0 = p0.sequence_features(states_seq[:, :, 0])
        if self.tat:
            a2t = F.one_hot(actions_seq[:, :, 0], self.action_dim_tracker).to(hx.dtype)
            x1 = states_seq.reshape(T, N, -1, states_seq.shape[4], states_seq.shape[5], states_seq.shape[6])
            f1 = p1.sequence_features(x1, a2t)
        else:
            f1 = p1.sequence_features(states_seq[:, :, 1])
        # both players' recurrences in lock step: one bmm + one fused cell per time step for the pair
        h_seq, _, _ = lstm_sequence([p0.lstm, p1.lstm], [f0, f1],
                                    hx.transpose(0, 1).contiguous(), cx.transpose(0, 1).contiguous(), keep)
        v0, e0, l0 = p0.sequence_heads(_player(h_seq, 0), actions_seq[:, :, 0])
        R_pred = 0
        if self.tat:
            v1, e1, l1, R_pred = p1.sequence_heads(_player(h_seq, 1), actions_seq[:, :, 1])
        else:
            v1, e1, l1 = p1.sequence_heads(_player(h_seq, 1), actions_seq[:, :, 1])
        return torch.stack([v0, v1], 2), torch.stack([e0, e1], 2), torch.stack([l0, l1], 2), R_pred

    @staticmethod
    def _to_ref(out):
        values, actions, entropies, log_probs, (hx, cx), R_pred = out
        actions = [np.squeeze(a.cpu().numpy()) for a in actions]            # model.py:50
        R_pred = R_pred if isinstance(R_pred, int) else R_pred              # [1,1] already
        return values[0], actions, entropies[0], log_probs[0], (hx[0], cx[0]), R_pred
