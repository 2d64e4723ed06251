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


@pytest.mark.parametrize("M", [1, 3, 64, 1000, 12288, 16391])      # (from 16384 frames up: the 16-frames-per-pass kernels)
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


def test_fused_lstm_sequence_matches_the_aten_recurrence():
    """csrc/lstm_hip.hip (one autograd node for the masked two-player recurrence) against the per-step ATen path:
    outputs, final state and every gradient (fp32; tolerances cover sigmoid/tanh implementation differences)."""
    from active_tracking_rl_amd import model as M
    torch.manual_seed(3)
    T, P, N, Fdim, R = 7, 2, 37, 256, 128
    lstms = [torch.nn.LSTMCell(Fdim, R).cuda() for _ in range(P)]
    feats = torch.randn(T, P, N, Fdim, device="cuda", requires_grad=True)
    h0 = torch.randn(P, N, R, device="cuda", requires_grad=True)
    c0 = torch.randn(P, N, R, device="cuda", requires_grad=True)
    keep = (torch.rand(T, N, device="cuda") > 0.3).float()
    gout = torch.randn(T, P, N, R, device="cuda")
    gh, gc = torch.randn(P, N, R, device="cuda"), torch.randn(P, N, R, device="cuda")
    params = [p for l in lstms for p in l.parameters()]
    res = []
    for fused_on in (True, False):
        M.fused_lstm = fused_on
        try:
            f_in = [feats[:, p] for p in range(P)] if fused_on else feats
            h_seq, h, c = M.lstm_sequence(lstms, f_in, h0, c0, keep)
        finally:
            M.fused_lstm = True
        if isinstance(h_seq, (list, tuple)):
            h_seq = torch.stack(list(h_seq), 1)
        loss = (h_seq * gout).sum() + (h * gh).sum()
        grads = torch.autograd.grad(loss, params + [feats, h0, c0])
        res.append((h_seq.detach(), h.detach(), c.detach(), grads))
    (hs_a, h_a, c_a, g_a), (hs_b, h_b, c_b, g_b) = res
    torch.testing.assert_close(hs_a, hs_b, rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(h_a, h_b, rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(c_a, c_b, rtol=1e-4, atol=2e-5)
    for a, b in zip(g_a, g_b):
        scale = float(b.abs().max()) + 1e-6
        assert float((a - b).abs().max()) <= 2e-4 * scale, (a.shape, float((a - b).abs().max()), scale)


def test_fused_lstm_single_player_and_rollout_cell():
    from active_tracking_rl_amd import fused, model as M
    torch.manual_seed(4)
    N, Fdim, R = 50, 256, 128
    lstm = torch.nn.LSTMCell(Fdim, R).cuda()
    x, h, c = torch.randn(N, Fdim, device="cuda"), torch.randn(N, R, device="cuda"), torch.randn(N, R, device="cuda")
    done = (torch.rand(N, device="cuda") > 0.5).to(torch.uint8)
    with torch.no_grad():
        k = (done == 0).float().unsqueeze(1)
        h_ref, c_ref = lstm(x, (h * k, c * k))
        ig = torch.addmm(lstm.bias_ih + lstm.bias_hh, x, lstm.weight_ih.t())
        h_f, c_f = fused.lstm_cell(ig, torch.mm(h, lstm.weight_hh.t()), c, done=done)
        h_n, c_n = fused.lstm_cell(ig, torch.mm(h, lstm.weight_hh.t()), c)
        h_r2, c_r2 = lstm(x, (h, c))
    torch.testing.assert_close(h_f, h_ref, rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(c_f, c_ref, rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(h_n, h_r2, rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(c_n, c_r2, rtol=1e-4, atol=2e-5)
    # single-player sequence (P = 1), final state included
    T = 5
    feats = torch.randn(T, N, Fdim, device="cuda")
    keep = (torch.rand(T, N, device="cuda") > 0.3).float()
    outs = []
    for fused_on in (True, False):
        M.fused_lstm = fused_on
        try:
            f_in = [feats] if fused_on else feats.unsqueeze(1)
            with torch.no_grad():
                h_seq, hT, cT = M.lstm_sequence([lstm], f_in, h.unsqueeze(0), c.unsqueeze(0), keep)
        finally:
            M.fused_lstm = True
        outs.append((M._player(h_seq, 0), hT, cT))
    for a, b in zip(*outs):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=2e-5)


def test_gae_kernel_matches_the_reference_recursion():
    from active_tracking_rl_amd import fused
    torch.manual_seed(5)
    T, N, A = 20, 300, 2
    rew = torch.randn(T, N, A, 1, device="cuda")
    val = torch.randn(T + 1, N, A, 1, device="cuda")
    nd = (torch.rand(T, N, device="cuda") > 0.1).float()
    gamma, tau = 0.9, 1.0
    R, gae = fused.gae_returns(rew, val, nd, gamma, tau)
    ndv = nd.view(T, N, 1, 1)
    r_run, g_run = val[T], torch.zeros_like(val[T])
    for i in reversed(range(T)):                              # player_util.py:118-141 of the reference
        r_run = gamma * r_run * ndv[i] + rew[i]
        delta_t = rew[i] + gamma * val[i + 1] * ndv[i] - val[i]
        g_run = g_run * gamma * tau * ndv[i] + delta_t
        torch.testing.assert_close(R[i], r_run, rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(gae[i], g_run, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("M0,M1", [(4096, 8192), (5, 3), (1, 700), (4096, 4096), (3100, 5), (9, 8183), (2048, 4096)])
def test_paired_stem_launch_equals_two_launches(M0, M1):
    """(the last four: launches that take the 16-frames-per-pass kernel — passes that fill the workgroup slots evenly —, with a
    one-pass problem on either side and ragged last passes)"""
    from active_tracking_rl_amd import fused
    from active_tracking_rl_amd.model import CNN_maze
    torch.manual_seed(M0 + M1)
    enc = [CNN_maze((1, 13, 13), 1).cuda(), CNN_maze((1, 13, 13), 2).cuda()]
    obs = torch.randint(0, 5, (max(M0, (M1 + 1) // 2), 2, 13, 13), device="cuda").float()
    xa = obs[:M0, 0]                                   # strided view (one agent's frames)
    xb = obs.reshape(-1, 13, 13)[:M1]                  # contiguous frames
    ya, yb = torch.empty(M0, 512, device="cuda"), torch.empty(M1, 512, device="cuda")
    fused.stem_into2(xa, enc[0], ya, xb, enc[1], yb)
    ra = fused.stem_into(xa, enc[0].conv1, enc[0].conv2, torch.empty_like(ya))
    rb = fused.stem_into(xb, enc[1].conv1, enc[1].conv2, torch.empty_like(yb))
    assert torch.equal(ya, ra) and torch.equal(yb, rb)


def test_actor_cell_kernel_cell_embedding_and_draw():
    """atr_lstm_cell_forward_act: same (h, c, gates) as the plain cell on ig + emb[a], and actions distributed as
    softmax(actor(h)) (chi-square on identical rows), different across ordinals."""
    from active_tracking_rl_amd import fused
    torch.manual_seed(6)
    N, R, A = 100000, 128, 4
    actor = torch.nn.Linear(R, A).cuda()
    with torch.no_grad():
        actor.weight.mul_(6.0); actor.bias.normal_()
    ig = torch.randn(1, 4 * R, device="cuda").expand(N, 4 * R).contiguous()
    hg = torch.randn(1, 4 * R, device="cuda").expand(N, 4 * R).contiguous()
    c = torch.randn(1, R, device="cuda").expand(N, R).contiguous()
    emb = torch.randn(4, 4 * R, device="cuda")
    a_in = torch.full((N,), 2, dtype=torch.int64, device="cuda")
    done = torch.zeros(N, dtype=torch.uint8, device="cuda")
    sampler = fused.ActionSampler(torch.device("cuda"), seed=99)
    sampler.begin_block()
    outs = [torch.empty(N, R, device="cuda") for _ in range(2)] + [torch.empty(N, 4 * R, device="cuda")]
    acts1 = fused.lstm_cell_act_into(ig, hg, c, done, outs[0], outs[1], outs[2], sampler, actor,
                                     torch.empty(N, dtype=torch.int64, device="cuda"), emb=emb, act_in=a_in)
    h_ref, c_ref = fused.lstm_cell(ig + emb[2], hg, c, done=done)
    torch.testing.assert_close(outs[0], h_ref, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(outs[1], c_ref, rtol=1e-5, atol=1e-6)
    p = torch.softmax(actor(h_ref[:1]), 1)[0].detach().double().cpu().numpy()
    counts = np.bincount(acts1.cpu().numpy(), minlength=A).astype(np.float64)
    chi2 = (((counts - N * p) ** 2) / (N * p + 1e-9)).sum()
    assert chi2 < 30.0, (chi2, counts / N, p)
    acts2 = fused.lstm_cell_act_into(ig, hg, c, done, outs[0], outs[1], None, sampler, actor,
                                     torch.empty(N, dtype=torch.int64, device="cuda"), emb=emb, act_in=a_in)
    sampler.end_block()
    assert not torch.equal(acts1, acts2)
    # done rows restart from a zero state
    done[: N // 2] = 1
    sampler.begin_block()
    fused.lstm_cell_act_into(ig, hg, c, done, outs[0], outs[1], None, sampler, actor,
                             torch.empty(N, dtype=torch.int64, device="cuda"))
    sampler.end_block()
    h_ref, c_ref = fused.lstm_cell(ig, hg, c, done=done)
    torch.testing.assert_close(outs[0], h_ref, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("K,M,N", [(81920, 256, 512), (20480, 512, 256), (4099, 128, 384), (8192, 128, 128),
                                   (4160, 128, 128), (20800, 128, 256), (41600, 256, 128), (4112, 128, 128)])
def test_gemm_tn_matches_float64_reference(K, M, N):
    """atr_gemm_tn (x1^T x2 for tall operands, csrc/gemm_tn_hip.hip) against a float64 matmul; ragged K tail included.
    The sizes cover both chunk sizes (16 rows below 20 481 rows of K, 32 above), the operands-straight-into-LDS path (K a
    whole number of chunks, no row factors: the first call) and the register-staged one (ragged K; row factors: the second
    call), and chunk counts that leave the planner's last K-slices EMPTY (4160 = 260 chunks over 256 slices of 2; 20 800 and
    41 600 likewise with 32-row chunks): those workgroups must store zeros without touching memory past the operands.
    Tolerance: fp32 accumulation over K products of N(0,1) operands."""
    from active_tracking_rl_amd import fused
    torch.manual_seed(K + M)
    x1, x2 = torch.randn(K, M, device="cuda"), torch.randn(K, N, device="cuda")
    c = fused.gemm_tn(x1, x2)
    ref = (x1.double().t() @ x2.double())
    assert c.shape == (M, N)
    assert float((c.double() - ref).abs().max()) < 3e-5 * K ** 0.5 * 4
    c2 = fused.gemm_tn(x1, x2)
    assert torch.equal(c, c2)                                   # fixed reduction order: bit-reproducible
    # fused extras: row scaling of x1 (the episode mask of dW_hh) and the column sums (the bias gradient)
    scale = (torch.rand(K, device="cuda") > 0.2).float()
    cs_, colsum = fused.gemm_tn(x1, x2, row_scale=scale, colsum=True)
    xs = x1.double() * scale.double().unsqueeze(1)
    assert float((cs_.double() - xs.t() @ x2.double()).abs().max()) < 3e-5 * K ** 0.5 * 4
    assert float((colsum.double() - xs.sum(0)).abs().max()) < 3e-5 * K ** 0.5 * 4


@pytest.mark.parametrize("K,M,N", [(81920, 256, 512), (20480, 512, 384), (4160, 128, 128), (4099, 128, 256)])
def test_gemm_tn_corun_mode_computes_the_same_products(K, M, N):
    """fused.gemm_tn_corun (atr_gemm_tn_set_corun: the launch planned for ONE workgroup per CU, which the pipelined schedule's
    learner graphs are captured with): the same products against float64 — 16-row chunks at every K, padded dynamic LDS, its own
    K split and workspace size — bit-reproducible, and the mode is back to what it was afterwards (also after an exception)."""
    from active_tracking_rl_amd import fused
    torch.manual_seed(K + N)
    x1, x2 = torch.randn(K, M, device="cuda"), torch.randn(K, N, device="cuda")
    ref = x1.double().t() @ x2.double()
    tol = 3e-5 * K ** 0.5 * 4
    c0 = fused.gemm_tn(x1, x2)
    with fused.gemm_tn_corun(True):
        c1, cs1 = fused.gemm_tn(x1, x2, colsum=True)
        c2, _ = fused.gemm_tn(x1, x2, colsum=True)
        scale = (torch.rand(K, device="cuda") > 0.2).float()
        c3 = fused.gemm_tn(x1, x2, row_scale=scale)
    assert float((c1.double() - ref).abs().max()) < tol and float((c0.double() - ref).abs().max()) < tol
    assert torch.equal(c1, c2)
    assert float((cs1.double() - x1.double().sum(0)).abs().max()) < tol
    assert float((c3.double() - (x1.double() * scale.double().unsqueeze(1)).t() @ x2.double()).abs().max()) < tol
    assert fused.lib().atr_gemm_tn_set_corun(0) == 0                     # the context manager restored "off"
    try:
        with fused.gemm_tn_corun(True):
            raise KeyError("x")
    except KeyError:
        pass
    assert fused.lib().atr_gemm_tn_set_corun(0) == 0
    assert torch.equal(fused.gemm_tn(x1, x2), c0)                        # and the default mode computes what it did before


@pytest.mark.parametrize("K", [10240, 4099])
def test_grouped_weight_gradients_fill_the_bucket_slices(K):
    """fused.DeferredWeightGrads (atr_gemm_tn_grouped: the weight-gradient products of one backward pass as one launch + one
    reduction writing into slices of the flat gradient bucket) against float64: six products of the learner's shapes over the
    same K rows, bias column sums into one and into two destinations, and the shifted episode mask of dW_hh (row k of x1
    scaled by keep[k - N], rows below N by 1). Then FlatParams.set_grads: slices already in place are left alone, the other
    gradients (one None) go in with the segment-scatter launch."""
    from active_tracking_rl_amd import fused
    from active_tracking_rl_amd.shared_optim import FlatParams
    torch.manual_seed(K)
    dev = "cuda"
    shapes = [(512, 256), (512,), (512,), (512, 128), (256, 512), (256,), (256, 1024), (256,), (512, 256), (512, 128), (7, 3),
              (1,), (5,)]
    params = [torch.nn.Parameter(torch.randn(*sh, device=dev)) for sh in shapes]
    bucket = FlatParams(params)
    bucket.grad.fill_(7.0)
    N = 512 if K % 512 == 0 else 17
    keep = (torch.rand(K, device=dev) > 0.1).float()
    dG = [torch.randn(K, 512, device=dev) for _ in range(2)]
    feat, h = torch.randn(K, 256, device=dev), torch.randn(K, 128, device=dev)
    dpre = [torch.randn(K, 256, device=dev) for _ in range(2)]
    y0, y1 = torch.randn(K, 512, device=dev), torch.randn(K, 1024, device=dev)
    with fused.deferred_weight_grads(bucket) as q:
        r = [q.add(dG[0], feat, params[0], biases=(params[1], params[2])),
             q.add(dG[0], h, params[3], row_scale=keep, shift=N),
             q.add(dpre[0], y0, params[4], biases=(params[5],)),
             q.add(dpre[1], y1, params[6], biases=(params[7],)),
             q.add(dG[1], feat, params[8]),
             q.add(dG[1], h, params[9], row_scale=keep, shift=N)]
        assert all(x is not None for x in r)
        assert q.add(dG[1][:, :100], h, params[10]) is None                 # not the kernel's shape class: caller's job
        assert q.add(dG[1][:4096], h[:4096], params[9]) is None             # another K cannot join this group
        grads = [None] * len(params)
        for idx, x in zip((0, 3, 4, 6, 8, 9), r):
            grads[idx] = x[0]
        grads[1], grads[2], grads[5], grads[7] = r[0][1][0], r[0][1][1], r[2][1][0], r[3][1][0]
        small = torch.randn(40, device=dev)
        grads[10], grads[12] = small[3:24].view(7, 3), small[30:35]        # unaligned slices of a packed output; [11] unused
        q.check(grads)
        q.flush()
    bucket.set_grads(grads)
    torch.cuda.synchronize()
    tol = 3e-5 * K ** 0.5 * 4
    scale = torch.cat([torch.ones(N, device=dev), keep[:K - N]]).double().unsqueeze(1)
    ref = {0: dG[0].double().t() @ feat.double(), 1: dG[0].double().sum(0), 2: dG[0].double().sum(0),
           3: (dG[0].double() * scale).t() @ h.double(), 4: dpre[0].double().t() @ y0.double(), 5: dpre[0].double().sum(0),
           6: dpre[1].double().t() @ y1.double(), 7: dpre[1].double().sum(0), 8: dG[1].double().t() @ feat.double(),
           9: (dG[1].double() * scale).t() @ h.double(), 10: small[3:24].view(7, 3).double(), 11: torch.zeros(1, device=dev),
           12: small[30:35].double()}
    views = bucket.grad_views()
    for i, want in ref.items():
        assert float((views[i].double() - want).abs().max()) <= (tol if i < 10 else 0.0), i
    # the padding between slots is untouched, nothing outside the slots was written
    used = torch.zeros_like(bucket.grad, dtype=torch.bool)
    for prm, off in zip(bucket.params, bucket.offsets):
        used[off:off + prm.numel()] = True
    assert bool((bucket.grad[~used] == 7.0).all())
    # fixed reduction order: a second pass gives the same bits
    before = bucket.grad.clone()
    with fused.deferred_weight_grads(bucket) as q:
        q.add(dG[0], feat, params[0], biases=(params[1], params[2]))
        q.add(dG[0], h, params[3], row_scale=keep, shift=N)
        q.add(dpre[0], y0, params[4], biases=(params[5],))
        q.add(dpre[1], y1, params[6], biases=(params[7],))
        q.add(dG[1], feat, params[8])
        q.add(dG[1], h, params[9], row_scale=keep, shift=N)
        q.flush()
    torch.cuda.synchronize()
    assert torch.equal(before, bucket.grad)


@pytest.mark.parametrize("M", [5, 4096])
def test_u8_frames_give_the_float_results_bit_for_bit(M):
    """atr_stem_*_u8 (the env's byte observations decoded inside conv1's load) against the float entry points on
    float(x): the same arithmetic on the same values, so forward and all four gradients are identical, also on a
    strided view (one agent's frames of an [N,2,13,13] observation tensor)."""
    from active_tracking_rl_amd import fused
    torch.manual_seed(M)
    dev = "cuda"
    conv1 = torch.nn.Conv2d(1, 16, 3, 2, 1).to(dev)
    conv2 = torch.nn.Conv2d(16, 32, 3, 2, 1).to(dev)
    obs = torch.tensor(np.random.RandomState(M).choice([0, 1, 2, 4], size=(M, 2, 13, 13)).astype(np.uint8), device=dev)
    params = [conv1.weight, conv1.bias, conv2.weight, conv2.bias]
    for view in (obs[:, 1], obs):                       # strided (stride 338 bytes) and dense
        yu = fused.stem(view, conv1, conv2)
        yf = fused.stem(view.float(), conv1, conv2)
        assert torch.equal(yu, yf)
        g = torch.randn_like(yf)
        gu = torch.autograd.grad((yu * g).sum(), params)
        gf = torch.autograd.grad((yf * g).sum(), params)
        assert all(torch.equal(a, b) for a, b in zip(gu, gf))
    out_u, out_f = torch.empty((M, 512), device=dev), torch.empty((M, 512), device=dev)
    fused.stem_into(obs[:, 0], conv1, conv2, out_u)
    fused.stem_into(obs[:, 0].float(), conv1, conv2, out_f)
    assert torch.equal(out_u, out_f)


@pytest.mark.parametrize("M", [4099, 16384, 16387, 20301])
def test_sixteen_frame_forward_equals_the_wave_per_frame_forward_bit_for_bit(M):
    """From 16384 frames up — and below that wherever the 16-frame passes fill the chip's workgroup slots evenly (4099 frames: 257
    passes on 512 slots) — atr_stem_forward* runs 16 frames per workgroup pass (k_stem_fwd16: frames on the MFMA rows, border
    taps not issued); otherwise one wave per frame (k_stem_fwd). A frame's output must not depend on the launch it was part
    of (the rollout evaluates 2 N frames per step, the recompute learner 20 x 2 N at once): the same frames in chunks of 1000
    give the same bits — floats, bytes, a strided view, a ragged last pass, and the two-problem launch."""
    from active_tracking_rl_amd import fused
    torch.manual_seed(M)
    dev = "cuda"
    encs = []
    for _ in range(2):
        conv1 = torch.nn.Conv2d(1, 16, 3, 2, 1).to(dev)
        conv2 = torch.nn.Conv2d(16, 32, 3, 2, 1).to(dev)
        with torch.no_grad():
            conv1.weight.mul_(2.0); conv2.weight.mul_(3.0); conv1.bias.normal_(0, 0.2); conv2.bias.normal_(0, 0.2)
        encs.append((conv1, conv2))
    obs = torch.tensor(np.random.RandomState(M).choice([0, 1, 2, 4], size=(M, 2, 13, 13)).astype(np.uint8), device=dev)

    def chunked(view, enc):
        return torch.cat([fused.stem(view[i:i + 1000], enc[0], enc[1]) for i in range(0, M, 1000)])
    for view in (obs[:, 1], obs[:, 0].contiguous(), obs[:, 0].float()):
        big = fused.stem(view, *encs[0])
        assert torch.equal(big, chunked(view, encs[0]))
        yr = _ref(view.float().reshape(M, 169), encs[0][0].weight, encs[0][0].bias, encs[0][1].weight, encs[0][1].bias)
        torch.testing.assert_close(big, yr, rtol=1e-5, atol=1e-5)

    class E:
        def __init__(self, c):
            self.conv1, self.conv2 = c
    oa, ob = torch.empty((M, 512), device=dev), torch.empty((M - 7, 512), device=dev)
    fused.stem_into2(obs[:, 0], E(encs[0]), oa, obs[:M - 7, 1], E(encs[1]), ob)
    assert torch.equal(oa, chunked(obs[:, 0], encs[0]))
    assert torch.equal(ob, chunked(obs[:, 1], encs[1])[:M - 7])


@pytest.mark.parametrize("M0,M1", [(4096, 8192), (5000, 11003), (8190, 4090), (1531, 1541), (3072, 9216)])
def test_paired_sixteen_frame_launch_hands_every_pass_to_exactly_one_workgroup(M0, M1):
    """The two-problem launch of k_stem_fwd16 assigns its passes CU by CU when the grid is two workgroups per CU (problems split over
    v = 2 u + half, the older half of every CU first in each problem's block numbering: csrc/stem_hip.hip): the headline's rollout
    shape (4096 + 8192 frames: an ODD split, 171 + 341 workgroups), uneven and ragged problems, the larger problem first, a launch
    below one pass per slot, and 768 passes on an even split — every frame of both outputs equals the frame evaluated in a
    1000-frame launch of its own (wave per frame), bit for bit; rows past the problems' ends are not written."""
    from active_tracking_rl_amd import fused
    torch.manual_seed(M0 + M1)
    dev = "cuda"

    class E:
        def __init__(self):
            self.conv1 = torch.nn.Conv2d(1, 16, 3, 2, 1).to(dev)
            self.conv2 = torch.nn.Conv2d(16, 32, 3, 2, 1).to(dev)
            with torch.no_grad():
                self.conv1.weight.mul_(2.0); self.conv2.weight.mul_(3.0); self.conv1.bias.normal_(0, 0.2); self.conv2.bias.normal_(0, 0.2)
    encs = [E(), E()]
    rs = np.random.RandomState(M0)
    x0 = torch.tensor(rs.choice([0, 1, 2, 4], size=(M0, 169)).astype(np.uint8), device=dev)
    x1 = torch.tensor(rs.choice([0, 1, 2, 4], size=(M1, 169)).astype(np.uint8), device=dev)
    o0 = torch.full((M0 + 3, 512), -7.0, device=dev)
    o1 = torch.full((M1 + 3, 512), -7.0, device=dev)
    fused.stem_into2(x0, encs[0], o0[:M0], x1, encs[1], o1[:M1])
    for x, e, o, M in ((x0, encs[0], o0, M0), (x1, encs[1], o1, M1)):
        ref = torch.cat([fused.stem(x[i:i + 1000], e.conv1, e.conv2) for i in range(0, M, 1000)])
        assert torch.equal(o[:M], ref)
        assert bool((o[M:] == -7.0).all())


@pytest.mark.parametrize("M", [16384, 40007])
def test_sixteen_frame_backward_equals_the_wave_per_frame_backward(M):
    """From 16384 frames up atr_stem_backward* runs 16 frames per workgroup pass (k_stem_bwd16: frames on an MFMA dimension, border
    products not issued, the conv1 filter gradient on the matrix cores too); below that one wave per frame (k_stem_bwd). Same
    (x, y, dy) -> the same four gradients up to the order of the fp32 sums: the launch against the sum over chunks of 8000
    frames, 2e-5 of the gradient's max (against F.conv2d's autograd the two differ by the same ReLU-mask flips of the forward:
    that comparison is test_fused_stem_matches_conv2d's)."""
    from active_tracking_rl_amd import fused
    torch.manual_seed(M)
    dev = "cuda"
    conv1 = torch.nn.Conv2d(1, 16, 3, 2, 1).to(dev)
    conv2 = torch.nn.Conv2d(16, 32, 3, 2, 1).to(dev)
    with torch.no_grad():
        conv1.weight.mul_(2.0); conv2.weight.mul_(3.0); conv1.bias.normal_(0, 0.2); conv2.bias.normal_(0, 0.2)
    obs = torch.tensor(np.random.RandomState(M).choice([0, 1, 2, 4], size=(M, 2, 13, 13)).astype(np.uint8), device=dev)
    w1, b1, w2 = conv1.weight.detach().contiguous(), conv1.bias.detach(), conv2.weight.detach().contiguous()
    shapes = (w1.shape, w2.shape)
    for x in (obs[:, 1], obs[:, 0].float()):
        y = fused.stem(x, conv1, conv2)
        dy = torch.randn_like(y)
        big = fused._stem_backward(fused.rows169(x), y, dy, w1, b1, w2, shapes)
        parts = [fused._stem_backward(fused.rows169(x[i:i + 8000]), y[i:i + 8000], dy[i:i + 8000], w1, b1, w2, shapes)
                 for i in range(0, M, 8000)]
        for k, name in enumerate(("dw1", "db1", "dw2", "db2")):
            want = torch.stack([pt[k].double() for pt in parts]).sum(0)
            scale = float(want.abs().max()) + 1e-6
            assert float((big[k].double() - want).abs().max()) <= 2e-5 * scale, (name, M)


@pytest.mark.parametrize("N", [1, 130, 4096])
def test_actor_step_mfma_kernel_matches_lstmcell(N):
    """atr_actor_step (both LSTMCell GEMMs + the cell as one f32-MFMA kernel) against torch.nn.LSTMCell in float64 on
    the masked state, with and without the tracker-action embedding; activated gates checked against the same
    reference. Tolerance 2e-5: an fp32 sum over 384 products in a different order (exact-f32 MFMA, no reduced precision)."""
    from active_tracking_rl_amd import fused
    torch.manual_seed(N)
    F_, R = 256, 128
    lstm = torch.nn.LSTMCell(F_, R).cuda()
    with torch.no_grad():
        lstm.bias_ih.normal_(0, 0.3); lstm.bias_hh.normal_(0, 0.3)
    f = torch.relu(torch.randn(N, F_, device="cuda"))
    h, c = torch.randn(N, R, device="cuda") * 0.5, torch.randn(N, R, device="cuda")
    done = (torch.rand(N, device="cuda") < 0.3).to(torch.uint8)
    emb = torch.randn(4, 4 * R, device="cuda") * 0.2
    a_in = torch.randint(0, 4, (N,), device="cuda")
    bsum = (lstm.bias_ih + lstm.bias_hh).detach()
    for use_emb in (False, True):
        h_out, c_out, acts = torch.empty_like(h), torch.empty_like(c), torch.empty(N, 4 * R, device="cuda")
        fused.actor_step_into(f, h, c, done, lstm, bsum, h_out, c_out, acts, emb=emb if use_emb else None,
                              act_in=a_in if use_emb else None)
        k = (done == 0).double().unsqueeze(1)
        pre = f.double() @ lstm.weight_ih.double().t() + (k * h.double()) @ lstm.weight_hh.double().t() + bsum.double()
        if use_emb:
            pre = pre + emb.double()[a_in]
        gi, gf, gg, go = pre.chunk(4, 1)
        c_ref = torch.sigmoid(gf) * (k * c.double()) + torch.sigmoid(gi) * torch.tanh(gg)
        h_ref = torch.sigmoid(go) * torch.tanh(c_ref)
        torch.testing.assert_close(c_out.double(), c_ref, rtol=2e-5, atol=2e-5)
        torch.testing.assert_close(h_out.double(), h_ref, rtol=2e-5, atol=2e-5)
        acts_ref = torch.cat([torch.sigmoid(gi), torch.sigmoid(gf), torch.tanh(gg), torch.sigmoid(go)], 1)
        torch.testing.assert_close(acts.double(), acts_ref, rtol=2e-5, atol=2e-5)
    # no mask, no gate store
    h_out, c_out = torch.empty_like(h), torch.empty_like(c)
    fused.actor_step_into(f, h, c, None, lstm, bsum, h_out, c_out, None)
    with torch.no_grad():
        h_t, c_t = lstm(f, (h, c))
    torch.testing.assert_close(h_out, h_t, rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(c_out, c_t, rtol=2e-5, atol=2e-5)


def test_two_player_actor_cell_kernel_equals_two_one_player_launches():
    """atr_lstm_cell_forward_act2 (both players' cell + head + draw in one launch, bias added inside, strided per-player
    stores) against two atr_lstm_cell_forward_act launches with consecutive ordinals: same states and gates to fp32
    round-off (the bias joins the sum at a different place) and the same actions."""
    from active_tracking_rl_amd import fused
    torch.manual_seed(3)
    N, R, T = 777, 128, 3
    lins = [torch.nn.Linear(R, 4).cuda() for _ in range(2)]
    ig = torch.randn(2, N, 4 * R, device="cuda")
    hg = torch.randn(2, N, 4 * R, device="cuda")
    bias = [torch.randn(4 * R, device="cuda") * 0.3 for _ in range(2)]
    c_all = torch.randn(2, T, N, R, device="cuda")                 # strided per-player views, as in the rollout cache
    done = (torch.rand(N, device="cuda") < 0.3).to(torch.uint8)
    sa, sb = fused.ActionSampler("cuda", seed=9), fused.ActionSampler("cuda", seed=9)
    sa.begin_block(); sb.begin_block()
    h2, c2, acts2 = torch.empty(2, T, N, R, device="cuda"), torch.empty(2, T, N, R, device="cuda"), torch.empty(2, T, N, 4 * R, device="cuda")
    act2 = torch.empty(2, N, dtype=torch.int64, device="cuda")
    fused.lstm_cell_act2_into(ig, hg, bias, c_all[:, 1], done, h2[:, 2], c2[:, 2], acts2[:, 2], sa, lins, act2)
    for p in range(2):
        h1, c1, a1 = torch.empty(N, R, device="cuda"), torch.empty(N, R, device="cuda"), torch.empty(N, 4 * R, device="cuda")
        act1 = torch.empty(N, dtype=torch.int64, device="cuda")
        fused.lstm_cell_act_into((ig[p] + bias[p]).contiguous(), hg[p].contiguous(), c_all[p, 1].contiguous(), done, h1, c1, a1,
                                 sb, lins[p], act1)
        torch.testing.assert_close(h2[p, 2], h1, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(c2[p, 2], c1, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(acts2[p, 2], a1, rtol=1e-5, atol=1e-6)
        assert float((act2[p] == act1).float().mean()) > 0.995
    assert sa._ordinal == sb._ordinal == 2


@pytest.mark.parametrize("M", [512, 1000, 37, 2048])
def test_pair_linear_matches_float64(M):
    """atr_pair_linear (csrc/pair_gemm_hip.hip: the rollout step's small GEMM pairs in one launch) against float64: the fc +
    ReLU shape of both encoders (K = 512 / 1024 -> 256) and the LSTMCell shape (features x W_ih^T + masked hidden x W_hh^T +
    bias -> 512 gate pre-activations), ragged row counts, strided outputs."""
    from active_tracking_rl_amd import fused
    dev = torch.device("cuda:0")
    torch.manual_seed(M)
    y = [torch.randn(M, 512, device=dev), torch.randn(M, 1024, device=dev)]
    w = [torch.randn(256, 512, device=dev) * 0.05, torch.randn(256, 1024, device=dev) * 0.05]
    b = [torch.randn(256, device=dev), torch.randn(256, device=dev)]
    store = torch.full((2, M, 256), float("nan"), device=dev)
    fused.pair_linear(y, w, [store[0], store[1]], bias=b, relu=True)
    for p in range(2):
        ref = torch.relu(y[p].double() @ w[p].double().t() + b[p].double())
        torch.testing.assert_close(store[p].double(), ref, rtol=2e-5, atol=2e-5)
    f = [store[0], store[1]]
    h = torch.randn(2, M, 128, device=dev)
    wih = [torch.randn(512, 256, device=dev) * 0.1 for _ in range(2)]
    whh = [torch.randn(512, 128, device=dev) * 0.1 for _ in range(2)]
    done = (torch.rand(M, device=dev) < 0.3).to(torch.uint8)
    g = torch.full((2, M, 512), float("nan"), device=dev)
    fused.pair_linear(f, wih, [g[0], g[1]], bias=b_g(dev), a2=[h[0], h[1]], w2=whh, done=done)
    k = (done == 0).double().unsqueeze(1)
    for p in range(2):
        ref = f[p].double() @ wih[p].double().t() + (k * h[p].double()) @ whh[p].double().t() + b_g(dev)[p].double()
        torch.testing.assert_close(g[p].double(), ref, rtol=2e-5, atol=2e-5)
    # no mask, no bias
    fused.pair_linear(f, wih, [g[0], g[1]], a2=[h[0], h[1]], w2=whh)
    for p in range(2):
        ref = f[p].double() @ wih[p].double().t() + h[p].double() @ whh[p].double().t()
        torch.testing.assert_close(g[p].double(), ref, rtol=2e-5, atol=2e-5)


def b_g(dev):
    gen = torch.Generator(device="cpu").manual_seed(3)
    return [torch.randn(512, generator=gen).to(dev), torch.randn(512, generator=gen).to(dev)]


@pytest.mark.parametrize("N,P,T", [(512, 2, 20), (1000, 2, 7), (37, 1, 5), (4096, 2, 20)])
def test_fused_bptt_matches_the_per_step_path(N, P, T):
    """atr_lstm_bptt (csrc/bptt_hip.hip: the whole recurrence backward of a rollout as ONE launch, W_hh in registers, the
    hidden-state gradient never leaving the MFMA accumulators) against the per-step path it replaces (atr_lstm_cell_backward
    + batched GEMM per step), on random stored activations with episode boundaries: dG, the gradient into the initial state
    and dW_hh; ragged row counts, one or two players, a missing head gradient."""
    from active_tracking_rl_amd import fused
    dev = torch.device("cuda:0")
    R = 128
    torch.manual_seed(N + T)
    whh = torch.randn(P, R, 4 * R, device=dev) * 0.08                    # W_hh^T per player
    keep = (torch.rand(T, N, device=dev) > 0.15).float()
    h_all = torch.randn(P, T + 1, N, R, device=dev) * 0.5
    c_all = torch.randn(P, T + 1, N, R, device=dev)
    acts = torch.rand(P, T, N, 4 * R, device=dev)
    acts[:, :, :, 2 * R:3 * R] = acts[:, :, :, 2 * R:3 * R] * 2 - 1       # the g gate is a tanh
    dhs = [torch.randn(T, N, R, device=dev) for _ in range(P)]
    if P == 2 and T == 7:
        dhs[1] = None                                                     # a player whose heads contribute nothing
    res = []
    for flag in (True, False):
        fused.use_fused_bptt = flag
        try:
            res.append(fused._lstm_bptt(whh, keep, h_all, c_all, acts, list(dhs)))
        finally:
            fused.use_fused_bptt = True
    for a_, b_, name in zip(res[0], res[1], ("dG", "dh0", "dc0", "dWhh")):
        scale = float(b_.abs().max())
        torch.testing.assert_close(a_, b_, rtol=2e-4, atol=2e-5 * max(scale, 1.0), msg=lambda m: name + ": " + m)


def test_embed_add_matches_one_hot_linear():
    """fused.embed_add (f + fc_action_tracker(one_hot(a)) as a row gather, csrc/driver_hip.hip) against the tensor expression
    it replaces (TAT.forward, model.py:193-194 of the reference): output, and gradients w.r.t. f, weight and bias."""
    from active_tracking_rl_amd import fused
    dev = torch.device("cuda:0")
    torch.manual_seed(2)
    for rows in (10240, 777):
        lin = torch.nn.Linear(4, 256).to(dev)
        f = torch.randn(rows, 256, device=dev, requires_grad=True)
        a = torch.randint(0, 4, (rows,), device=dev)
        g = torch.randn(rows, 256, device=dev)
        out = fused.embed_add(f, lin, a)
        ref = f + lin(torch.nn.functional.one_hot(a, 4).float())
        torch.testing.assert_close(out, ref, rtol=1e-6, atol=1e-6)
        got = torch.autograd.grad(out, [f, lin.weight, lin.bias], g)
        want = torch.autograd.grad(ref, [f, lin.weight, lin.bias], g)
        torch.testing.assert_close(got[0], want[0], rtol=0, atol=0)
        torch.testing.assert_close(got[1], want[1], rtol=1e-4, atol=1e-3)
        torch.testing.assert_close(got[2], want[2], rtol=1e-4, atol=1e-3)
    # actions read in place from the rollout's [T, players, N] store: the tracker's column as a [T, N] view (no gathered copy)
    T, N = 20, 37
    store = torch.randint(0, 4, (T, 2, N), device=dev)
    lin = torch.nn.Linear(4, 256).to(dev)
    f = torch.randn(T * N, 256, device=dev, requires_grad=True)
    g = torch.randn(T * N, 256, device=dev)
    for col in (0, 1):
        view = store.transpose(1, 2)[:, :, col]                       # [T, N], strides (2 N, 1): what the learner passes
        assert not view.is_contiguous()
        out_v = fused.embed_add(f, lin, view)
        out_f = fused.embed_add(f, lin, view.reshape(T * N))
        assert torch.equal(out_v, out_f)
        gv = torch.autograd.grad(out_v, [lin.weight, lin.bias], g)
        gf = torch.autograd.grad(out_f, [lin.weight, lin.bias], g)
        assert torch.equal(gv[0], gf[0]) and torch.equal(gv[1], gf[1])


@pytest.mark.parametrize("rows_t,N,aux", [(20, 512, True), (7, 37, False)])
def test_two_player_heads_loss_launch_equals_two_one_player_launches(rows_t, N, aux):
    """fused.heads_loss_pair (atr_heads_loss_multi: both players' heads + loss terms in one launch + one reduction launch,
    actions read in place from the [T, players, N] store, statistics scaled and written to one [2, 4] tensor, the objective
    term finished inside the reduction) against two fused.heads_loss calls: terms, statistics, dL/dh and every head gradient
    bit for bit; and fused.heads_values2 against two heads_values calls."""
    from active_tracking_rl_amd import fused
    dev = torch.device("cuda:0")
    torch.manual_seed(rows_t + N)
    T, R, A = rows_t, 128, 4
    rows = T * N
    actors = [torch.nn.Linear(R, A).to(dev) for _ in range(2)]
    critics = [torch.nn.Linear(R, 1).to(dev) for _ in range(2)]
    auxl = torch.nn.Linear(R, 1).to(dev) if aux else None
    hs = [torch.randn(rows, R, device=dev, requires_grad=True) for _ in range(2)]
    store = torch.randint(0, A, (T, 2, N), device=dev)
    actions = store.transpose(1, 2)                                     # [T, N, players] view
    ret, gae = torch.randn(T, N, 2, 1, device=dev), torch.randn(T, N, 2, 1, device=dev)
    val = torch.zeros(T + 1, N, 2, 1, device=dev)
    rew = torch.randn(T, N, 2, 1, device=dev)
    v2 = torch.zeros_like(val)
    fused.heads_values2([h.detach() for h in hs], critics, val)
    for p in range(2):
        fused.heads_values(hs[p].detach(), critics[p], v2, p)
    assert torch.equal(val, v2)
    scale, w_ent = [1.0 / N, 0.0 if not aux else 1.0 / N], [0.01, 0.2]
    cfg = [dict(actions=actions[:, :, p], ret=ret, gae=gae, val=val, off=p, r_aux=rew if (aux and p == 1) else None, aux_off=0,
                scale=scale[p], scale_aux=(1.0 / N if (aux and p == 1) else 0.0), w_ent=w_ent[p], stats_scale=1.0 / N)
           for p in range(2)]
    l0, l1, st = fused.heads_loss_pair(hs, actors, critics, [None, auxl], cfg)
    prm = [x for p in range(2) for x in (actors[p].weight, actors[p].bias, critics[p].weight, critics[p].bias)]
    if aux:
        prm += [auxl.weight, auxl.bias]
    one = torch.ones((), device=dev)
    got = torch.autograd.grad([l0, l1], hs + prm, grad_outputs=[one, one])
    terms, stats = [], []
    for p in range(2):
        lp, sp = fused.heads_loss(hs[p], actors[p], critics[p], auxl if (aux and p == 1) else None,
                                  actions[:, :, p].reshape(rows), ret, gae, val, p, rew if (aux and p == 1) else None, 0,
                                  scale[p], 1.0 / N if (aux and p == 1) else 0.0, w_ent[p], unit_coeff=True)
        terms.append(lp)
        stats.append(sp)
    want = torch.autograd.grad(terms[0] + terms[1], hs + prm)
    assert torch.equal(l0, terms[0]) and torch.equal(l1, terms[1])
    assert torch.equal(st, torch.stack(stats, 0) * (1.0 / N))
    for a_, b_ in zip(got, want):
        assert torch.equal(a_, b_)


@pytest.mark.parametrize("A,aux", [(4, True), (8, False)])
def test_heads_loss_matches_autograd_in_float64(A, aux):
    """fused.heads_loss (csrc/heads_hip.hip; the 4-action and the 8-action ('Moore') instantiation) against the loss terms of
    Agent.optimize written with torch ops in float64 (player_util.py:118-154 of the reference: policy term -log p(a) * gae -
    w_ent * entropy, value term 0.5 (R - V)^2 weighted 0.5, aux |pred - r|): objective term, statistics, dL/dh and every head
    gradient."""
    from active_tracking_rl_amd import fused
    dev = torch.device("cuda:0")
    torch.manual_seed(A)
    T, N, R = 6, 53, 128
    rows = T * N
    actor, critic = torch.nn.Linear(R, A).to(dev), torch.nn.Linear(R, 1).to(dev)
    auxl = torch.nn.Linear(R, 1).to(dev) if aux else None
    h = torch.randn(rows, R, device=dev, requires_grad=True)
    actions = torch.randint(0, A, (rows,), device=dev)
    ret, gae, rew = (torch.randn(T, N, 1, 1, device=dev) for _ in range(3))
    val = torch.zeros(T + 1, N, 1, 1, device=dev)
    fused.heads_values(h.detach(), critic, val, 0)
    scale, scale_aux, w_ent = 1.0 / N, (1.0 / N if aux else 0.0), 0.05
    lp, st = fused.heads_loss(h, actor, critic, auxl, actions, ret, gae, val, 0, rew if aux else None, 0, scale, scale_aux, w_ent,
                              unit_coeff=True)
    prm = [actor.weight, actor.bias, critic.weight, critic.bias] + ([auxl.weight, auxl.bias] if aux else [])
    got = torch.autograd.grad(lp, [h] + prm)
    # float64 reference
    hd = h.detach().double().requires_grad_(True)
    P = [p.detach().double().requires_grad_(True) for p in prm]
    logits = hd @ P[0].t() + P[1]
    v = (hd @ P[2].t() + P[3]).reshape(-1)
    logp = torch.log_softmax(logits, 1)
    prob = logp.exp()
    ent = -(logp * prob).sum(1)
    lpa = logp.gather(1, actions.view(-1, 1)).reshape(-1)
    R_, G_ = ret.double().reshape(-1), gae.double().reshape(-1)
    pol = (-lpa * G_ - w_ent * ent).sum()
    vl = (0.5 * (R_ - v) ** 2).sum()
    obj = scale * (pol + 0.5 * vl)
    aux_sum = torch.zeros((), dtype=torch.float64, device=dev)
    if aux:
        pred = (hd @ P[4].t() + P[5]).reshape(-1)
        aux_sum = (pred - rew.double().reshape(-1)).abs().sum()
        obj = obj + scale_aux * aux_sum
    want = torch.autograd.grad(obj, [hd] + P)
    assert abs(float(lp.detach()) - float(obj.detach())) < 1e-4 * max(1.0, abs(float(obj.detach())))
    ref_stats = torch.stack([pol, vl, ent.sum(), aux_sum]).float()
    torch.testing.assert_close(st, ref_stats, rtol=2e-4, atol=2e-3)
    for g_, w_ in zip(got, want):
        torch.testing.assert_close(g_.double().reshape(w_.shape), w_, rtol=2e-4, atol=2e-5)
