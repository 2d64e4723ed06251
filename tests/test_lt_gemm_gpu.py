"""-m gpu: the pieces of the one-GEMM LSTMCell path of the rollout step (round 4), each against plain PyTorch fp32:
atr_linear (hipBLASLt called directly: strided output, fused bias + ReLU, batched), atr_relu_backward_ld, atr_embed_add_ld,
the grouped weight-gradient launch with a row-strided operand, and k_act_step's masked hidden-row output.
Reference layers: perception.py:81,90 (fc + ReLU), model.py:110,137,172,203 (nn.LSTMCell)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("M", [96, 1024, 4096])
def test_linear_lt_strided_output_bias_relu_matches_torch(M):
    from active_tracking_rl_amd import fused
    torch.manual_seed(3)
    for K in (512, 1024):
        a = torch.randn(M, K, device=DEV)
        w = torch.randn(256, K, device=DEV) * 0.05
        b = torch.randn(256, device=DEV)
        rows = torch.full((M, 384), float("nan"), device=DEV)
        ws = torch.empty(32 << 20, dtype=torch.uint8, device=DEV)
        fused.linear_lt(a, w, rows[:, :256], bias=b, relu=True, workspace=ws)
        want = torch.relu(torch.nn.functional.linear(a.double(), w.double(), b.double())).float()
        torch.testing.assert_close(rows[:, :256], want, rtol=2e-5, atol=2e-5)
        assert torch.isnan(rows[:, 256:]).all(), "columns beyond the output block were written"
        info = fused.linear_lt_info(a, w, rows[:, :256], bias=b, relu=True, workspace=ws)
        assert info is not None and info["tuned"] and info["candidates"] >= 1
        # second call: same kernel, bit-identical result; and the library's own workspace
        first = rows[:, :256].clone()
        fused.linear_lt(a, w, rows[:, :256], bias=b, relu=True, workspace=ws)
        assert torch.equal(first, rows[:, :256])
        dense = torch.empty(M, 256, device=DEV)
        fused.linear_lt(a, w, dense, bias=None, relu=False)
        torch.testing.assert_close(dense, torch.nn.functional.linear(a.double(), w.double()).float(), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("N", [130, 2048, 4096])
def test_linear_lt_batched_gate_gemm_matches_torch(N):
    """Both players' LSTMCell GEMMs as one batched product over [features | k h] rows taken from a [2, T + 1, N, 384] store."""
    from active_tracking_rl_amd import fused
    torch.manual_seed(4)
    store = torch.randn(2, 3, N, 384, device=DEV)
    w_ih, w_hh = torch.randn(2, 512, 256, device=DEV) * 0.05, torch.randn(2, 512, 128, device=DEV) * 0.05
    w_cat = torch.cat([w_ih, w_hh], 2).contiguous()
    gates = torch.empty(2, N, 512, device=DEV)
    x = store[:, 1]
    assert not x.is_contiguous()
    fused.linear_lt(x, w_cat, gates)
    want = (torch.bmm(x[:, :, :256].double(), w_ih.double().transpose(1, 2))
            + torch.bmm(x[:, :, 256:].double(), w_hh.double().transpose(1, 2))).float()
    torch.testing.assert_close(gates, want, rtol=2e-5, atol=2e-5)
    # inside a hipGraph (what the rollout driver does): the captured launch replays to the same bits
    g = torch.cuda.CUDAGraph()
    out2 = torch.empty_like(gates)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fused.linear_lt(x, w_cat, out2)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    out2.zero_()
    with torch.cuda.graph(g):
        fused.linear_lt(x, w_cat, out2)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out2, gates)


@pytest.mark.parametrize("N", [1024, 4096, 1536])
def test_gate_gemm_kernel_does_not_depend_on_where_its_batches_land(N):
    """The gate GEMM writes a [2, N, 512] scratch tensor or slot t of a [2, T, N, 512] rollout store — the same products. The
    kernel is chosen per problem FAMILY (fused._lt_family: the key without the output's batch stride), recorded or not (1536
    rows: no record, the first problem of the family is timed and its siblings follow), so the outputs are equal bit for bit:
    otherwise two runs of one seed part ways at the first sampled action (store_preacts on / off, another rollout length)."""
    from active_tracking_rl_amd import fused
    torch.manual_seed(5)
    store = torch.randn(2, 3, N, 384, device=DEV)
    w_cat = (torch.randn(2, 512, 384, device=DEV) * 0.05).contiguous()
    x = store[:, 1]
    outs, sols = [], []
    for T in (None, 20, 8):
        dst = torch.empty(2, N, 512, device=DEV) if T is None else torch.empty(2, T, N, 512, device=DEV)[:, T // 2]
        fused.linear_lt(x, w_cat, dst)
        info = fused.linear_lt_info(x, w_cat, dst)
        outs.append(dst.clone())
        sols.append(info["solution"])
    assert sols[0] >= 0 and sols[0] == sols[1] == sols[2], sols
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_relu_backward_and_embed_add_on_strided_features():
    from active_tracking_rl_amd import fused
    torch.manual_seed(5)
    rows = 4096 * 3 + 5
    store = torch.randn(rows, 384, device=DEV)
    f = store[:, :256]
    df = torch.randn(rows, 256, device=DEV)
    got = fused.relu_backward_ld(df, f)
    assert torch.equal(got, torch.ops.aten.threshold_backward(df, f.contiguous(), 0.0))
    lin = torch.nn.Linear(4, 256).to(DEV)
    acts = torch.randint(0, 4, (rows,), device=DEV)
    a = fused.embed_add(f, lin, acts)
    b = fused.embed_add(f.contiguous(), lin, acts)
    assert torch.equal(a, b)
    want = f + lin(torch.nn.functional.one_hot(acts, 4).float())
    torch.testing.assert_close(a, want, rtol=1e-6, atol=1e-6)


def test_grouped_weight_gradients_with_a_strided_operand():
    """dW = dG^T f for f read in place from [f | h] rows (ld2 = 384): same bits as the dense copy of f."""
    from active_tracking_rl_amd import fused
    from active_tracking_rl_amd.shared_optim import FlatParams
    torch.manual_seed(6)
    K = 8192
    store = torch.randn(K, 384, device=DEV)
    dG = torch.randn(K, 512, device=DEV)
    res = []
    for strided in (True, False):
        w = torch.nn.Parameter(torch.zeros(512, 256, device=DEV))
        b = torch.nn.Parameter(torch.zeros(512, device=DEV))
        bucket = FlatParams([w, b])
        q = fused.DeferredWeightGrads(bucket)
        x2 = store[:, :256] if strided else store[:, :256].contiguous()
        r = q.add(dG, x2, w, biases=(b,))
        assert r is not None
        q.flush()
        torch.cuda.synchronize()
        res.append((r[0].clone(), r[1][0].clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    torch.testing.assert_close(res[0][0], (dG.double().t() @ store[:, :256].double()).float(), rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(res[0][1], dG.double().sum(0).float(), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("n", [512, 257])
def test_act_env_step_writes_the_masked_hidden_rows(n):
    """k_act_step with hm_out: the next step's GEMM rows receive h where the env goes on and 0 where this step's done flag is
    set (player_util.py:98-102: a finished env restarts from a zero hidden state); everything else as without hm_out."""
    from active_tracking_rl_amd import fused, vec_env
    dev = torch.device(DEV)
    R, A = 128, 4
    torch.manual_seed(7)
    envs = [vec_env.VecTrack2D("Track2D-BlockPartialPZR-v0", num_envs=n, seed=5) for _ in range(2)]
    for e in envs:
        e.reset()
    actors = [torch.nn.Linear(R, A).to(dev) for _ in range(2)]
    samplers = [fused.ActionSampler(dev, seed=9) for _ in range(2)]
    emb = torch.randn(A, 4 * R, device=dev) * 0.5
    bias = [torch.randn(4 * R, device=dev) * 0.1 for _ in range(2)]
    c_prev = torch.randn(2, n, R, device=dev)
    outs = [(torch.zeros(n, 2, 13, 13, dtype=torch.uint8, device=dev), torch.zeros(n, 2, device=dev),
             torch.zeros(n, dtype=torch.uint8, device=dev)) for _ in range(2)]
    n_done = 0
    done_prev = None
    for t in range(40):
        ig = torch.randn(2, n, 4 * R, device=dev)
        res = []
        for k in range(2):
            h, c = torch.empty(2, n, R, device=dev), torch.empty(2, n, R, device=dev)
            acts = torch.empty(2, n, 4 * R, device=dev)
            actions = torch.empty(2, n, dtype=torch.int64, device=dev)
            rows = torch.full((2, n, 384), float("nan"), device=dev)
            samplers[k].begin_block()
            fused.act_env_step(envs[k], [ig[0], ig[1]], None, bias, [c_prev[0], c_prev[1]], done_prev, [h[0], h[1]],
                               [c[0], c[1]], [acts[0], acts[1]], samplers[k], actors, actions, emb=emb, env_out=outs[k],
                               hm_out=[rows[0][:, 256:], rows[1][:, 256:]] if k == 0 else None)
            samplers[k].end_block()
            res.append((h, c, acts, actions, rows))
        for a_, b_ in zip(res[0][:4], res[1][:4]):
            assert torch.equal(a_, b_), t
        for a_, b_ in zip(outs[0], outs[1]):
            assert torch.equal(a_, b_), t
        keep = (outs[0][2] == 0).float().view(1, n, 1)
        rows = res[0][4]
        assert torch.equal(rows[:, :, 256:], res[0][0] * keep), t
        assert torch.isnan(rows[:, :, :256]).all()
        n_done += int(outs[0][2].sum().item())
        done_prev = outs[0][2].clone()
        c_prev = res[0][1]
    assert n_done > 20
    for e in envs:
        e.close()


@pytest.mark.parametrize("n", [256, 1024])
def test_rollout_first_launch_makes_the_per_rollout_constants(n):
    """atr_rollout_begin2: LSTM state and observation into the stores AND, in the same launch, b_ih + b_hh, the tracker-action
    embedding projected through the target's W_ih (model.py:193-194), the concatenated LSTMCell weight, the hidden columns of
    slot 0 of the [features | k h] rows, the draw counter's bump — against the tensor expressions they replace."""
    from active_tracking_rl_amd.train import default_args, make_player, rollout
    args = default_args(env="Track2D-BlockPartialPZR-v0", num_envs=n, num_steps=5, network="tat-maze-lstm", seed=3)
    player, opt = make_player(args, torch.device(DEV))
    rollout(player, args.num_steps)                       # (creates the sampler, leaves a non-trivial LSTM state behind)
    player.optimize(None, opt, player.model, -1, torch.device(DEV))
    m = player.model
    before = int(m._sampler.counter.item())
    hxs, cxs = player.hxs.clone(), player.cxs.clone()
    player.begin_rollout(args.num_steps)
    torch.cuda.synchronize()
    c = player._cache
    assert c.consts is None and int(m._sampler.counter.item()) == before + 1 and m._sampler._ordinal == 0
    for p, l in enumerate((m.player0.lstm, m.player1.lstm)):
        assert torch.equal(c.bsum[p], l.bias_ih + l.bias_hh)
    fa = m.player1.fc_action_tracker
    want = (fa.weight.t().double() + fa.bias.double()) @ m.player1.lstm.weight_ih.t().double()
    torch.testing.assert_close(c.emb_ih, want.float(), rtol=1e-5, atol=1e-6)
    assert torch.equal(c.h_all[:, 0], hxs.transpose(0, 1)) and torch.equal(c.c_all[:, 0], cxs.transpose(0, 1))
    assert torch.equal(player._buf[0][0].reshape(-1), player.state.reshape(-1))
    if n >= m.cat_gemm_min_rows:
        assert c.fh_all is not None
        for p, l in enumerate((m.player0.lstm, m.player1.lstm)):
            assert torch.equal(c.w_cat[p], torch.cat([l.weight_ih, l.weight_hh], 1))
        assert torch.equal(c.fh_all[:, 0, :, 256:], c.h_all[:, 0])
    else:
        assert c.fh_all is None
    player.end_rollout() if False else None
    player.env.close()


def _problem(m=640, seed=0):
    torch.manual_seed(seed)
    dev = "cuda"
    a = torch.randn(m, 512, device=dev)
    w = torch.randn(256, 512, device=dev)
    b = torch.randn(256, device=dev)
    rows = torch.zeros(m, 384, device=dev)
    ws = torch.empty(32 << 20, dtype=torch.uint8, device=dev)
    return a, w, b, rows, ws


def test_recorded_choice_is_checked_against_the_loaded_library():
    """A recorded kernel choice is (position in the heuristic's list, solution index). An out-of-range position is refused
    (the list is timed instead); a position whose kernel is not the recorded solution is not believed — the list is searched
    for the solution; a solution the list does not hold is refused. The product is right in every case."""
    from active_tracking_rl_amd import fused
    # (shapes no other test or the tuning file uses: the plans below start from nothing)
    a, w, b, rows, ws = _problem(m=608)
    want = torch.relu(torch.nn.functional.linear(a.double(), w.double(), b.double())).float()
    fused.linear_lt_set_choice(a, w, rows[:, :256], index=100000, bias=b, relu=True, workspace=ws)
    fused.linear_lt(a, w, rows[:, :256], bias=b, relu=True, workspace=ws)
    info = fused.linear_lt_info(a, w, rows[:, :256], bias=b, relu=True, workspace=ws)
    assert fused.LT_SOURCES[info["source"]] == "refused-timed" and 0 <= info["chosen"] < info["candidates"] and info["tuned"]
    torch.testing.assert_close(rows[:, :256], want, rtol=2e-4, atol=2e-3)
    if info["solution"] < 0:
        pytest.skip("this hipBLASLt build does not export hipblaslt_ext::getIndexFromAlgo")
    best_idx, best_sol = info["chosen"], info["solution"]
    # the right solution recorded under a wrong position: found by its identity
    wrong_pos = (best_idx + 1) % info["candidates"]
    fused.linear_lt_set_choice(a, w, rows[:, :256], index=wrong_pos, solution=best_sol, bias=b, relu=True, workspace=ws)
    rows.zero_()
    fused.linear_lt(a, w, rows[:, :256], bias=b, relu=True, workspace=ws)
    info = fused.linear_lt_info(a, w, rows[:, :256], bias=b, relu=True, workspace=ws)
    assert fused.LT_SOURCES[info["source"]] == "recorded" and info["solution"] == best_sol and info["chosen"] == best_idx
    torch.testing.assert_close(rows[:, :256], want, rtol=2e-4, atol=2e-3)
    # a solution this list does not hold: refused, timed
    fused.linear_lt_set_choice(a, w, rows[:, :256], index=0, solution=2 ** 30, bias=b, relu=True, workspace=ws)
    rows.zero_()
    fused.linear_lt(a, w, rows[:, :256], bias=b, relu=True, workspace=ws)
    info = fused.linear_lt_info(a, w, rows[:, :256], bias=b, relu=True, workspace=ws)
    assert fused.LT_SOURCES[info["source"]] == "refused-timed" and info["solution"] != 2 ** 30
    torch.testing.assert_close(rows[:, :256], want, rtol=2e-4, atol=2e-3)


def test_tuning_file_of_another_library_build_is_ignored(tmp_path, monkeypatch):
    """lt_tuning_gfx950.json is only believed when it was made with the hipBLASLt build that is loaded (version + revision);
    otherwise every problem is timed at first use and the status line says why. The committed file must match this image."""
    import json
    from active_tracking_rl_amd import fused
    have = fused.lt_library()
    assert have["version"] > 0
    committed = json.load(open(fused.LT_TUNING_FILE))
    assert committed["hipblaslt"] == {"version": have["version"], "git": have["git"]}, \
        "regenerate lt_tuning_gfx950.json with tools/tune_lt.py for this image's hipBLASLt"
    a, w, b, rows, ws = _problem(m=672)
    g = fused._linear_args(a, w, rows[:, :256], b, True, ws)
    key = fused._lt_key(g)
    for hip, expect in (({"version": have["version"] + 1, "git": have["git"]}, False),
                        ({"version": have["version"], "git": have["git"] + "x"}, False),
                        ({"version": have["version"], "git": have["git"]}, True)):
        f = tmp_path / "lt.json"
        f.write_text(json.dumps({"hipblaslt": hip, "choices": {key: {"index": 0, "solution": -1}}}))
        monkeypatch.setattr(fused, "LT_TUNING_FILE", str(f))
        monkeypatch.setattr(fused, "_lt_choices", None)
        ch = fused.lt_choices()
        assert (key in ch) == expect
        assert fused.lt_tuning_status().startswith("recorded" if expect else "timed at first use (tuning file made with")
    monkeypatch.setattr(fused, "_lt_choices", None)
