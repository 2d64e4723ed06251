"""Which of the learner's kernels hold up the rollout chain when they run next to it (pipelined schedule)? The rollout graph
of an N-env shard replayed on one stream while ONE learner kernel class loops on another.   python tools/corun_kernels.py N"""
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")   # (see bench.py: room for the CU-partitioned stream pair)

import sys

import torch

from active_tracking_rl_amd import fused
from active_tracking_rl_amd.shared_optim import FlatParams
from active_tracking_rl_amd.train import PipelinedIteration, default_args, make_player

n = int(sys.argv[1])
dev = torch.device("cuda:0")
args = default_args(num_envs=n)
player, opt = make_player(args, dev)
g = PipelinedIteration(player, opt, args)
g.tune_streams()
for _ in range(6):
    g.run()
g.finish()
torch.cuda.synchronize()
g_r = g.graphs[(g.mode0, 0)][0]
K, T = 20 * n, 20
# operands of the learner's shapes
dG = torch.randn(2, K, 512, device=dev)
feat = torch.randn(K, 256, device=dev)
h = torch.randn(K, 128, device=dev)
dpre = torch.randn(K, 256, device=dev)
y1 = torch.randn(K, 1024, device=dev)
wih = torch.randn(512, 256, device=dev)
wfc1 = torch.randn(256, 1024, device=dev)
wfc0 = torch.randn(256, 512, device=dev)
keep = torch.ones(T, n, device=dev)
params = [torch.nn.Parameter(torch.randn(*sh, device=dev)) for sh in ((512, 256), (512, 128), (256, 1024), (256,))]
bucket = FlatParams(params)
conv1 = torch.nn.Conv2d(1, 16, 3, 2, 1).to(dev)
conv2 = torch.nn.Conv2d(16, 32, 3, 2, 1).to(dev)
xs = torch.randint(0, 5, (2 * K, 169), device=dev).to(torch.uint8)
ys = torch.randn(2 * K, 512, device=dev)
dys = torch.randn(2 * K, 512, device=dev)
h_all = torch.randn(2, T + 1, n, 128, device=dev)
c_all = torch.randn(2, T + 1, n, 128, device=dev)
acts = torch.rand(2, T, n, 512, device=dev)
whh = [torch.randn(512, 128, device=dev) for _ in range(2)]
dhs = [torch.randn(T, n, 128, device=dev) for _ in range(2)]
out_a = torch.empty(K, 256, device=dev)
out_b = torch.empty(K, 1024, device=dev)


def k_gemm_tn():
    with fused.deferred_weight_grads(bucket) as q:
        q.add(dG[0], feat, params[0])
        q.add(dG[0], h, params[1])
        q.add(dpre, y1, params[2], biases=(params[3],))
        q.flush()


def k_gemm_tn_corun():
    with fused.gemm_tn_corun(True):
        k_gemm_tn()


def k_bptt():
    fused._lstm_bptt(None, keep, h_all, c_all, acts, dhs, whh_nn=whh, want_dwhh=False)


def k_stem_bwd():
    fused._stem_backward(xs, ys, dys, conv1.weight.detach().contiguous(), conv1.bias.detach(), conv2.weight.detach().contiguous(),
                         (conv1.weight.shape, conv2.weight.shape))


def k_dx_lstm():
    torch.mm(dG[0], wih, out=out_a)


def k_dx_fc():
    torch.mm(dpre, wfc1, out=out_b)


def k_relu_bwd():
    torch.ops.aten.threshold_backward(dpre, feat, 0.0)


E = lambda: torch.cuda.Event(enable_timing=True)
for name, load in (("nothing", None), ("grouped weight-gradient GEMM (k_gemm_tn + reduce)", k_gemm_tn),
                   ("the same in its co-run form (1 workgroup per CU)", k_gemm_tn_corun), ("k_lstm_bptt", k_bptt),
                   ("k_stem_bwd (+ reduce)", k_stem_bwd), ("dX GEMM of the LSTM (library)", k_dx_lstm),
                   ("dX GEMM of the target's fc (library)", k_dx_fc), ("ReLU backward (torch elementwise)", k_relu_bwd)):
    if load is not None:
        load()
    torch.cuda.synchronize()
    if load is not None:        # how long one call takes alone
        e0, e1 = E(), E()
        e0.record()
        for _ in range(10):
            load()
        e1.record()
        torch.cuda.synchronize()
        alone = e0.elapsed_time(e1) * 100
        per_rollout = max(2, int(900.0 / alone) + 1)
    else:
        alone, per_rollout = 0.0, 0
    e0, e1 = E(), E()
    reps = 20
    with torch.cuda.stream(g.sL):
        for _ in range(per_rollout):
            load()
    with torch.cuda.stream(g.sR):
        e0.record(g.sR)
    for _ in range(reps):
        with torch.cuda.stream(g.sL):
            for _ in range(per_rollout):
                load()
        with torch.cuda.stream(g.sR):
            g_r.replay()
    with torch.cuda.stream(g.sR):
        e1.record(g.sR)
    torch.cuda.synchronize()
    print("n=%d rollout graph next to %-52s (%7.1f us per call alone): %8.1f us per rollout" % (
        n, name, alone, e0.elapsed_time(e1) * 1e3 / reps), flush=True)
player.env.close()
