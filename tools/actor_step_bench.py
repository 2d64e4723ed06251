"""The actor's LSTMCell step: one MFMA kernel (atr_actor_step) vs the unfused sequence (input GEMM + hidden GEMM + cell)."""
import torch
from active_tracking_rl_amd import fused

e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def t_us(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for N in (1024, 2048, 4096, 8192):
    lstm = torch.nn.LSTMCell(256, 128).cuda()
    f = torch.relu(torch.randn(N, 256, device="cuda"))
    h, c = torch.randn(N, 128, device="cuda"), torch.randn(N, 128, device="cuda")
    done = torch.zeros(N, dtype=torch.uint8, device="cuda")
    bsum = (lstm.bias_ih + lstm.bias_hh).detach()
    ho, co, acts = torch.empty_like(h), torch.empty_like(c), torch.empty(N, 512, device="cuda")
    wih_t, whh_t = lstm.weight_ih.detach().t().contiguous(), lstm.weight_hh.detach().t().contiguous()

    def unfused():
        ig = torch.addmm(bsum, f, wih_t)
        hg = torch.mm(h, whh_t)
        fused.lstm_cell_into(ig, hg, c, done, ho, co, acts)

    with torch.no_grad():
        tu = t_us(unfused)
        tf = t_us(lambda: fused.actor_step_into(f, h, c, done, lstm, bsum, ho, co, acts))
    gf = 2.0 * N * 512 * 384 / 1e9
    print("N=%5d  unfused %7.1f us   fused %7.1f us (%.0f TFLOP/s)" % (N, tu, tf, gf / tf / 1e-3))
