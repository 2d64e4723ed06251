"""SharedAdam numerics on one flat fp32 parameter bucket (replaces shared_optim.py:90-175 + utils.py:36-44).

The reference keeps Adam state in OS shared memory and lets 16 Hogwild workers race on it. The MI355X design
is synchronous data parallel: every rank holds a bit-identical replica whose parameters AND gradients are
views into two flat buffers, so the only exchange is ONE RCCL all-reduce of the flat gradient bucket
(801 291 fp32 for tat-maze-lstm) followed by an identical update on every replica.

Update rule = SharedAdam.step (shared_optim.py:122-175) with its non-standard details kept:
  amsgrad, eps = 1e-3 added AFTER sqrt(max second moment), bias corrections folded into
  step_size = lr * sqrt(1 - b2^t) / (1 - b1^t), no weight decay by default.
The step counter and the two bias-correction powers live on the device, so the whole update is
hipGraph-capturable (no host sync, no Python-side scalar math per step).
"""
import torch


class FlatParams(object):
    """Re-homes a module's parameters (and their .grad) as views of two contiguous fp32 buffers. Every parameter starts
    on a 16-byte boundary (its slot is padded to a multiple of 4 floats: the two players' 1-element critic / aux biases
    would otherwise leave everything behind them dword-aligned only, and the kernels fetch weight rows 16 bytes at a
    time); the padding elements are zero in both buffers and stay zero under every update rule used here."""

    ALIGN = 4   # floats

    def __init__(self, params):
        self.params = [p for p in params]
        ref = self.params[0]
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        total = off
        self.flat = torch.zeros(total, dtype=torch.float32, device=ref.device)
        self.grad = torch.zeros(total, dtype=torch.float32, device=ref.device)
        for p, off in zip(self.params, self.offsets):
            n = p.numel()
            self.flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + n].view_as(p.data)
            p.grad = self.grad[off:off + n].view_as(p.data)
        self.numel = total
        self.param_numel = sum(p.numel() for p in self.params)
        # stand-in for gradients autograd reports as unused and for the padding (set_grads): ordinary memory zeroed once,
        # allocated HERE and not on first use — first use can fall inside a hipGraph capture, whose private pool and memset
        # node would then belong to that one graph while the other per-mode graphs read the buffer too
        self._zeros = torch.zeros(max(p.numel() for p in self.params) + self.ALIGN, dtype=torch.float32, device=ref.device)
        self._views = None

    def zero_grad(self):
        self.grad.zero_()
        for p, off in zip(self.params, self.offsets):  # re-attach in case autograd replaced .grad
            n = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.grad[off:off + n].data_ptr():
                p.grad = self.grad[off:off + n].view_as(p.data)

    def grad_views(self):
        return [self.grad[off:off + p.numel()].view_as(p.data) for p, off in zip(self.params, self.offsets)]

    def set_grads(self, grads):
        """Store freshly computed gradients (torch.autograd.grad output, None = unused) in the bucket with ONE launch —
        instead of zeroing the bucket and letting autograd accumulate into ~40 views one add kernel at a time. Gradients that
        already ARE their slice of the bucket (the grouped weight-gradient launch writes there: fused.DeferredWeightGrads)
        are left alone; the others go in as one segment-scatter launch (csrc/driver_hip.hip; CPU buckets: one cat)."""
        if self._views is None:
            self._views = self.grad_views()
        in_place = [g is not None and g.data_ptr() == v.data_ptr() for g, v in zip(grads, self._views)]
        if self.grad.is_cuda:
            import ctypes as C
            from . import fused
            L = fused.lib()
            todo = [(g, off, p.numel()) for g, p, off, ip in zip(grads, self.params, self.offsets, in_place) if not ip]
            keep = []
            for i in range(0, len(todo), 64):
                part = todo[i:i + 64]
                n = len(part)
                src, dst, cnt = (C.c_void_p * n)(), (C.c_longlong * n)(), (C.c_int * n)()
                for q, (g, off, numel) in enumerate(part):
                    if g is not None:
                        g = g.detach()
                        if g.dtype != torch.float32 or not g.is_contiguous():
                            g = g.to(torch.float32).contiguous()
                        keep.append(g)
                        src[q] = g.data_ptr()
                    else:
                        src[q] = None
                    dst[q], cnt[q] = off, numel
                rc = L.atr_scatter_segments(src, dst, cnt, n, C.c_void_p(self.grad.data_ptr()),
                                            C.c_void_p(torch.cuda.current_stream(self.grad.device).cuda_stream))
                if rc != 0:
                    raise RuntimeError("atr_scatter_segments failed (%d)" % rc)
        else:
            assert not any(in_place)
            flat = []
            for g, p in zip(grads, self.params):
                n = p.numel()
                flat.append(g.reshape(-1) if g is not None else self._zeros[:n])
                pad = -n % self.ALIGN
                if pad:
                    flat.append(self._zeros[:pad])
            torch.cat(flat, out=self.grad)
        for p, v in zip(self.params, self._views):
            if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                p.grad = v


class SharedAdam(torch.optim.Optimizer):
    """Adam+AMSGrad with the reference's SharedAdam numerics over a FlatParams bucket."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-3, weight_decay=0, amsgrad=True, torch_eps=False):
        params = list(params)
        self.torch_eps = bool(torch_eps)   # torch.optim.Adam's placement of eps / bias corrections (see local_optimizer)
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad)
        super(SharedAdam, self).__init__(params, defaults)
        self.bucket = FlatParams(params)
        dev = self.bucket.flat.device
        z = lambda: torch.zeros_like(self.bucket.flat)
        self.exp_avg, self.exp_avg_sq, self.max_exp_avg_sq = z(), z(), z()
        # float64 scalars on the device: step, beta1^t, beta2^t (one 3-vector, so that the fused step sees them together)
        self._scalars = torch.tensor([0.0, 1.0, 1.0], dtype=torch.float64, device=dev)
        self.step_t, self.b1_pow, self.b2_pow = self._scalars[0], self._scalars[1], self._scalars[2]
        self._step_size = torch.zeros(2, dtype=torch.float32, device=dev)
        self.fused = True       # on the GPU: the whole update as one elementwise launch (csrc/driver_hip.hip)

    def share_memory(self):
        """API compatibility with main.py:93 — there is no shared-memory state in the data-parallel design."""
        return self

    def zero_grad(self, set_to_none=False):
        self.bucket.zero_grad()

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        g = self.param_groups[0]
        beta1, beta2 = g['betas']
        grad, p = self.bucket.grad, self.bucket.flat
        if self.fused and p.is_cuda:
            from . import fused
            fused.adam_step(p, grad, self.exp_avg, self.exp_avg_sq, self.max_exp_avg_sq if g['amsgrad'] else None,
                            self._scalars, self._step_size, g['lr'], beta1, beta2, g['eps'], g['weight_decay'],
                            torch_eps=self.torch_eps)
            return loss
        if g['weight_decay'] != 0:
            grad = grad.add(p, alpha=g['weight_decay'])
        self.step_t += 1
        self.b1_pow *= beta1
        self.b2_pow *= beta2
        self.exp_avg.mul_(beta1).add_(grad, alpha=1 - beta1)
        self.exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
        second = self.exp_avg_sq
        if g['amsgrad']:
            torch.max(self.max_exp_avg_sq, self.exp_avg_sq, out=self.max_exp_avg_sq)
            second = self.max_exp_avg_sq
        if self.torch_eps:      # torch.optim.Adam: sqrt(v / (1 - b2^t)) + eps, step size lr / (1 - b1^t)
            denom = (second.sqrt() * (1.0 / torch.sqrt(1 - self.b2_pow)).to(torch.float32)).add_(g['eps'])
            step_size = (g['lr'] / (1 - self.b1_pow)).to(torch.float32)
        else:
            denom = second.sqrt().add_(g['eps'])
            step_size = (g['lr'] * torch.sqrt(1 - self.b2_pow) / (1 - self.b1_pow)).to(torch.float32)
        p.sub_(self.exp_avg / denom * step_size)
        return loss

    def state_dict_flat(self):
        return dict(exp_avg=self.exp_avg, exp_avg_sq=self.exp_avg_sq, max_exp_avg_sq=self.max_exp_avg_sq,
                    step=self.step_t, b1_pow=self.b1_pow, b2_pow=self.b2_pow)


class SharedRMSprop(torch.optim.Optimizer):
    """The reference's SharedRMSprop (shared_optim.py:8-87: alpha 0.99, eps 0.1, no momentum, not centered — what
    main.py:89-90 constructs) over a FlatParams bucket; same role as SharedAdam in the data-parallel design."""

    def __init__(self, params, lr=7e-4, alpha=0.99, eps=0.1, weight_decay=0, momentum=0, centered=False):
        if momentum != 0 or centered:
            raise NotImplementedError("SharedRMSprop: momentum / centered are never used by the reference's drivers")
        params = list(params)
        super(SharedRMSprop, self).__init__(params, dict(lr=lr, alpha=alpha, eps=eps, weight_decay=weight_decay))
        self.bucket = FlatParams(params)
        self.square_avg = torch.zeros_like(self.bucket.flat)
        self.fused = True

    def share_memory(self):
        return self

    def zero_grad(self, set_to_none=False):
        self.bucket.zero_grad()

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        g = self.param_groups[0]
        grad, p = self.bucket.grad, self.bucket.flat
        if self.fused and p.is_cuda:
            from . import fused
            fused.rmsprop_step(p, grad, self.square_avg, g['lr'], g['alpha'], g['eps'], g['weight_decay'])
            return loss
        if g['weight_decay'] != 0:
            grad = grad.add(p, alpha=g['weight_decay'])
        self.square_avg.mul_(g['alpha']).addcmul_(grad, grad, value=1 - g['alpha'])
        p.addcdiv_(grad, self.square_avg.sqrt().add_(g['eps']), value=-g['lr'])
        return loss


def make_optimizer(params, args):
    """The optimizer main.py:86-95 + train.py:45-49 of the reference arrive at for these flags. With --shared-optimizer:
    SharedAdam(lr, amsgrad) / SharedRMSprop(lr). Without it every worker builds torch.optim.Adam(lr) / RMSprop(lr) for
    itself; in the synchronous data-parallel design all replicas see the same averaged gradient, so those per-worker
    states are one and the same — kept as one flat bucket with torch.optim's numerics (Adam: eps 1e-8 inside the
    bias-corrected root, no AMSGrad; RMSprop: alpha 0.99, eps 1e-8)."""
    name = getattr(args, "optimizer", "Adam")
    shared = bool(getattr(args, "shared_optimizer", True))
    if name == 'Adam':
        if shared:
            return SharedAdam(params, lr=args.lr, amsgrad=args.amsgrad)
        return SharedAdam(params, lr=args.lr, eps=1e-8, amsgrad=False, torch_eps=True)
    if name == 'RMSprop':
        return SharedRMSprop(params, lr=args.lr) if shared else SharedRMSprop(params, lr=args.lr, eps=1e-8)
    raise ValueError("--optimizer %s: the reference knows Adam and RMSprop (main.py:28)" % name)
