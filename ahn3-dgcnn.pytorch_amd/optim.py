"""Adam over ONE flat parameter buffer: the reference builds
`optim.Adam(model.parameters(), lr=args.lr, weight_decay=1e-4)` (train.py:118) and calls `opt.zero_grad()` /
`opt.step()` once per batch (train.py:178,199).  Here the 25 parameter tensors of the model become views into one
flat fp32 buffer, their gradients views into a second one (shared with parallel.FlatGradAllReduce when the step is
data-parallel), and a step is one HIP kernel over 980 864 elements instead of a multi-tensor launch sequence.

Same update rule as torch.optim.Adam (no amsgrad, L2 weight decay added to the gradient).  No CPU fallback.
"""
import torch

from . import _lib


class FlatAdam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_buffer=None):
        seen, uniq = set(), []
        for p in params:  # bnN and convN.1 alias the same tensors in the reference's layout
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                uniq.append(p)
        if not uniq:
            raise ValueError("no parameters to optimise")
        self.params = uniq
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        dev = uniq[0].device
        _lib.require_device(uniq[0], "parameters")
        total = sum(p.numel() for p in uniq)
        self.flat = torch.empty(total, dtype=torch.float32, device=dev)
        own_grads = grad_buffer is None
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev) if own_grads else grad_buffer
        if self.grad.numel() != total:
            raise ValueError("grad_buffer has %d elements, the parameters %d" % (self.grad.numel(), total))
        off = 0
        self._grad_views = []
        with torch.no_grad():
            for p in uniq:
                n = p.numel()
                self.flat[off:off + n].copy_(p.reshape(-1))
                p.data = self.flat[off:off + n].view_as(p)       # the module now computes on views of the flat buffer
                self._grad_views.append(self.grad[off:off + n].view_as(p))
                off += n
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.steps = 0

    def zero_grad(self, set_to_none=True):
        """set_to_none=True (torch's default): drop the .grad tensors, so that backward hands each parameter its
        gradient without an accumulation kernel (25 tiny launches per step otherwise); `step` packs them into the
        flat buffer with one multi-tensor copy.  set_to_none=False: the .grad tensors become (zeroed) views of the flat
        buffer and backward accumulates in place -- what a shared parallel.FlatGradAllReduce buffer expects."""
        if set_to_none:
            for p in self.params:
                p.grad = None
        else:
            self.grad.zero_()
            for p, v in zip(self.params, self._grad_views):
                p.grad = v

    def _pack_grads(self):
        src, dst = [], []
        for p, v in zip(self.params, self._grad_views):
            if p.grad is None:
                v.zero_()  # (torch.optim.Adam would skip such a parameter; every DGCNN parameter gets a gradient)
            elif p.grad.data_ptr() != v.data_ptr():
                src.append(p.grad)
                dst.append(v)
        if src:
            torch._foreach_copy_(dst, src)

    def step(self):
        self._pack_grads()
        self.steps += 1
        lib = _lib.load()
        _lib.check(lib.dgcnn_adam_step(_lib.ptr(self.flat), _lib.ptr(self.grad), _lib.ptr(self.exp_avg),
                                       _lib.ptr(self.exp_avg_sq), self.flat.numel(), self.lr, self.betas[0], self.betas[1],
                                       self.eps, self.weight_decay, self.steps, _lib.stream_ptr(self.flat.device)),
                   "dgcnn_adam_step")

    def state_dict(self):
        return {"step": self.steps, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq,
                "lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay}

    def load_state_dict(self, sd):
        self.steps = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.lr, self.betas, self.eps, self.weight_decay = sd["lr"], tuple(sd["betas"]), sd["eps"], sd["weight_decay"]
