"""Adam over ONE flat parameter buffer: the reference builds
`optim.Adam(model.parameters(), lr=args.lr, weight_decay=1e-4)` (train.py:118), wraps it in CosineAnnealingLR / StepLR
(train.py:121-124), restores `checkpoint['optimizer_state_dict']` into it (train.py:131), calls `opt.zero_grad()` /
`opt.step()` once per batch (train.py:178,199) and saves `opt.state_dict()` (train.py:306-309).  Here the 25 parameter
tensors of the model become views into one flat fp32 buffer, their gradients views into a second one (shared with
parallel.FlatGradAllReduce when the step is data-parallel), and a step is one HIP kernel over 980 864 elements instead
of a multi-tensor launch sequence.

`FlatAdam` IS a `torch.optim.Optimizer` with one parameter group, so the reference's schedulers drive it through
`param_groups[0]['lr']` (read at every step), and its `state_dict()` / `load_state_dict()` speak torch.optim.Adam's
layout (per-parameter `step`, `exp_avg`, `exp_avg_sq`; the moments are views of the flat buffers): checkpoints
interchange with the reference in both directions.  Same update rule as torch.optim.Adam (no amsgrad, L2 weight decay
added to the gradient); a parameter without a gradient is skipped, as torch does.  No CPU fallback.
"""
import torch

from . import _lib

_ADAM_GROUP_DEFAULTS = dict(amsgrad=False, maximize=False, foreach=None, capturable=False, differentiable=False,
                            fused=None, decoupled_weight_decay=False)


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_buffer=None):
        params = list(params)
        if params and isinstance(params[0], dict):
            if len(params) != 1:
                raise ValueError("FlatAdam keeps ONE parameter group (the reference's train.py:118 has one)")
            group_opts = {k: v for k, v in params[0].items() if k != "params"}
            params = list(params[0]["params"])
            lr = group_opts.get("lr", lr)
            betas = group_opts.get("betas", betas)
            eps = group_opts.get("eps", eps)
            weight_decay = group_opts.get("weight_decay", weight_decay)
        seen, uniq = set(), []
        for p in params:  # bnN and convN.1 alias the same tensors in the reference's layout
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                uniq.append(p)
        if not uniq:
            raise ValueError("no parameters to optimise")
        if not 0.0 <= float(lr):
            raise ValueError("Invalid learning rate: %r" % (lr,))
        defaults = dict(lr=float(lr), betas=(float(betas[0]), float(betas[1])), eps=float(eps),
                        weight_decay=float(weight_decay), **_ADAM_GROUP_DEFAULTS)
        _lib.require_device(uniq[0], "parameters")
        dev = uniq[0].device
        for p in uniq:
            if p.dtype != torch.float32 or p.device != dev:
                raise TypeError("FlatAdam needs float32 parameters on one HIP device")
        super().__init__(uniq, defaults)
        self.params = uniq
        total = sum(p.numel() for p in uniq)
        self.flat = torch.empty(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev) if grad_buffer is None else grad_buffer
        if self.grad.numel() != total or self.grad.dtype != torch.float32 or self.grad.device != dev:
            raise ValueError("grad_buffer must be %d float32 elements on %s" % (total, dev))
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self._offsets, self._grad_views, off = [], [], 0
        with torch.no_grad():
            for p in uniq:
                n = p.numel()
                self.flat[off:off + n].copy_(p.reshape(-1))
                p.data = self.flat[off:off + n].view_as(p)       # the module now computes on views of the flat buffer
                self._grad_views.append(self.grad[off:off + n].view_as(p))
                # the backward passes of ops write a parameter's gradient straight into this slice when they find it
                # (ops._EdgeConv, ops._DGCNNFull): `_pack_grads` then has nothing to copy.  A sink a
                # parallel.FlatGradAllReduce has named (the shared exchange buffer) stays.
                if getattr(p, "_dgcnn_grad_sink", None) is None:
                    p._dgcnn_grad_sink = self._grad_views[-1]
                self._offsets.append(off)
                off += n
        self._steps = [0] * len(uniq)   # torch.optim.Adam counts the steps per parameter (a skipped one lags)

    # -- gradients ------------------------------------------------------------------------------------------
    def zero_grad(self, set_to_none=True):
        """set_to_none=True (torch's default): drop the .grad tensors, so that backward hands each parameter its
        gradient without an accumulation kernel (25 tiny launches per step otherwise); `step` packs them into the
        flat buffer with one multi-tensor copy.  set_to_none=False: the .grad tensors become (zeroed) views of the flat
        buffer and backward accumulates in place -- what a shared parallel.FlatGradAllReduce buffer expects."""
        for p in self.params:
            p._dgcnn_sink_claimed = False   # a new cycle: each sink may be handed out once again
        if set_to_none:
            for p in self.params:
                p.grad = None
        else:
            self.grad.zero_()
            for p, v in zip(self.params, self._grad_views):
                p.grad = v

    def _pack_grads(self):
        """-> per parameter, whether it has a gradient this step; gradients that live elsewhere are copied into their
        slices of the flat buffer with one multi-tensor copy"""
        src, dst, has = [], [], []
        for p, v in zip(self.params, self._grad_views):
            p._dgcnn_sink_claimed = False
            has.append(p.grad is not None)
            if p.grad is not None and p.grad.data_ptr() != v.data_ptr():
                src.append(p.grad)
                dst.append(v)
        if src:
            torch._foreach_copy_(dst, src)
        return has

    def _check_storage(self):
        base = self.flat.data_ptr()
        for p, off in zip(self.params, self._offsets):
            if p.data_ptr() != base + 4 * off:
                raise RuntimeError("a parameter no longer lives in FlatAdam's flat buffer (the module was moved or cast "
                                   "after the optimizer was built, e.g. model.to()/float()): build the optimizer again")

    # -- the step ---------------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._check_storage()
        has = self._pack_grads()
        g = self.param_groups[0]
        if g.get("amsgrad") or g.get("maximize"):
            raise NotImplementedError("FlatAdam implements torch.optim.Adam without amsgrad / maximize (train.py:118)")
        lr, (b1, b2), eps, wd = float(g["lr"]), g["betas"], float(g["eps"]), float(g["weight_decay"])
        lib = _lib.load()
        st = _lib.stream_ptr(self.flat.device)
        # one launch per run of adjacent parameters that have a gradient and the same step count: ONE for the model
        i, n_par = 0, len(self.params)
        while i < n_par:
            if not has[i]:
                i += 1
                continue
            j = i
            while j + 1 < n_par and has[j + 1] and self._steps[j + 1] == self._steps[i]:
                j += 1
            lo = self._offsets[i]
            hi = self._offsets[j] + self.params[j].numel()
            step = self._steps[i] + 1
            _lib.check(lib.dgcnn_adam_step(self.flat.data_ptr() + 4 * lo, self.grad.data_ptr() + 4 * lo,
                                           self.exp_avg.data_ptr() + 4 * lo, self.exp_avg_sq.data_ptr() + 4 * lo,
                                           hi - lo, lr, float(b1), float(b2), eps, wd, step, st), "dgcnn_adam_step")
            for q in range(i, j + 1):
                self._steps[q] = step
            i = j + 1
        return loss

    @property
    def steps(self):
        return max(self._steps)

    # -- torch.optim.Adam's state layout ---------------------------------------------------------------------------
    def _refresh_state(self):
        self.state.clear()
        for p, off, s in zip(self.params, self._offsets, self._steps):
            if s == 0:
                continue     # torch creates a parameter's state at its first step
            n = p.numel()
            self.state[p] = {"step": torch.tensor(float(s), dtype=torch.float32),
                             "exp_avg": self.exp_avg[off:off + n].view_as(p),
                             "exp_avg_sq": self.exp_avg_sq[off:off + n].view_as(p)}

    def state_dict(self):
        self._refresh_state()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        """accepts what torch.optim.Adam.state_dict() (or this class) wrote for the same parameter list"""
        super().load_state_dict(state_dict)      # validates sizes, maps indices to parameters, casts to the device
        if len(self.param_groups) != 1 or len(self.param_groups[0]["params"]) != len(self.params):
            raise ValueError("the loaded state has a different parameter list")
        with torch.no_grad():
            for i, (p, off) in enumerate(zip(self.params, self._offsets)):
                n = p.numel()
                st = self.state.get(p)
                if not st:
                    self._steps[i] = 0
                    self.exp_avg[off:off + n].zero_()
                    self.exp_avg_sq[off:off + n].zero_()
                    continue
                if st.get("max_exp_avg_sq") is not None:
                    raise NotImplementedError("amsgrad state cannot be loaded into FlatAdam")
                self._steps[i] = int(round(float(st["step"])))
                self.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1))
                self.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
        self._refresh_state()
