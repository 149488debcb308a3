"""Data-parallel exchange step of the EdgeConv path: one process per GPU, batch elements sharded
across ranks (they are independent units: kNN, gather and max-over-k never cross a batch element,
SURVEY.md 8e), and ONE all-reduce per step of a single flat fp32 gradient buffer over
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

The reference has no distributed code at all (train.py:110-111 only prints the device count).

BatchNorm: per-replica batch statistics (standard data-parallel semantics); running buffers can
be averaged with `sync_buffers`.  The whole model's gradient message is 980 864 floats = 3.9 MB,
latency-bound on a fully connected xGMI node, so it is sent as one bucket after backward rather
than sliced and overlapped.
"""
import torch
import torch.distributed as dist


class FlatGradAllReduce:
    """Packs the gradients of `params` into one flat buffer (the .grad of every parameter becomes a
    view into it, so backward accumulates in place and no per-step packing copy is needed) and
    averages it across the process group with a single all-reduce."""

    def __init__(self, params, group=None):
        self.params = [p for p in params if p.requires_grad]
        seen, uniq = set(), []
        for p in self.params:  # bnN and convN.1 alias the same tensors in the reference's layout
            if id(p) not in seen:
                seen.add(id(p))
                uniq.append(p)
        self.params = uniq
        self.group = group
        self._avg = None
        if not self.params:
            raise ValueError("no parameters to synchronise")
        dev, dt = self.params[0].device, self.params[0].dtype
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, dtype=dt, device=dev)
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            off += n

    @property
    def world_size(self):
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def zero(self):
        self.flat.zero_()

    def sync(self):
        """sum over ranks, then divide by the world size (gradient of the mean loss over the global batch)."""
        ws = self.world_size
        if ws > 1:
            if self._avg is None:
                self._avg = dist.get_backend(self.group) == "nccl"  # RCCL averages inside the collective
            if self._avg:
                try:
                    dist.all_reduce(self.flat, op=dist.ReduceOp.AVG, group=self.group)
                    return self.flat
                except RuntimeError:  # a build without ncclAvg: sum and scale instead
                    self._avg = False
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            self.flat.div_(ws)
        return self.flat


def sync_buffers(module, group=None):
    """Average the floating-point buffers (BatchNorm running statistics) across ranks."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    bufs = [b for b in module.buffers() if b.is_floating_point()]
    if not bufs:
        return
    flat = torch.cat([b.reshape(-1) for b in bufs])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.div_(dist.get_world_size(group))
    off = 0
    for b in bufs:
        n = b.numel()
        b.copy_(flat[off:off + n].view_as(b))
        off += n


def shard_batch(x, rank, world_size):
    """Contiguous shard of the batch dimension for this rank (points of one block never split)."""
    B = x.shape[0]
    if B % world_size:
        raise ValueError("global batch %d is not divisible by %d ranks" % (B, world_size))
    per = B // world_size
    return x[rank * per:(rank + 1) * per]
