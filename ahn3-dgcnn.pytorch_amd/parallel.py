"""Data-parallel exchange step of the EdgeConv path: one process per GPU, batch elements sharded
across ranks (they are independent units: kNN, gather and max-over-k never cross a batch element,
SURVEY.md 8e), and ONE all-reduce per step of a single flat fp32 gradient buffer -- over
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests) or,
with a `HipComm`, over the library's own RCCL binding behind the C ABI (dgcnn_comm_*).

The reference has no distributed code at all (train.py:110-111 only prints the device count).

BatchNorm: per-replica batch statistics (standard data-parallel semantics); running buffers can
be averaged with `sync_buffers`.  The whole model's gradient message is 980 864 floats = 3.9 MB,
latency-bound on a fully connected xGMI node, so it is sent as one bucket after backward rather
than sliced and overlapped.
"""
import torch
import torch.distributed as dist


def _host_staged(t, group):
    """gloo (the CPU rehearsal backend) moves host memory: a device buffer goes through a host copy.  RCCL ("nccl")
    reduces device memory in place."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def all_reduce_sum(t, group=None):
    if _host_staged(t, group):
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def broadcast(t, src=0, group=None):
    """same tensor on every rank (initial parameters of the replicas)"""
    if _host_staged(t, group):
        h = t.cpu()
        dist.broadcast(h, src, group=group)
        t.copy_(h)
    else:
        dist.broadcast(t, src, group=group)
    return t


class HipComm:
    """The exchange step behind the C ABI (dgcnn_comm_*, comm.hip): an RCCL communicator of this process's current HIP
    device, reached without torch.distributed in the data path.  Bootstrap: rank 0 makes the 128-byte id, every rank
    needs the same bytes -- pass `id_bytes`, or let an initialised torch.distributed group (any backend: it only carries
    128 bytes once) broadcast them; then every rank constructs its HipComm (a collective call)."""

    def __init__(self, rank, world, id_bytes=None, group=None):
        import ctypes
        from . import _lib
        self._lib, self._libmod = _lib.load(), _lib
        self.rank, self.world = int(rank), int(world)
        if id_bytes is None:
            buf = ctypes.create_string_buffer(128)
            if self.rank == 0:
                _lib.check(self._lib.dgcnn_comm_unique_id(buf), "dgcnn_comm_unique_id")
            box = [bytes(buf.raw)]
            if self.world > 1:
                if not (dist.is_available() and dist.is_initialized()):
                    raise RuntimeError("HipComm: pass id_bytes (rank 0's dgcnn_comm_unique_id) or initialise torch.distributed "
                                       "so that it can be broadcast")
                dist.broadcast_object_list(box, src=0, group=group)
            id_bytes = box[0]
        if len(id_bytes) != 128:
            raise ValueError("HipComm: the id must be 128 bytes")
        self._handle = ctypes.c_void_p()
        _lib.check(self._lib.dgcnn_comm_init(ctypes.byref(self._handle), ctypes.c_char_p(id_bytes), self.world, self.rank),
                   "dgcnn_comm_init")

    def _check(self, t):
        self._libmod.require_device(t, "buffer")
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise TypeError("HipComm moves contiguous float32 device buffers")

    def all_reduce(self, t, average=False):
        """in place, on the current stream: sum (or mean) over the ranks"""
        self._check(t)
        self._libmod.check(self._lib.dgcnn_comm_allreduce_f32(self._handle, self._libmod.ptr(t), t.numel(), 1 if average else 0,
                                                            self._libmod.stream_ptr(t.device)), "dgcnn_comm_allreduce_f32")
        return t

    def broadcast(self, t, root=0):
        self._check(t)
        self._libmod.check(self._lib.dgcnn_comm_broadcast_f32(self._handle, self._libmod.ptr(t), t.numel(), int(root),
                                                            self._libmod.stream_ptr(t.device)), "dgcnn_comm_broadcast_f32")
        return t

    def close(self):
        if self._handle:
            self._libmod.check(self._lib.dgcnn_comm_destroy(self._handle), "dgcnn_comm_destroy")
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FlatGradAllReduce:
    """Keeps the gradients of `params` in one flat buffer and averages it across the process group with a
    single all-reduce.  `zero()` drops the .grad tensors, so that backward hands every parameter its
    gradient without an accumulation kernel, and `sync()` packs them into the flat buffer with ONE
    multi-tensor copy and re-points every .grad at its slice; with `zero(set_to_none=False)` the .grad
    tensors stay (zeroed) views of the buffer and backward accumulates in place (gradient accumulation
    over several backward passes)."""

    def __init__(self, params, group=None, comm=None, overlap=False):
        """comm: a HipComm -- the all-reduce then runs through the C ABI (dgcnn_comm_allreduce_f32) instead of
        torch.distributed.
        overlap: the backward of ops.dgcnn_full starts the all-reduce of the buffer's TAIL -- conv6 .. conv9, the last four
        tensors, 951 552 of the model's 980 864 gradient floats -- as soon as the head's backward is enqueued
        (`reduce_tail_async`), so that it runs on the collective's own stream under the three EdgeConv blocks' backward;
        `sync` then reduces the short front part and waits for the tail.  torch.distributed exchange only."""
        self.comm = comm
        self.overlap = bool(overlap)
        self._tail = None        # (offset, work, host copy or None) of a tail reduction in flight
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
        self.views, off = [], 0
        for p in self.params:
            n = p.numel()
            self.views.append(self.flat[off:off + n].view_as(p))
            p.grad = self.views[-1]
            # ops' backward passes write a parameter's gradient straight into its slice when they find this (ops._EdgeConv):
            # `pack` then has nothing to copy for it
            p._dgcnn_grad_sink = self.views[-1]
            p._dgcnn_grad_sync = self
            off += n

    @property
    def world_size(self):
        if self.comm is not None:
            return self.comm.world
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def zero(self, set_to_none=True):
        for p in self.params:
            p._dgcnn_sink_claimed = False   # a new cycle: each sink may be handed out once again (ops._EdgeConv.backward)
        if set_to_none:
            for p in self.params:
                p.grad = None
        else:
            self.flat.zero_()
            for p, v in zip(self.params, self.views):
                p.grad = v

    def pack(self):
        """gather the .grad tensors backward produced into the flat buffer (one multi-tensor copy); afterwards every
        .grad is a view of its slice again"""
        src, dst = [], []
        for p, v in zip(self.params, self.views):
            p._dgcnn_sink_claimed = False
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                src.append(p.grad)
                dst.append(v)
            p.grad = v
        if src:
            torch._foreach_copy_(dst, src)

    def _use_avg(self):
        """RCCL averages inside the collective (ncclAvg); decided ONCE, by a one-element probe every rank runs at the same
        point of its first exchange (a build without it: sum and scale instead)"""
        if self._avg is None:
            self._avg = False
            if dist.get_backend(self.group) == "nccl":
                try:
                    probe = torch.ones(1, dtype=self.flat.dtype, device=self.flat.device)
                    dist.all_reduce(probe, op=dist.ReduceOp.AVG, group=self.group)
                    self._avg = True
                except RuntimeError:
                    self._avg = False
        return self._avg

    def wants_overlap(self):
        return self.overlap and self.comm is None and self._tail is None and self.world_size > 1

    def reduce_tail_async(self, first, final):
        """start the all-reduce (mean) of the flat buffer from parameter `first` to its end.  `final`: the parameters whose
        gradients are final and already sit in their slices (enqueued on the current stream); nothing is started (False)
        unless every parameter from `first` on is one of them.  Returns at once: the collective runs on torch.distributed's
        own stream."""
        k = next((i for i, p in enumerate(self.params) if p is first), None)
        if k is None or not all(any(p is q for q in final) for p in self.params[k:]):
            return False
        off = sum(p.numel() for p in self.params[:k])
        tail = self.flat[off:]
        if _host_staged(tail, self.group):          # gloo rehearsal: the host copy travels
            h = tail.cpu()
            work = dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._tail = (off, work, h)
        else:
            op = dist.ReduceOp.AVG if self._use_avg() else dist.ReduceOp.SUM
            work = dist.all_reduce(tail, op=op, group=self.group, async_op=True)
            self._tail = (off, work, None)
        return True

    def _finish_tail(self, ws):
        off, work, h = self._tail
        self._tail = None
        work.wait()
        tail = self.flat[off:]
        if h is not None:
            tail.copy_(h)
            tail.div_(ws)
        elif not self._avg:
            tail.div_(ws)

    def sync(self):
        """sum over ranks, then divide by the world size (gradient of the mean loss over the global batch)."""
        self.pack()
        ws = self.world_size
        if self._tail is not None:      # the tail is in flight since the backward: reduce the front, then join
            front = self.flat[:self._tail[0]]
            if front.numel():
                if not _host_staged(front, self.group) and self._use_avg():
                    dist.all_reduce(front, op=dist.ReduceOp.AVG, group=self.group)
                else:
                    all_reduce_sum(front, self.group)
                    front.div_(ws)
            self._finish_tail(ws)
            return self.flat
        if self.comm is not None:
            if ws > 1:
                self.comm.all_reduce(self.flat, average=True)
            return self.flat
        if ws > 1:
            if not _host_staged(self.flat, self.group) and self._use_avg():
                dist.all_reduce(self.flat, op=dist.ReduceOp.AVG, group=self.group)
                return self.flat
            all_reduce_sum(self.flat, self.group)
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
    all_reduce_sum(flat, group)
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
