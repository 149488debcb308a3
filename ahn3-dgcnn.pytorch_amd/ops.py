"""Host-side operators of the EdgeConv path: thin wrappers that allocate outputs/workspaces
with torch and enqueue the HIP kernels of libdgcnn_hip.so on the current stream.

Internal layout is point-major [B, N, C] fp32 ("pm"); neighbour indices are int32, local to
the batch element.
"""
import ctypes

import torch

from . import _lib


class batch_counters:
    """Context in which the `num_batches_tracked += 1` of every BatchNorm stage (one tiny kernel each: eight per
    DGCNN forward) is collected and applied as ONE multi-tensor add on exit.  Outside it the counters are bumped at
    once, as nn.BatchNorm does."""
    _open = None

    def __enter__(self):
        self._outer, batch_counters._open = batch_counters._open, []
        return self

    def __exit__(self, *exc):
        pending, batch_counters._open = batch_counters._open, self._outer
        if pending and exc[0] is None:
            torch._foreach_add_(pending, 1)
        return False


def _count_batch(bn):
    if bn is not None and bn.num_batches_tracked is not None:
        if batch_counters._open is not None:
            batch_counters._open.append(bn.num_batches_tracked)
        else:
            bn.num_batches_tracked += 1


def _f32c(t, name):
    _lib.require_device(t, name)
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32, got %s" % (name, t.dtype))
    return t if t.is_contiguous() else t.contiguous()


def to_point_major(x):
    """[B, C, N] fp32 view with arbitrary strides -> contiguous point-major [B, N, C].

    Replaces x.transpose(2, 1).contiguous() (reference model.py:73) and absorbs the strided
    inputs the callers produce (train.py:176, test.py:119, test.py:304-306)."""
    _lib.require_device(x, "x")
    if x.dtype != torch.float32:
        raise TypeError("x must be float32, got %s" % x.dtype)
    B, C, N = x.shape
    out = torch.empty((B, N, C), dtype=torch.float32, device=x.device)
    sb, sc, sn = x.stride()
    lib = _lib.load()
    _lib.check(lib.dgcnn_to_point_major(_lib.ptr(x), B, C, N, sb, sc, sn, _lib.ptr(out), _lib.stream_ptr(x.device)),
               "dgcnn_to_point_major")
    return out


def from_point_major(x_pm):
    """point-major [B, N, C] -> contiguous [B, C, N] (the layout the reference returns)."""
    x_pm = _f32c(x_pm, "x_pm")
    B, N, C = x_pm.shape
    out = torch.empty((B, C, N), dtype=torch.float32, device=x_pm.device)
    lib = _lib.load()
    _lib.check(lib.dgcnn_from_point_major(_lib.ptr(x_pm), B, C, N, _lib.ptr(out), C * N, N, 1,
                                          _lib.stream_ptr(x_pm.device)), "dgcnn_from_point_major")
    return out


def knn_pm(x_pm, k):
    """kNN graph of point-major features [B, N, C] -> int32 [B, N, k] (nearest first, self included)."""
    x_pm = _f32c(x_pm.detach(), "x_pm")
    B, N, C = x_pm.shape
    lib = _lib.load()
    if k > N:
        # same failure the reference gets from topk (model.py:39)
        raise RuntimeError("selected index k out of range")
    idx = torch.empty((B, N, k), dtype=torch.int32, device=x_pm.device)
    ws_bytes = lib.dgcnn_knn_workspace_bytes(B, C, N, k)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=x_pm.device)
    _lib.check(lib.dgcnn_knn(_lib.ptr(x_pm), B, N, C, k, _lib.ptr(idx), _lib.ptr(ws), ws_bytes,
                             _lib.stream_ptr(x_pm.device)), "dgcnn_knn")
    return idx


# ------------------------------------------------------------------------------------------------
class _GraphFeature(torch.autograd.Function):
    """Materialised (x_j - x_i, x_i) edge tuples, memory [B, N, k, 2C] (reference model.py:64-81)."""

    @staticmethod
    def forward(ctx, x_pm, idx):
        B, N, C = x_pm.shape
        k = idx.shape[2]
        out = torch.empty((B, N, k, 2 * C), dtype=torch.float32, device=x_pm.device)
        lib = _lib.load()
        _lib.check(lib.dgcnn_graph_feature_fwd(_lib.ptr(x_pm), _lib.ptr(idx), B, N, C, k, _lib.ptr(out),
                                               _lib.stream_ptr(x_pm.device)), "dgcnn_graph_feature_fwd")
        ctx.save_for_backward(idx)
        ctx.shape = (B, N, C, k)
        return out

    @staticmethod
    def backward(ctx, gout):
        (idx,) = ctx.saved_tensors
        B, N, C, k = ctx.shape
        gout = gout.contiguous()
        gx = torch.empty((B, N, C), dtype=torch.float32, device=gout.device)
        lib = _lib.load()
        ws_bytes = lib.dgcnn_graph_feature_bwd_workspace_bytes(B, N, C, k)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=gout.device)
        _lib.check(lib.dgcnn_graph_feature_bwd(_lib.ptr(gout), _lib.ptr(idx), B, N, C, k, _lib.ptr(gx), _lib.ptr(ws),
                                               ws_bytes, _lib.stream_ptr(gout.device)), "dgcnn_graph_feature_bwd")
        return gx, None


def graph_feature_pm(x_pm, idx):
    """x_pm [B, N, C], idx int32 [B, N, k] -> [B, N, k, 2C] (differentiable w.r.t. x_pm)."""
    x_pm = _f32c(x_pm, "x_pm")
    _lib.require_device(idx, "idx")
    if idx.dtype != torch.int32:
        idx = idx.to(torch.int32)
    return _GraphFeature.apply(x_pm, idx.contiguous())


# ------------------------------------------------------------------------------------------------
def _bn_struct(weight, bias, rmean, rvar, nbt=None):
    s = _lib.BnParams()
    s.weight, s.bias = weight.data_ptr(), bias.data_ptr()
    s.running_mean = rmean.data_ptr() if rmean is not None else None
    s.running_var = rvar.data_ptr() if rvar is not None else None
    s.num_batches_tracked = nbt.data_ptr() if nbt is not None else None
    return s


def _make_desc(B, N, C, k, training, eps, momentum, slope, w1, g1, b1, rm1, rv1, w2, g2, b2, rm2, rv2, nbt1=None, nbt2=None):
    d = _lib.EdgeConvDesc()
    d.B, d.N, d.C, d.k = B, N, C, k
    d.n_stage = 2 if w2 is not None else 1
    d.training = 1 if training else 0
    d.eps, d.momentum, d.slope = eps, momentum, slope
    d.w1 = w1.data_ptr()
    d.bn1 = _bn_struct(g1, b1, rm1, rv1, nbt1)
    if w2 is not None:
        d.w2 = w2.data_ptr()
        d.bn2 = _bn_struct(g2, b2, rm2, rv2, nbt2)
    return d


class _EdgeConv(torch.autograd.Function):
    """One EdgeConv block (reference model.py:147-150 / 154-157 / 161-163) through the C ABI."""

    @staticmethod
    def forward(ctx, x_pm, idx, w1, g1, b1, w2, g2, b2, rm1, rv1, rm2, rv2, training, eps, momentum, slope, nbt1, nbt2):
        B, N, C = x_pm.shape
        k = idx.shape[2]
        lib = _lib.load()
        dev = x_pm.device
        w1c = w1.detach().reshape(64, 2 * C).contiguous()
        w2c = w2.detach().reshape(64, 64).contiguous() if w2 is not None else None
        tens = [t.detach().contiguous() if t is not None else None for t in (g1, b1, g2, b2)]
        g1c, b1c, g2c, b2c = tens
        d = _make_desc(B, N, C, k, training, eps, momentum, slope, w1c, g1c, b1c, rm1, rv1, w2c, g2c, b2c, rm2, rv2,
                       nbt1, nbt2)  # (the counters are bumped by the launch that finishes the batch statistics)
        y = torch.empty((B, N, 64), dtype=torch.float32, device=dev)
        saved = torch.empty((lib.dgcnn_edgeconv_saved_bytes(d),), dtype=torch.uint8, device=dev)
        ws_bytes = lib.dgcnn_edgeconv_fwd_workspace_bytes(d)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        _lib.check(lib.dgcnn_edgeconv_fwd(d, _lib.ptr(x_pm), _lib.ptr(idx), _lib.ptr(y), _lib.ptr(saved), _lib.ptr(ws),
                                          ws_bytes, _lib.stream_ptr(dev)), "dgcnn_edgeconv_fwd")
        ctx.save_for_backward(x_pm, idx, y, saved, w1c, g1c, b1c, w2c, g2c, b2c, rm1, rv1, rm2, rv2)
        ctx.cfg = (training, eps, momentum, slope, tuple(w1.shape), tuple(w2.shape) if w2 is not None else None)
        # where the gradients of the parameters are to be written, if their owner has said so (parallel.FlatGradAllReduce:
        # slices of its flat exchange buffer -- the backward then fills the buffer itself and nothing is packed)
        ctx.sinks = tuple((t, getattr(t, "_dgcnn_grad_sink", None)) if t is not None else (None, None) for t in (w1, g1, b1, w2, g2, b2))
        return y

    @staticmethod
    def backward(ctx, gy):
        x_pm, idx, y, saved, w1c, g1c, b1c, w2c, g2c, b2c, rm1, rv1, rm2, rv2 = ctx.saved_tensors
        training, eps, momentum, slope, w1_shape, w2_shape = ctx.cfg
        B, N, C = x_pm.shape
        k = idx.shape[2]
        lib = _lib.load()
        dev = x_pm.device
        gy = gy.contiguous()
        # training=… only selects the BatchNorm backward form; running statistics are not touched here
        d = _make_desc(B, N, C, k, training, eps, momentum, slope, w1c, g1c, b1c, rm1, rv1, w2c, g2c, b2c, rm2, rv2)
        g = _lib.EdgeConvGrads()
        need_x = ctx.needs_input_grad[0]
        gx = torch.empty_like(x_pm) if need_x else None

        def out_like(ref, owner_sink):
            # a NEW tensor object on the sink's memory (autograd keeps a gradient it alone refers to instead of cloning it).
            # Only while the parameter has no gradient yet: an accumulating backward must not overwrite what .grad holds
            # and only ONCE per zero() / pack() cycle of the owner (parallel.FlatGradAllReduce clears the claim): a parameter
            # that feeds two nodes of one backward has no .grad yet when the second node runs -- both would write the slice
            # and autograd would add the aliased tensors
            owner, sink = owner_sink
            if (sink is not None and owner.grad is None and not getattr(owner, "_dgcnn_sink_claimed", False)
                    and owner.requires_grad and sink.numel() == ref.numel() and sink.dtype == ref.dtype
                    and sink.device == ref.device and sink.is_contiguous()):
                owner._dgcnn_sink_claimed = True
                return sink.detach().view(ref.shape)
            return torch.empty_like(ref)

        s_w1, s_g1, s_b1, s_w2, s_g2, s_b2 = ctx.sinks
        gw1 = out_like(w1c, s_w1)
        gg1, gb1 = out_like(g1c, s_g1), out_like(b1c, s_b1)
        g.gx = gx.data_ptr() if need_x else None
        g.gw1, g.gbn1_w, g.gbn1_b = gw1.data_ptr(), gg1.data_ptr(), gb1.data_ptr()
        gw2 = gg2 = gb2 = None
        if w2c is not None:
            gw2, gg2, gb2 = out_like(w2c, s_w2), out_like(g2c, s_g2), out_like(b2c, s_b2)
            g.gw2, g.gbn2_w, g.gbn2_b = gw2.data_ptr(), gg2.data_ptr(), gb2.data_ptr()
        ws_bytes = lib.dgcnn_edgeconv_bwd_workspace_bytes(d)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        _lib.check(lib.dgcnn_edgeconv_bwd(d, _lib.ptr(x_pm), _lib.ptr(idx), _lib.ptr(y), _lib.ptr(gy), _lib.ptr(saved),
                                          ctypes.byref(g), _lib.ptr(ws), ws_bytes, _lib.stream_ptr(dev)),
                   "dgcnn_edgeconv_bwd")
        return (gx, None, gw1.reshape(w1_shape), gg1, gb1,
                gw2.reshape(w2_shape) if gw2 is not None else None, gg2, gb2,
                None, None, None, None, None, None, None, None, None, None)


def edgeconv_pm(x_pm, idx, w1, bn1, w2=None, bn2=None, training=False, slope=0.2):
    """EdgeConv block on point-major features.

    x_pm [B, N, C] fp32, idx int32 [B, N, k]; w1 = Conv2d(2C, 64, 1).weight; bn1 / bn2 = the
    BatchNorm2d modules (weight, bias, running_mean, running_var, eps, momentum are read; the
    running statistics are updated in place when training).  Returns y [B, N, 64]."""
    x_pm = _f32c(x_pm, "x_pm")
    _lib.require_device(idx, "idx")
    if idx.dtype != torch.int32:
        idx = idx.to(torch.int32)
    idx = idx.contiguous()

    def unpack(bn):
        if bn is None:
            return None, None, None, None
        return bn.weight, bn.bias, bn.running_mean, bn.running_var

    g1, b1, rm1, rv1 = unpack(bn1)
    g2, b2, rm2, rv2 = unpack(bn2)
    use_batch_stats = training or rm1 is None
    momentum = bn1.momentum if bn1.momentum is not None else 0.1
    def counter(bn):  # nn.BatchNorm bumps it in training mode; an int64 scalar on the device goes with the launch
        t = bn.num_batches_tracked if (training and bn is not None) else None
        return t if (t is not None and t.dtype == torch.int64 and t.device == x_pm.device) else None

    nbt1, nbt2 = counter(bn1), counter(bn2)
    y = _EdgeConv.apply(x_pm, idx, w1, g1, b1, w2, g2, b2, rm1, rv1, rm2, rv2, bool(use_batch_stats),
                        float(bn1.eps), float(momentum), float(slope), nbt1, nbt2)
    if training:
        for bn, t in ((bn1, nbt1), (bn2, nbt2)):
            if t is None:
                _count_batch(bn)
    return y


# ------------------------------------------------------------------------------------------------
class _BNLReLURows(torch.autograd.Function):
    """BatchNorm1d + LeakyReLU on point-major rows [P, C] (head stages conv6/7/8 of the reference, model.py:132-139)."""

    @staticmethod
    def forward(ctx, x, weight, bias, rmean, rvar, training, eps, momentum, slope):
        P, C = x.shape
        lib = _lib.load()
        dev = x.device
        wc, bc = weight.detach().contiguous(), bias.detach().contiguous()
        bn = _bn_struct(wc, bc, rmean, rvar)
        y = torch.empty_like(x)
        stats = torch.empty((4, C), dtype=torch.float32, device=dev)
        ws_bytes = lib.dgcnn_bn_lrelu_workspace_bytes(P, C)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        _lib.check(lib.dgcnn_bn_lrelu_fwd(_lib.ptr(x), P, C, ctypes.byref(bn), 1 if training else 0, eps, momentum, slope,
                                          _lib.ptr(y), _lib.ptr(stats), _lib.ptr(ws), ws_bytes, _lib.stream_ptr(dev)),
                   "dgcnn_bn_lrelu_fwd")
        ctx.save_for_backward(x, y, stats)
        ctx.cfg = (training, slope)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y, stats = ctx.saved_tensors
        training, slope = ctx.cfg
        P, C = x.shape
        lib = _lib.load()
        dev = x.device
        gy = gy.contiguous()
        gx = torch.empty_like(x)
        gw = torch.empty((C,), dtype=torch.float32, device=dev)
        gb = torch.empty((C,), dtype=torch.float32, device=dev)
        ws_bytes = lib.dgcnn_bn_lrelu_workspace_bytes(P, C)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        _lib.check(lib.dgcnn_bn_lrelu_bwd(_lib.ptr(x), _lib.ptr(y), _lib.ptr(gy), _lib.ptr(stats), P, C,
                                          1 if training else 0, slope, _lib.ptr(gx), _lib.ptr(gw), _lib.ptr(gb),
                                          _lib.ptr(ws), ws_bytes, _lib.stream_ptr(dev)), "dgcnn_bn_lrelu_bwd")
        return gx, gw, gb, None, None, None, None, None, None


def bn_lrelu_rows(x, bn, training, slope=0.2):
    """y = LeakyReLU(BatchNorm1d(x)) for point-major rows x [P, C]; `bn` is the nn.BatchNorm1d module (running
    statistics updated in place when training)."""
    x = _f32c(x, "x")
    use_batch_stats = training or bn.running_mean is None
    momentum = bn.momentum if bn.momentum is not None else 0.1
    y = _BNLReLURows.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bool(use_batch_stats),
                           float(bn.eps), float(momentum), float(slope))
    if training:
        _count_batch(bn)
    return y


class _BNLReLUMax(torch.autograd.Function):
    """max over the N points of LeakyReLU(BatchNorm1d(x)) for point-major rows [B*N, C] (conv6 -> bn6 -> max of the
    reference, model.py:172-175) without materialising the activated tensor."""

    @staticmethod
    def forward(ctx, x, weight, bias, rmean, rvar, B, N, training, eps, momentum, slope):
        P, C = x.shape
        lib = _lib.load()
        dev = x.device
        wc, bc = weight.detach().contiguous(), bias.detach().contiguous()
        bn = _bn_struct(wc, bc, rmean, rvar)
        g = torch.empty((B, C), dtype=torch.float32, device=dev)
        arg = torch.empty((B, C), dtype=torch.int32, device=dev)
        stats = torch.empty((4, C), dtype=torch.float32, device=dev)
        ws_bytes = lib.dgcnn_bn_lrelu_max_workspace_bytes(B, N, C)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        _lib.check(lib.dgcnn_bn_lrelu_max_fwd(_lib.ptr(x), B, N, C, ctypes.byref(bn), 1 if training else 0, eps, momentum,
                                              slope, _lib.ptr(g), _lib.ptr(arg), _lib.ptr(stats), _lib.ptr(ws), ws_bytes,
                                              _lib.stream_ptr(dev)), "dgcnn_bn_lrelu_max_fwd")
        ctx.save_for_backward(x, arg, stats)
        ctx.cfg = (B, N, training, slope)
        ctx.mark_non_differentiable(arg)
        return g, arg

    @staticmethod
    def backward(ctx, gg, _garg):
        x, arg, stats = ctx.saved_tensors
        B, N, training, slope = ctx.cfg
        P, C = x.shape
        lib = _lib.load()
        dev = x.device
        gg = gg.contiguous()
        gx = torch.empty_like(x)
        gw = torch.empty((C,), dtype=torch.float32, device=dev)
        gb = torch.empty((C,), dtype=torch.float32, device=dev)
        ws_bytes = lib.dgcnn_bn_lrelu_max_workspace_bytes(B, N, C)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        _lib.check(lib.dgcnn_bn_lrelu_max_bwd(_lib.ptr(x), _lib.ptr(gg), _lib.ptr(arg), _lib.ptr(stats), B, N, C,
                                              1 if training else 0, slope, _lib.ptr(gx), _lib.ptr(gw), _lib.ptr(gb),
                                              _lib.ptr(ws), ws_bytes, _lib.stream_ptr(dev)), "dgcnn_bn_lrelu_max_bwd")
        return gx, gw, gb, None, None, None, None, None, None, None, None


def bn_lrelu_max_rows(x, bn, B, N, training, slope=0.2):
    """g[b] = max over the N rows of batch element b of LeakyReLU(BatchNorm1d(x)); x point-major rows [B*N, C] -> [B, C]
    (`bn` is the nn.BatchNorm1d module; running statistics updated in place when training)."""
    x = _f32c(x, "x")
    if x.shape[0] != B * N:
        raise ValueError("x must have B*N rows")
    use_batch_stats = training or bn.running_mean is None
    momentum = bn.momentum if bn.momentum is not None else 0.1
    g, _ = _BNLReLUMax.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, int(B), int(N),
                             bool(use_batch_stats), float(bn.eps), float(momentum), float(slope))
    if training:
        _count_batch(bn)
    return g


def rows_absmax(x):
    """one int32 word on the device: the bits of max |x| over a 2-d float32 array (rows may be strided) -- the operand
    magnitude that rows_gemm / rows_wgrad scale by.  Measure a tensor once, pass the word to every product that uses it."""
    _lib.require_device(x, "x")
    if x.dtype != torch.float32 or x.dim() != 2:
        raise TypeError("rows_absmax needs a 2-d float32 tensor")
    if x.stride(1) != 1:
        x = x.contiguous()
    out = torch.empty((1,), dtype=torch.int32, device=x.device)
    lib = _lib.load()
    _lib.check(lib.dgcnn_rows_absmax(_lib.ptr(x), x.stride(0), x.shape[0], x.shape[1], _lib.ptr(out),
                                     _lib.stream_ptr(x.device)), "dgcnn_rows_absmax")
    return out


def rows_stats(x):
    """(amax word, column sums [cols]) of a 2-d float32 array in ONE pass (dgcnn_rows_stats): the operand magnitude every
    product that reads x scales by, and sum_p x[p, :] (conv6's backward needs it of cat).  cols a multiple of 4, <= 256."""
    _lib.require_device(x, "x")
    if x.dtype != torch.float32 or x.dim() != 2:
        raise TypeError("rows_stats needs a 2-d float32 tensor")
    if x.stride(1) != 1 or x.stride(0) % 4 or x.data_ptr() % 16:
        x = x.contiguous()
    rows, cols = x.shape
    lib = _lib.load()
    amax = torch.empty((1,), dtype=torch.int32, device=x.device)
    colsum = torch.empty((cols,), dtype=torch.float32, device=x.device)
    ws_bytes = lib.dgcnn_rows_stats_workspace_bytes(rows, cols)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=x.device)
    _lib.check(lib.dgcnn_rows_stats(_lib.ptr(x), x.stride(0), rows, cols, _lib.ptr(amax), _lib.ptr(colsum), _lib.ptr(ws), ws_bytes,
                                    _lib.stream_ptr(x.device)), "dgcnn_rows_stats")
    return amax, colsum


def rows_wgrad(a, b, amax_a=None, amax_b=None):
    """sum over the rows p of a[p, :]^T b[p, :]:  a [P, M], b [P, N] (rows may be strided views) -> [M, N].
    The weight-gradient product of a 1x1 convolution on point-major rows, split over the points (dgcnn_rows_wgrad);
    amax_a / amax_b: the operands' rows_absmax words, when already known."""
    _lib.require_device(a, "a")
    _lib.require_device(b, "b")
    if a.dtype != torch.float32 or b.dtype != torch.float32:
        raise TypeError("rows_wgrad needs float32 operands")
    if a.stride(1) != 1:
        a = a.contiguous()
    if b.stride(1) != 1:
        b = b.contiguous()
    P, M = a.shape
    N = b.shape[1]
    if b.shape[0] != P:
        raise ValueError("a and b must have the same number of rows")
    lib = _lib.load()
    dw = torch.empty((M, N), dtype=torch.float32, device=a.device)
    ws_bytes = lib.dgcnn_rows_wgrad_workspace_bytes(M, N, P)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=a.device)
    _lib.check(lib.dgcnn_rows_wgrad(_lib.ptr(a), a.stride(0), M, _lib.ptr(b), b.stride(0), N, P,
                                    _lib.ptr(amax_a) if amax_a is not None else None,
                                    _lib.ptr(amax_b) if amax_b is not None else None, _lib.ptr(dw), _lib.ptr(ws), ws_bytes,
                                    _lib.stream_ptr(a.device)), "dgcnn_rows_wgrad")
    return dw


def rows_gemm(x, w, add=None, rows_per_add=0, amax_x=None, amax_w=None):
    """x [P, K] . w [N, K]^T (+ add[row // rows_per_add]) -> [P, N] on the fp16 matrix cores with split operands
    (dgcnn_rows_gemm; fp32 in, fp32 out).  x and w may be views with a row stride; K a multiple of 32.  amax_x / amax_w:
    the operands' rows_absmax words, when already known."""
    _lib.require_device(x, "x")
    _lib.require_device(w, "w")
    if x.dtype != torch.float32 or w.dtype != torch.float32:
        raise TypeError("rows_gemm needs float32 operands")
    if x.stride(1) != 1 or x.stride(0) % 4 or x.data_ptr() % 16:
        x = x.contiguous()
    if w.stride(1) != 1 or w.stride(0) % 4 or w.data_ptr() % 16:
        w = w.contiguous()
    P, K = x.shape
    N = w.shape[0]
    if w.shape[1] != K:
        raise ValueError("x [P, K] and w [N, K] disagree on K")
    if add is not None:
        add = add.contiguous()
        if add.shape[1] != N or add.shape[0] * rows_per_add < P:
            raise ValueError("add must be [ceil(P / rows_per_add), N]")
    if amax_x is None:
        amax_x = rows_absmax(x)
    if amax_w is None:
        amax_w = rows_absmax(w)
    lib = _lib.load()
    y = torch.empty((P, N), dtype=torch.float32, device=x.device)
    _lib.check(lib.dgcnn_rows_gemm(_lib.ptr(x), x.stride(0), _lib.ptr(w), w.stride(0), P, K, N, _lib.ptr(amax_x),
                                   _lib.ptr(amax_w), _lib.ptr(add) if add is not None else None, int(rows_per_add),
                                   _lib.ptr(y), N, _lib.stream_ptr(x.device)), "dgcnn_rows_gemm")
    return y


def _gemm_ok(K):
    return K % 32 == 0


class _LinearRows(torch.autograd.Function):
    """y = x W^T (+ add[b] on the rows of batch element b) on point-major rows (a 1x1 Conv1d without bias,
    model.py:132-141): forward and dL/dx through dgcnn_rows_gemm (fp16 matrix cores, split operands); dL/dW, a
    contraction over the points with a small output, through rows_wgrad.  (A contraction length that is not a multiple of
    32 -- conv9's five classes in dL/dx -- falls to the library product.)"""

    @staticmethod
    def forward(ctx, x, w, add, rows_per_add):
        ctx.rows_per_add = rows_per_add
        ctx.has_add = add is not None
        xd, wd = x.detach(), w.detach()
        # the operand magnitudes: measured once here, used again by the two products of the backward
        ax, aw = rows_absmax(xd), rows_absmax(wd)
        ctx.save_for_backward(x, w, ax, aw)
        if _gemm_ok(x.shape[1]):
            return rows_gemm(xd, wd, add.detach() if add is not None else None, rows_per_add, ax, aw)
        y = torch.mm(xd, wd.t())
        if add is not None:
            y = (y.view(add.shape[0], rows_per_add, -1) + add.detach().unsqueeze(1)).view(y.shape)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, ax, aw = ctx.saved_tensors
        gy = gy.contiguous()
        ag = rows_absmax(gy)
        gx = gw = gadd = None
        if ctx.needs_input_grad[0]:
            if _gemm_ok(gy.shape[1]):
                gx = rows_gemm(gy, w.detach().t().contiguous(), amax_x=ag, amax_w=aw)   # (max |W^T| = max |W|)
            else:
                gx = torch.mm(gy, w)
        if ctx.needs_input_grad[1]:
            gw = rows_wgrad(gy, x, ag, ax)
        if ctx.has_add and ctx.needs_input_grad[2]:
            gadd = gy.view(-1, ctx.rows_per_add, gy.shape[1]).sum(1)
        return gx, gw, gadd, None


def linear_rows(x, w, add=None, rows_per_add=0):
    """x [P, C_in] point-major rows, w [C_out, C_in] -> [P, C_out]; add [B, C_out] (optional) is added to the
    rows_per_add rows of each batch element."""
    return _LinearRows.apply(x, w, add, int(rows_per_add))


class _Classifier(torch.autograd.Function):
    """conv9 (model.py:141, :191): x [P, K] . w [NC, K]^T -> [P, NC] for NC <= 8 classes; streaming kernels
    (dgcnn_cls_fwd / dgcnn_cls_bwd: dL/dx and dL/dw in one pass over x and dL/dy)."""

    @staticmethod
    def forward(ctx, x, w):
        P, K = x.shape
        NC = w.shape[0]
        lib = _lib.load()
        xd, wd = x.detach().contiguous(), w.detach().contiguous()
        y = torch.empty((P, NC), dtype=torch.float32, device=x.device)
        _lib.check(lib.dgcnn_cls_fwd(_lib.ptr(xd), _lib.ptr(wd), P, K, NC, _lib.ptr(y), _lib.stream_ptr(x.device)), "dgcnn_cls_fwd")
        ctx.save_for_backward(xd, wd)
        return y

    @staticmethod
    def backward(ctx, gy):
        xd, wd = ctx.saved_tensors
        P, K = xd.shape
        NC = wd.shape[0]
        lib = _lib.load()
        gy = gy.contiguous()
        gx = torch.empty_like(xd) if ctx.needs_input_grad[0] else None
        gw = torch.empty_like(wd) if ctx.needs_input_grad[1] else None
        if gx is None and gw is None:
            return None, None
        ws_bytes = lib.dgcnn_cls_bwd_workspace_bytes(P, K, NC)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=xd.device)
        _lib.check(lib.dgcnn_cls_bwd(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(gy), P, K, NC,
                                     _lib.ptr(gx) if gx is not None else None, _lib.ptr(gw) if gw is not None else None,
                                     _lib.ptr(ws), ws_bytes, _lib.stream_ptr(xd.device)), "dgcnn_cls_bwd")
        return gx, gw


def classifier_rows(x, w):
    """the classifier on point-major rows: x [P, K], w [num_classes, K] -> logits [P, num_classes]"""
    _lib.require_device(x, "x")
    if x.dtype != torch.float32 or w.dtype != torch.float32:
        raise TypeError("classifier_rows needs float32 operands")
    if w.shape[0] > 8 or x.shape[1] % 4:
        return torch.nn.functional.linear(x, w)   # more classes than the streaming kernel holds: the library product
    return _Classifier.apply(x, w)


class _EmbedMax(torch.autograd.Function):
    """conv6 -> bn6 -> LeakyReLU -> max over the points (model.py:172-175) as ONE autograd node, so that the backward
    never forms the dense [B*N, emb] gradient: with u = cat W^T, dL/du[p, c] = A_c + B_c u[p, c] + (s[b, c] at the
    winning point of (b, c)) -- the BatchNorm mean terms are affine in u -- hence

        dcat = 1 (A W)^T + cat (W^T diag(B) W) + [winners: s W rows]
        dW   = A (sum_p cat)^T + diag(B) W (cat^T cat) + [winners: s cat rows]

    i.e. two 192 x 192 products over the points instead of two 1024-wide ones (44 -> 8 GFLOP at B = 8, N = 7000).  The
    forward is one kernel pass (dgcnn_embed_max_fwd): the pre-activation u ([B*N, emb], 229 MB there) never leaves the
    matrix kernel's accumulators -- column sums and the running extreme are its epilogue."""

    @staticmethod
    def forward(ctx, cat, w, weight, bias, rmean, rvar, B, N, training, eps, momentum, slope, amax_cat=None, catsum=None):
        P, F_ = cat.shape
        C = w.shape[0]
        lib = _lib.load()
        dev = cat.device
        wd = w.detach()
        catd = cat.detach()
        wc, bc = weight.detach().contiguous(), bias.detach().contiguous()
        bn = _bn_struct(wc, bc, rmean, rvar)
        g = torch.empty((B, C), dtype=torch.float32, device=dev)
        arg = torch.empty((B, C), dtype=torch.int32, device=dev)
        stats = torch.empty((4, C), dtype=torch.float32, device=dev)
        # max |cat|: this product and the two 192-wide ones of the backward (the caller's word when it has measured cat)
        ac = amax_cat if amax_cat is not None else rows_absmax(catd)
        fused = _gemm_ok(F_) and C % 4 == 0 and catd.stride(0) % 4 == 0 and catd.data_ptr() % 16 == 0
        # one call forward and one call backward (dgcnn_embed_fwd / _bwd) where the shape allows it
        one_call = (fused and N >= 128 and F_ <= 256 and C <= 8192 and P < (1 << 23) and wd.is_contiguous() and wd.data_ptr() % 16 == 0
                    and catd.is_contiguous() and (catsum is not None or not training))
        ctx.one_call = one_call
        if one_call:
            uwin = torch.empty((B, C), dtype=torch.float32, device=dev)
            ws_bytes = lib.dgcnn_embed_fwd_workspace_bytes(B, N, F_, C)
            ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
            _lib.check(lib.dgcnn_embed_fwd(_lib.ptr(catd), catd.stride(0), _lib.ptr(wd), wd.stride(0), B, N, F_, C, _lib.ptr(ac),
                                           ctypes.byref(bn), 1 if training else 0, eps, momentum, slope, _lib.ptr(g), _lib.ptr(arg),
                                           _lib.ptr(uwin), _lib.ptr(stats), _lib.ptr(ws), ws_bytes, _lib.stream_ptr(dev)),
                       "dgcnn_embed_fwd")
        elif fused:
            # ONE pass: the [P, emb] product never leaves the matrix kernel's accumulators (dgcnn_embed_max_fwd)
            if wd.stride(1) != 1 or wd.stride(0) % 4 or wd.data_ptr() % 16:
                wd = wd.contiguous()
            uwin = torch.empty((B, C), dtype=torch.float32, device=dev)
            ws_bytes = lib.dgcnn_embed_max_fwd_workspace_bytes(B, N, C)
            ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
            _lib.check(lib.dgcnn_embed_max_fwd(_lib.ptr(catd), catd.stride(0), _lib.ptr(wd), wd.stride(0), B, N, F_, C,
                                               _lib.ptr(ac), _lib.ptr(rows_absmax(wd)), ctypes.byref(bn), 1 if training else 0,
                                               eps, momentum, slope, _lib.ptr(g), _lib.ptr(arg), _lib.ptr(uwin),
                                               _lib.ptr(stats), _lib.ptr(ws), ws_bytes, _lib.stream_ptr(dev)),
                       "dgcnn_embed_max_fwd")
        else:
            u = torch.mm(catd, wd.t())
            ws_bytes = lib.dgcnn_bn_lrelu_max_workspace_bytes(B, N, C)
            ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
            _lib.check(lib.dgcnn_bn_lrelu_max_fwd(_lib.ptr(u), B, N, C, ctypes.byref(bn), 1 if training else 0, eps, momentum,
                                                  slope, _lib.ptr(g), _lib.ptr(arg), _lib.ptr(stats), _lib.ptr(ws), ws_bytes,
                                                  _lib.stream_ptr(dev)), "dgcnn_bn_lrelu_max_fwd")
            uwin = torch.gather(u.view(B, N, C), 1, arg.long().unsqueeze(1)).squeeze(1).contiguous()   # [B, C]
        ctx.save_for_backward(cat, w, arg, stats, uwin, ac, catsum if (one_call and training) else None)
        ctx.cfg = (B, N, training, slope)
        return g

    @staticmethod
    def backward(ctx, gg):
        cat, w, arg, stats, uwin, ac, catsum = ctx.saved_tensors
        B, N, training, slope = ctx.cfg
        P, F_ = cat.shape
        C = w.shape[0]
        lib = _lib.load()
        dev = cat.device
        st = _lib.stream_ptr(dev)
        gg = gg.contiguous()
        gw_bn = torch.empty((C,), dtype=torch.float32, device=dev)
        gb_bn = torch.empty((C,), dtype=torch.float32, device=dev)
        if ctx.one_call:
            catd, wd = cat.detach(), w.detach()
            dcat = torch.empty((P, F_), dtype=torch.float32, device=dev)
            dw = torch.empty((C, F_), dtype=torch.float32, device=dev)
            ws_bytes = lib.dgcnn_embed_bwd_workspace_bytes(B, N, F_, C)
            ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
            _lib.check(lib.dgcnn_embed_bwd(_lib.ptr(catd), catd.stride(0), _lib.ptr(ac), _lib.ptr(catsum) if catsum is not None else None,
                                           _lib.ptr(wd), wd.stride(0), _lib.ptr(arg), _lib.ptr(uwin), _lib.ptr(stats), _lib.ptr(gg),
                                           B, N, F_, C, 1 if training else 0, slope, _lib.ptr(dcat), F_, _lib.ptr(dw),
                                           _lib.ptr(gw_bn), _lib.ptr(gb_bn), _lib.ptr(ws), ws_bytes, st), "dgcnn_embed_bwd")
            return dcat, dw.view_as(w), gw_bn, gb_bn, None, None, None, None, None, None, None, None, None, None
        coef_a = torch.empty((C,), dtype=torch.float32, device=dev)
        coef_b = torch.empty((C,), dtype=torch.float32, device=dev)
        s = torch.empty((B, C), dtype=torch.float32, device=dev)
        _lib.check(lib.dgcnn_embed_max_bwd_prepare(_lib.ptr(uwin), _lib.ptr(gg), _lib.ptr(stats), B, N, C,
                                                   1 if training else 0, slope, _lib.ptr(gw_bn), _lib.ptr(gb_bn),
                                                   _lib.ptr(coef_a), _lib.ptr(coef_b), _lib.ptr(s), st),
                   "dgcnn_embed_max_bwd_prepare")
        wd = w.detach().contiguous()
        catd = cat.detach()
        rows = (arg.long() + (torch.arange(B, device=dev) * N).unsqueeze(1)).reshape(-1)     # [B*C] global points
        win_rows = catd.index_select(0, rows).view(B, C, F_)                                 # cat at the winners
        dw = (s.unsqueeze(2) * win_rows).sum(0)                                              # [C, F], fixed order
        if training:
            wb = wd * coef_b.unsqueeze(1)                                                    # diag(B) W
            m = torch.mm(wd.t(), wb)                                                         # M = W^T diag(B) W (symmetric)
            if _gemm_ok(F_):
                dcat = rows_gemm(catd, m, torch.mv(wd.t(), coef_a).unsqueeze(0), P, amax_x=ac)   # 1 (A W)^T + cat M
            else:
                dcat = torch.addmm(torch.mv(wd.t(), coef_a), catd, m)
            dw.addmm_(wb, rows_wgrad(catd, catd, ac, ac))                                    # + diag(B) W (cat^T cat)
            dw.addr_(coef_a, catd.sum(0))                                                    # + A (sum cat)^T
        else:
            dcat = torch.zeros_like(catd)
        _lib.check(lib.dgcnn_embed_max_bwd_scatter(_lib.ptr(arg), _lib.ptr(s), _lib.ptr(wd), B, N, C, F_, _lib.ptr(dcat), st),
                   "dgcnn_embed_max_bwd_scatter")
        return dcat, dw.view_as(w), gw_bn, gb_bn, None, None, None, None, None, None, None, None, None, None


def embed_max_rows(cat, w, bn, B, N, training, slope=0.2, amax_cat=None, catsum=None):
    """g[b] = max over the N points of batch element b of LeakyReLU(BatchNorm1d(cat W^T)): cat point-major rows
    [B*N, F], w [emb, F] (conv6's weight) -> [B, emb]; `bn` is the nn.BatchNorm1d module (running statistics updated
    in place when training)."""
    cat = _f32c(cat, "cat")
    if cat.shape[0] != B * N:
        raise ValueError("cat must have B*N rows")
    use_batch_stats = training or bn.running_mean is None
    momentum = bn.momentum if bn.momentum is not None else 0.1
    g = _EmbedMax.apply(cat, w, bn.weight, bn.bias, bn.running_mean, bn.running_var, int(B), int(N),
                        bool(use_batch_stats), float(bn.eps), float(momentum), float(slope), amax_cat, catsum)
    if training:
        _count_batch(bn)
    return g


# ------------------------------------------------------------------------------------------------
class _HeadTail(torch.autograd.Function):
    """conv7 -> bn7 -> LeakyReLU -> conv8 -> bn8 -> LeakyReLU -> dropout -> conv9 (model.py:179-191) on point-major rows as
    ONE forward and ONE backward call (dgcnn_head_tail_fwd / _bwd, head_fused.hip): only the pre-activations u7 and u8 are
    written; BatchNorm statistics, BatchNorm + LeakyReLU (+ dropout) and every BatchNorm input gradient are prologues /
    epilogues of the products."""

    @staticmethod
    def forward(ctx, cat, add7, w7c, g7, b7, w8, g8, b8, w9, rm7, rv7, rm8, rv8, amax_cat, B, N, training, eps7, eps8, mom7,
                mom8, slope, p_drop, seed):
        P, F_ = cat.shape
        C7, C8, NC = w7c.shape[0], w8.shape[0], w9.shape[0]
        lib = _lib.load()
        dev = cat.device
        tens = [t.detach().contiguous() for t in (g7, b7, g8, b8)]
        w8c, w9c = w8.detach().contiguous(), w9.detach().contiguous()
        w7d = w7c.detach()
        if w7d.stride(1) != 1 or w7d.stride(0) % 4 or w7d.data_ptr() % 16:
            w7d = w7d.contiguous()
        catd = cat.detach()
        d = _lib.HeadTailDesc()
        d.B, d.N, d.F, d.C7, d.C8, d.NC = B, N, F_, C7, C8, NC
        d.ldcat, d.ldw7 = catd.stride(0), w7d.stride(0)
        d.training = 1 if training else 0
        d.eps7, d.eps8, d.momentum7, d.momentum8, d.slope, d.p_drop = eps7, eps8, mom7, mom8, slope, p_drop
        d.seed = seed
        d.w7c, d.w8, d.w9 = w7d.data_ptr(), w8c.data_ptr(), w9c.data_ptr()
        d.bn7 = _bn_struct(tens[0], tens[1], rm7, rv7)
        d.bn8 = _bn_struct(tens[2], tens[3], rm8, rv8)
        logits = torch.empty((P, NC), dtype=torch.float32, device=dev)
        saved = torch.empty((lib.dgcnn_head_tail_saved_bytes(d),), dtype=torch.uint8, device=dev)
        ws_bytes = lib.dgcnn_head_tail_fwd_workspace_bytes(d)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        addd = add7.detach().contiguous()
        _lib.check(lib.dgcnn_head_tail_fwd(d, _lib.ptr(catd), _lib.ptr(amax_cat), _lib.ptr(addd), _lib.ptr(logits), _lib.ptr(saved),
                                           _lib.ptr(ws), ws_bytes, _lib.stream_ptr(dev)), "dgcnn_head_tail_fwd")
        ctx.save_for_backward(cat, amax_cat, saved, w7d, w8c, w9c, *tens, rm7, rv7, rm8, rv8)
        ctx.cfg = (B, N, training, eps7, eps8, mom7, mom8, slope, p_drop, seed, tuple(w7c.shape), tuple(w8.shape), tuple(w9.shape))
        return logits

    @staticmethod
    def backward(ctx, gl):
        cat, amax_cat, saved, w7d, w8c, w9c, g7c, b7c, g8c, b8c, rm7, rv7, rm8, rv8 = ctx.saved_tensors
        B, N, training, eps7, eps8, mom7, mom8, slope, p_drop, seed, w7_shape, w8_shape, w9_shape = ctx.cfg
        P, F_ = cat.shape
        C7, C8, NC = w7d.shape[0], w8c.shape[0], w9c.shape[0]
        lib = _lib.load()
        dev = cat.device
        catd = cat.detach()
        d = _lib.HeadTailDesc()
        d.B, d.N, d.F, d.C7, d.C8, d.NC = B, N, F_, C7, C8, NC
        d.ldcat, d.ldw7 = catd.stride(0), w7d.stride(0)
        d.training = 1 if training else 0
        d.eps7, d.eps8, d.momentum7, d.momentum8, d.slope, d.p_drop = eps7, eps8, mom7, mom8, slope, p_drop
        d.seed = seed
        d.w7c, d.w8, d.w9 = w7d.data_ptr(), w8c.data_ptr(), w9c.data_ptr()
        d.bn7 = _bn_struct(g7c, b7c, rm7, rv7)
        d.bn8 = _bn_struct(g8c, b8c, rm8, rv8)
        gl = gl.contiguous()
        g = _lib.HeadTailGrads()
        gcat = torch.empty((P, F_), dtype=torch.float32, device=dev)
        gadd = torch.empty((B, C7), dtype=torch.float32, device=dev)
        gw7 = torch.empty((C7, F_), dtype=torch.float32, device=dev)
        gw8, gw9 = torch.empty_like(w8c), torch.empty_like(w9c)
        gg7, gb7, gg8, gb8 = torch.empty_like(g7c), torch.empty_like(b7c), torch.empty_like(g8c), torch.empty_like(b8c)
        g.gcat, g.ldgcat, g.gadd7, g.gw7c = gcat.data_ptr(), F_, gadd.data_ptr(), gw7.data_ptr()
        g.gbn7_w, g.gbn7_b, g.gw8, g.gbn8_w, g.gbn8_b, g.gw9 = (gg7.data_ptr(), gb7.data_ptr(), gw8.data_ptr(), gg8.data_ptr(),
                                                               gb8.data_ptr(), gw9.data_ptr())
        ws_bytes = lib.dgcnn_head_tail_bwd_workspace_bytes(d)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        _lib.check(lib.dgcnn_head_tail_bwd(d, _lib.ptr(catd), _lib.ptr(amax_cat), _lib.ptr(gl), _lib.ptr(saved), ctypes.byref(g),
                                           _lib.ptr(ws), ws_bytes, _lib.stream_ptr(dev)), "dgcnn_head_tail_bwd")
        return (gcat, gadd, gw7.view(w7_shape), gg7, gb7, gw8.view(w8_shape), gg8, gb8, gw9.view(w9_shape),
                None, None, None, None, None, None, None, None, None, None, None, None, None, None, None)


def head_tail_supported(B, N, F_, C7, C8, NC):
    """shapes dgcnn_head_tail_* takes (include/dgcnn_hip.h); others go through the per-layer operators"""
    return (N >= 128 and 1 <= B <= 64 and B * N < (1 << 23) and F_ % 32 == 0 and C7 % 32 == 0 and C8 % 32 == 0
            and C7 <= 1024 and C8 <= 1024 and 1 <= NC <= 8)


def head_tail_rows(cat, add7, w7c, bn7, w8, bn8, w9, B, N, training, slope=0.2, p_drop=0.0, amax_cat=None):
    """logits [B*N, classes] of the head's tail on point-major rows: cat [B*N, F], add7 [B, C7] (conv7's share of the pooled
    feature), w7c [C7, F] (a view of conv7's weight), w8 [C8, C7], w9 [classes, C8]; bn7 / bn8 the nn.BatchNorm1d modules
    (running statistics updated in place when training).  Dropout (training only) draws its 64-bit seed from torch's CPU
    generator, so torch.manual_seed makes a step reproducible; the mask itself is dgcnn_dropout_mask's."""
    cat = _f32c(cat, "cat")
    if cat.shape[0] != B * N:
        raise ValueError("cat must have B*N rows")
    if amax_cat is None:
        amax_cat = rows_absmax(cat.detach())
    use_batch_stats = training or bn7.running_mean is None
    mom = [bn.momentum if bn.momentum is not None else 0.1 for bn in (bn7, bn8)]
    drop = float(p_drop) if training else 0.0
    seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64)) if drop > 0.0 else 0
    y = _HeadTail.apply(cat, add7, w7c, bn7.weight, bn7.bias, w8, bn8.weight, bn8.bias, w9, bn7.running_mean, bn7.running_var,
                        bn8.running_mean, bn8.running_var, amax_cat, int(B), int(N), bool(use_batch_stats), float(bn7.eps),
                        float(bn8.eps), float(mom[0]), float(mom[1]), float(slope), drop, seed)
    if training:
        _count_batch(bn7)
        _count_batch(bn8)
    return y


def dropout_mask(seed, P, K, p_drop, device):
    """the keep mask (uint8 [P, K]) dgcnn_head_tail_* uses for (seed, p_drop)"""
    mask = torch.empty((P, K), dtype=torch.uint8, device=device)
    lib = _lib.load()
    _lib.check(lib.dgcnn_dropout_mask(int(seed), P, K, float(p_drop), _lib.ptr(mask), _lib.stream_ptr(device)), "dgcnn_dropout_mask")
    return mask


# ------------------------------------------------------------------------------------------------
# The whole network at depth 10 as one forward and one backward call (dgcnn_model_fwd / _bwd, model.hip)
def _model_desc(mods, B, N, F_, k, training, slope, p_drop, seed, idx, idx_given, params, counters=True):
    """params: the 25 tensors in the order conv1..conv9 weights, bn1..bn8 weights, bn1..bn8 biases (detached, contiguous)"""
    d = _lib.ModelDesc()
    d.B, d.N, d.F, d.k = B, N, F_, k
    w = params[:9]
    d.num_classes, d.emb, d.c7, d.c8 = w[8].shape[0], w[5].shape[0], w[6].shape[0], w[7].shape[0]
    d.training, d.idx_given = (1 if training else 0), (1 if idx_given else 0)
    d.slope, d.p_drop, d.seed = slope, p_drop, seed
    for i in range(9):
        d.w[i] = w[i].data_ptr()
    for i, bn in enumerate(mods):
        nbt = bn.num_batches_tracked if (counters and training) else None
        d.bn[i] = _bn_struct(params[9 + i], params[17 + i], bn.running_mean, bn.running_var, nbt)
        d.eps[i] = float(bn.eps)
        d.momentum[i] = float(bn.momentum if bn.momentum is not None else 0.1)
    for b in range(3):
        d.idx[b] = idx[b].data_ptr()
    return d


class _DGCNNFull(torch.autograd.Function):
    """DGCNN.forward(x, depth=10) (model.py:144-193) through dgcnn_model_fwd / dgcnn_model_bwd: one C-ABI call each way.
    x gets no gradient (the reference's loops never ask for one)."""

    @staticmethod
    def forward(ctx, x, mods, k, training, slope, p_drop, seed, idx, idx_given, *params):
        # idx: three int32 [B, N, k] tensors on the device -- the graphs to use (idx_given) or the buffers the forward fills
        B, F_, N = x.shape
        lib = _lib.load()
        dev = x.device
        pd = [p.detach().contiguous() for p in params]
        pd = [t.reshape(t.shape[0], -1) if t.dim() > 1 else t for t in pd]
        d = _model_desc(mods, B, N, F_, k, training, slope, p_drop, seed, idx, idx_given, pd)
        nc = pd[8].shape[0]
        logits = torch.empty((B, nc, N), dtype=torch.float32, device=dev)
        nsaved = lib.dgcnn_model_saved_bytes(d)
        ws_bytes = lib.dgcnn_model_fwd_workspace_bytes(d)
        if nsaved == 0 or ws_bytes == 0:
            _lib.check(1, "dgcnn_model_fwd (shape)")
        saved = torch.empty((nsaved,), dtype=torch.uint8, device=dev)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        _lib.check(lib.dgcnn_model_fwd(d, _lib.ptr(x), x.stride(0), x.stride(1), x.stride(2), _lib.ptr(logits), logits.stride(0),
                                       logits.stride(1), logits.stride(2), _lib.ptr(saved), _lib.ptr(ws), ws_bytes,
                                       _lib.stream_ptr(dev)), "dgcnn_model_fwd")
        ctx.save_for_backward(x, saved, *idx, *pd)
        ctx.cfg = (mods, k, training, slope, p_drop, seed, [tuple(p.shape) for p in params])
        ctx.sinks = tuple((p, getattr(p, "_dgcnn_grad_sink", None)) for p in params)
        return logits

    @staticmethod
    def backward(ctx, gl):
        x, saved, i0, i1, i2, *pd = ctx.saved_tensors
        mods, k, training, slope, p_drop, seed, shapes = ctx.cfg
        B, F_, N = x.shape
        lib = _lib.load()
        dev = x.device
        if gl.dtype != torch.float32:
            gl = gl.float()
        d = _model_desc(mods, B, N, F_, k, training, slope, p_drop, seed, (i0, i1, i2), True, pd, counters=False)
        outs = []
        for (owner, sink), ref in zip(ctx.sinks, pd):
            # a parameter's slice of the flat exchange buffer, if its owner has named one and nothing has claimed it in this
            # cycle (see _EdgeConv.backward)
            if (sink is not None and owner.grad is None and not getattr(owner, "_dgcnn_sink_claimed", False) and owner.requires_grad
                    and sink.numel() == ref.numel() and sink.dtype == ref.dtype and sink.device == ref.device and sink.is_contiguous()):
                owner._dgcnn_sink_claimed = True
                outs.append(sink.detach().view(ref.shape))
            else:
                outs.append(torch.empty_like(ref))
        g = _lib.ModelGrads()
        for i in range(9):
            g.gw[i] = outs[i].data_ptr()
        for i in range(8):
            g.gbn_w[i] = outs[9 + i].data_ptr()
            g.gbn_b[i] = outs[17 + i].data_ptr()
        ws_bytes = lib.dgcnn_model_bwd_workspace_bytes(d)
        if ws_bytes == 0:
            _lib.check(1, "dgcnn_model_bwd (shape)")
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)

        def run(phases):
            _lib.check(lib.dgcnn_model_bwd(d, _lib.ptr(x), x.stride(0), x.stride(1), x.stride(2), _lib.ptr(gl), gl.stride(0),
                                           gl.stride(1), gl.stride(2), _lib.ptr(saved), ctypes.byref(g), _lib.ptr(ws), ws_bytes,
                                           phases, _lib.stream_ptr(dev)), "dgcnn_model_bwd")

        # data parallel: when conv6 .. conv9 (the last four tensors of the flat exchange buffer, 97 % of its bytes) have been
        # written into their slices, their all-reduce can start as soon as the head's backward is enqueued and runs under the
        # three blocks' backward (parallel.FlatGradAllReduce(overlap=True))
        sync = getattr(ctx.sinks[5][0], "_dgcnn_grad_sync", None)
        in_place = all(outs[i].data_ptr() == ctx.sinks[i][1].data_ptr() for i in (5, 6, 7, 8) if ctx.sinks[i][1] is not None) and \
            all(ctx.sinks[i][1] is not None for i in (5, 6, 7, 8))
        if sync is not None and in_place and sync.wants_overlap():
            run(1)
            sync.reduce_tail_async(ctx.sinks[5][0], [ctx.sinks[i][0] for i in (5, 6, 7, 8)])
            run(2)
        else:
            run(3)
        return (None,) * 9 + tuple(o.view(sh) for o, sh in zip(outs, shapes))


def model_full_supported(B, F_, N, k, emb, c7, c8, nc):
    """shapes dgcnn_model_fwd / _bwd take (include/dgcnn_hip.h); others go through the per-layer operators"""
    return (1 <= B <= 64 and N >= 128 and 1 <= F_ <= 64 and 1 <= k <= min(64, N) and B * N < (1 << 23) and emb % 32 == 0
            and 32 <= emb <= 8192 and c7 % 32 == 0 and c8 % 32 == 0 and 32 <= c7 <= 1024 and 32 <= c8 <= 1024 and 1 <= nc <= 8
            and N <= 40000)


def dgcnn_full(x, convs, bns, k, training, slope, p_drop, idx=None):
    """(logits [B, classes, N], the three neighbour graphs used) of the whole network at depth 10.
    x [B, F, N] float32 on the HIP device (any strides); convs: the nine convolution weights (conv1 .. conv9), bns: the eight
    BatchNorm modules (running statistics and counters updated in place when training); idx: optional list of three int
    [B, N, k] graphs to use instead of searching."""
    _lib.require_device(x, "x")
    B, _, N = x.shape
    dev = x.device
    drop = float(p_drop) if training else 0.0
    seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64)) if drop > 0.0 else 0
    params = list(convs) + [bn.weight for bn in bns] + [bn.bias for bn in bns]
    if idx is not None:
        graphs = [t.to(device=dev, dtype=torch.int32).contiguous() for t in idx]
    else:
        # ONE array for the three graphs: the backward then builds their reverse lists in one go (dgcnn_model_bwd)
        graphs = list(torch.empty((3, B, N, int(k)), dtype=torch.int32, device=dev).unbind(0))
    y = _DGCNNFull.apply(x, tuple(bns), int(k), bool(training), float(slope), drop, seed, graphs, idx is not None, *params)
    return y, graphs


def model_full_ready(bns, training):
    """the module state dgcnn_full needs: running statistics present, counters int64 on the device"""
    for bn in bns:
        if bn.running_mean is None or bn.running_var is None or not bn.running_mean.is_cuda:
            return False
        t = bn.num_batches_tracked
        if training and (t is None or not t.is_cuda or t.dtype != torch.int64):
            return False
    return True


# ------------------------------------------------------------------------------------------------
class _ToPointMajor(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return to_point_major(x)

    @staticmethod
    def backward(ctx, g):
        return from_point_major(g)


class _FromPointMajor(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_pm):
        return from_point_major(x_pm)

    @staticmethod
    def backward(ctx, g):
        return to_point_major(g)


def to_point_major_autograd(x):
    return _ToPointMajor.apply(x) if x.requires_grad else to_point_major(x)


def from_point_major_autograd(x_pm):
    return _FromPointMajor.apply(x_pm) if x_pm.requires_grad else from_point_major(x_pm)


# ------------------------------------------------------------------------------------------------
# The steps either side of DGCNN.forward in the reference's loops
def _logit_strides(logits):
    if logits.dim() != 3:
        raise ValueError("logits must be [B, num_classes, N] (what DGCNN.forward returns)")
    _lib.require_device(logits, "logits")
    if logits.dtype != torch.float32:
        raise TypeError("logits must be float32, got %s" % logits.dtype)
    return logits.stride()


class _SegLoss(torch.autograd.Function):
    """softmax -> (training mask) -> cal_loss -> arg-max in one pass; the pass also leaves d loss / d logits."""

    @staticmethod
    def forward(ctx, logits, labels, mask, smoothing):
        B, C, N = logits.shape
        sb, sc, sn = _logit_strides(logits)
        lib = _lib.load()
        dev = logits.device
        loss = torch.empty((), dtype=torch.float32, device=dev)
        grad = torch.empty_strided(logits.shape, logits.stride(), dtype=torch.float32, device=dev)
        pred = torch.empty((B, N), dtype=torch.int64, device=dev)
        ws_bytes = lib.dgcnn_seg_loss_workspace_bytes(B, N)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        _lib.check(lib.dgcnn_seg_loss_labels(_lib.ptr(logits), B, N, C, sb, sc, sn, _lib.ptr(labels),
                                             1 if labels.dtype == torch.float64 else 0,
                                             _lib.ptr(mask) if mask is not None else None, smoothing, _lib.ptr(loss),
                                             _lib.ptr(grad), _lib.ptr(pred), _lib.ptr(ws), ws_bytes, _lib.stream_ptr(dev)),
                   "dgcnn_seg_loss")
        ctx.save_for_backward(grad)
        ctx.mark_non_differentiable(pred)
        return loss, pred

    @staticmethod
    def backward(ctx, gloss, _gpred):
        (grad,) = ctx.saved_tensors
        return grad * gloss, None, None, None


def seg_loss(logits, labels, mask=None, smoothing=0.2):
    """Loss and prediction of one training / validation step of the reference, from the logits DGCNN.forward
    returns ([B, num_classes, N], any dense strides):

        p = softmax(logits, dim=1)                                   train.py:180 / :267
        (mask given) rows (p, 0) with the label where mask != 0, (0, .., 0, 1) with label num_classes elsewhere
                                                                     train.py:182-196
        loss = cal_loss(rows, labels, smoothing=True)                util.py:17-34 (eps = `smoothing`; 0 = cross entropy)
        pred = rows.max(dim=-1)[1]                                   train.py:201 / :271

    labels: integer (or the reference's float64) [B, N]; mask: bool / uint8 / float [B, N] or None.
    Returns (loss: 0-d tensor with autograd to the logits, pred: int64 [B, N])."""
    B, C, N = logits.shape
    dev = logits.device
    # float64 labels (what the reference's loader hands over, data.py:169-182) are read as they are; others become int64
    if labels.dtype == torch.float64:
        labels = labels.to(device=dev).reshape(B, N).contiguous()
    else:
        labels = labels.to(device=dev, dtype=torch.int64).reshape(B, N).contiguous()
    if mask is not None:
        mask = (mask.to(device=dev) != 0).to(torch.uint8).reshape(B, N).contiguous()
    # overlapping / broadcast strides cannot hold a gradient of the same layout
    if not (logits.is_contiguous() or logits.permute(0, 2, 1).is_contiguous()):
        logits = logits.contiguous()
    return _SegLoss.apply(logits, labels, mask, float(smoothing))


def seg_predict(logits, min_class_confidence=0.0, unclassified=31):
    """Inference post-step (test.py:123-127, :308-311): softmax over the classes, maximum, arg-max, and the label
    `unclassified` wherever the maximum probability is below `min_class_confidence`.
    logits [B, num_classes, N] -> (pred int64 [B, N], prob float32 [B, N])."""
    B, C, N = logits.shape
    logits = logits.detach()
    sb, sc, sn = _logit_strides(logits)
    lib = _lib.load()
    dev = logits.device
    pred = torch.empty((B, N), dtype=torch.int64, device=dev)
    prob = torch.empty((B, N), dtype=torch.float32, device=dev)
    _lib.check(lib.dgcnn_seg_predict(_lib.ptr(logits), B, N, C, sb, sc, sn, float(min_class_confidence),
                                     int(unclassified), _lib.ptr(pred), _lib.ptr(prob), _lib.stream_ptr(dev)),
               "dgcnn_seg_predict")
    return pred, prob


def points_to_float(points):
    """float64 point block(s) on the HIP device -> float32, same shape (the `.float()` of train.py:176 / test.py:119;
    the permute around it is a view and stays one)."""
    _lib.require_device(points, "points")
    if points.dtype != torch.float64:
        raise TypeError("points must be float64, got %s" % points.dtype)
    src = points if points.is_contiguous() else points.contiguous()
    out = torch.empty(src.shape, dtype=torch.float32, device=src.device)
    lib = _lib.load()
    _lib.check(lib.dgcnn_cast_f64_f32(_lib.ptr(src), src.numel(), _lib.ptr(out), _lib.stream_ptr(src.device)),
               "dgcnn_cast_f64_f32")
    return out
