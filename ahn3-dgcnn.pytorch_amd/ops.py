"""Host-side operators of the EdgeConv path: thin wrappers that allocate outputs/workspaces
with torch and enqueue the HIP kernels of libdgcnn_hip.so on the current stream.

Internal layout is point-major [B, N, C] fp32 ("pm"); neighbour indices are int32, local to
the batch element.
"""
import torch

from . import _lib


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
