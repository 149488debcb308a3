"""ctypes binding of libdgcnn_hip.so (the C ABI declared in include/dgcnn_hip.h).

There is no fallback: if the shared library is missing or a symbol cannot be resolved the
import of the compute path fails loudly, and every call checks the status code and raises
with dgcnn_last_error().  PyTorch only supplies device memory and the current HIP stream.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdgcnn_hip.so")

_c_int, _c_i64, _c_sz, _c_vp, _c_f = ctypes.c_int, ctypes.c_int64, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_float

# name -> (restype, argtypes); must list every symbol of include/dgcnn_hip.h
SIGNATURES = {
    "dgcnn_last_error": (ctypes.c_char_p, []),
    "dgcnn_abi_version": (_c_int, []),
    "dgcnn_to_point_major": (_c_int, [_c_vp, _c_int, _c_int, _c_int, _c_i64, _c_i64, _c_i64, _c_vp, _c_vp]),
    "dgcnn_from_point_major": (_c_int, [_c_vp, _c_int, _c_int, _c_int, _c_vp, _c_i64, _c_i64, _c_i64, _c_vp]),
    "dgcnn_knn_workspace_bytes": (_c_sz, [_c_int, _c_int, _c_int, _c_int]),
    "dgcnn_knn": (_c_int, [_c_vp, _c_int, _c_int, _c_int, _c_int, _c_vp, _c_vp, _c_sz, _c_vp]),
}

_lib = None


class HipExtensionError(RuntimeError):
    pass


def load():
    """Load libdgcnn_hip.so and bind every declared symbol (no compute, works without a GPU)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipExtensionError(
                "libdgcnn_hip.so not found at %s -- build it with `python __graft_entry__.py build` "
                "(there is no CPU or eager fallback for the EdgeConv path)" % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc, what):
    if rc != 0:
        msg = load().dgcnn_last_error()
        raise HipExtensionError("%s failed: %s" % (what, msg.decode() if msg else "rc=%d" % rc))


def stream_ptr(device=None):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def require_device(t, name="tensor"):
    if not t.is_cuda:
        raise HipExtensionError("%s must live on the HIP device (got %s); the EdgeConv path has no CPU fallback"
                                % (name, t.device))
