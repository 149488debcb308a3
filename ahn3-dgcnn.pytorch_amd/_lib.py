"""ctypes binding of libdgcnn_hip.so (the C ABI declared in include/dgcnn_hip.h).

There is no fallback: if the shared library is missing or a symbol cannot be resolved the
import of the compute path fails loudly, and every call checks the status code and raises
with dgcnn_last_error().  PyTorch only supplies device memory and the current HIP stream.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# (DGCNN_HIP_LIB: another build of the same library -- ablation builds of scripts/ablate_rgather.sh; never a fallback)
LIB_PATH = os.environ.get("DGCNN_HIP_LIB") or os.path.join(_HERE, "libdgcnn_hip.so")

_c_int, _c_i64, _c_sz, _c_vp, _c_f = ctypes.c_int, ctypes.c_int64, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_float



class BnParams(ctypes.Structure):
    """mirror of `dgcnn_bn_params` (include/dgcnn_hip.h)"""
    _fields_ = [("weight", _c_vp), ("bias", _c_vp), ("running_mean", _c_vp), ("running_var", _c_vp),
                ("num_batches_tracked", _c_vp)]


class EdgeConvDesc(ctypes.Structure):
    """mirror of `dgcnn_edgeconv_desc` (include/dgcnn_hip.h)"""
    _fields_ = [("B", ctypes.c_int32), ("N", ctypes.c_int32), ("C", ctypes.c_int32), ("k", ctypes.c_int32),
                ("n_stage", ctypes.c_int32), ("training", ctypes.c_int32),
                ("eps", _c_f), ("momentum", _c_f), ("slope", _c_f),
                ("w1", _c_vp), ("bn1", BnParams), ("w2", _c_vp), ("bn2", BnParams)]


class EdgeConvGrads(ctypes.Structure):
    """mirror of `dgcnn_edgeconv_grads` (include/dgcnn_hip.h)"""
    _fields_ = [("gx", _c_vp), ("gw1", _c_vp), ("gbn1_w", _c_vp), ("gbn1_b", _c_vp),
                ("gw2", _c_vp), ("gbn2_w", _c_vp), ("gbn2_b", _c_vp)]


_c_desc_p = ctypes.POINTER(EdgeConvDesc)


class HeadTailDesc(ctypes.Structure):
    """mirror of `dgcnn_head_tail_desc` (include/dgcnn_hip.h)"""
    _fields_ = [("B", ctypes.c_int32), ("N", ctypes.c_int32), ("F", ctypes.c_int32), ("C7", ctypes.c_int32),
                ("C8", ctypes.c_int32), ("NC", ctypes.c_int32), ("ldcat", ctypes.c_int32), ("ldw7", ctypes.c_int32),
                ("training", ctypes.c_int32),
                ("eps7", _c_f), ("eps8", _c_f), ("momentum7", _c_f), ("momentum8", _c_f), ("slope", _c_f), ("p_drop", _c_f),
                ("seed", ctypes.c_uint64),
                ("w7c", _c_vp), ("bn7", BnParams), ("w8", _c_vp), ("bn8", BnParams), ("w9", _c_vp)]


class HeadTailGrads(ctypes.Structure):
    """mirror of `dgcnn_head_tail_grads` (include/dgcnn_hip.h)"""
    _fields_ = [("gcat", _c_vp), ("ldgcat", _c_i64), ("gadd7", _c_vp), ("gw7c", _c_vp), ("gbn7_w", _c_vp), ("gbn7_b", _c_vp),
                ("gw8", _c_vp), ("gbn8_w", _c_vp), ("gbn8_b", _c_vp), ("gw9", _c_vp)]


_c_ht_p = ctypes.POINTER(HeadTailDesc)


class ModelDesc(ctypes.Structure):
    """mirror of `dgcnn_model_desc` (include/dgcnn_hip.h)"""
    _fields_ = [("B", ctypes.c_int32), ("N", ctypes.c_int32), ("F", ctypes.c_int32), ("k", ctypes.c_int32),
                ("num_classes", ctypes.c_int32), ("emb", ctypes.c_int32), ("c7", ctypes.c_int32), ("c8", ctypes.c_int32),
                ("training", ctypes.c_int32), ("idx_given", ctypes.c_int32),
                ("slope", _c_f), ("p_drop", _c_f), ("seed", ctypes.c_uint64),
                ("w", _c_vp * 9), ("bn", BnParams * 8), ("eps", _c_f * 8), ("momentum", _c_f * 8), ("idx", _c_vp * 3)]


class ModelGrads(ctypes.Structure):
    """mirror of `dgcnn_model_grads` (include/dgcnn_hip.h)"""
    _fields_ = [("gw", _c_vp * 9), ("gbn_w", _c_vp * 8), ("gbn_b", _c_vp * 8)]


_c_md_p = ctypes.POINTER(ModelDesc)

# name -> (restype, argtypes); must list every symbol of include/dgcnn_hip.h
SIGNATURES = {
    "dgcnn_last_error": (ctypes.c_char_p, []),
    "dgcnn_abi_version": (_c_int, []),
    "dgcnn_to_point_major": (_c_int, [_c_vp, _c_int, _c_int, _c_int, _c_i64, _c_i64, _c_i64, _c_vp, _c_vp]),
    "dgcnn_from_point_major": (_c_int, [_c_vp, _c_int, _c_int, _c_int, _c_vp, _c_i64, _c_i64, _c_i64, _c_vp]),
    "dgcnn_knn_workspace_bytes": (_c_sz, [_c_int, _c_int, _c_int, _c_int]),
    "dgcnn_knn": (_c_int, [_c_vp, _c_int, _c_int, _c_int, _c_int, _c_vp, _c_vp, _c_sz, _c_vp]),
    "dgcnn_graph_feature_fwd": (_c_int, [_c_vp, _c_vp, _c_int, _c_int, _c_int, _c_int, _c_vp, _c_vp]),
    "dgcnn_graph_feature_bwd_workspace_bytes": (_c_sz, [_c_int, _c_int, _c_int, _c_int]),
    "dgcnn_graph_feature_bwd": (_c_int, [_c_vp, _c_vp, _c_int, _c_int, _c_int, _c_int, _c_vp, _c_vp, _c_sz, _c_vp]),
    "dgcnn_edgeconv_saved_bytes": (_c_sz, [_c_desc_p]),
    "dgcnn_edgeconv_fwd_workspace_bytes": (_c_sz, [_c_desc_p]),
    "dgcnn_edgeconv_bwd_workspace_bytes": (_c_sz, [_c_desc_p]),
    "dgcnn_edgeconv_fwd": (_c_int, [_c_desc_p, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_sz, _c_vp]),
    "dgcnn_edgeconv_bwd": (_c_int, [_c_desc_p, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp,
                                    ctypes.POINTER(EdgeConvGrads), _c_vp, _c_sz, _c_vp]),
    "dgcnn_bn_lrelu_workspace_bytes": (_c_sz, [_c_int, _c_int]),
    "dgcnn_bn_lrelu_fwd": (_c_int, [_c_vp, _c_int, _c_int, ctypes.POINTER(BnParams), _c_int, _c_f, _c_f, _c_f,
                                    _c_vp, _c_vp, _c_vp, _c_sz, _c_vp]),
    "dgcnn_bn_lrelu_bwd": (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_int, _c_int, _c_int, _c_f, _c_vp, _c_vp, _c_vp,
                                    _c_vp, _c_sz, _c_vp]),
    "dgcnn_bn_lrelu_max_workspace_bytes": (_c_sz, [_c_int, _c_int, _c_int]),
    "dgcnn_bn_lrelu_max_fwd": (_c_int, [_c_vp, _c_int, _c_int, _c_int, ctypes.POINTER(BnParams), _c_int, _c_f, _c_f, _c_f,
                                        _c_vp, _c_vp, _c_vp, _c_vp, _c_sz, _c_vp]),
    "dgcnn_bn_lrelu_max_bwd": (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_int, _c_int, _c_int, _c_int, _c_f, _c_vp, _c_vp,
                                        _c_vp, _c_vp, _c_sz, _c_vp]),
    "dgcnn_rows_wgrad_workspace_bytes": (_c_sz, [_c_int, _c_int, _c_int]),
    "dgcnn_rows_wgrad": (_c_int, [_c_vp, _c_int, _c_int, _c_vp, _c_int, _c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_sz,
                                  _c_vp]),
    "dgcnn_rows_gemm": (_c_int, [_c_vp, _c_int, _c_vp, _c_int, _c_int, _c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_int, _c_vp,
                                 _c_int, _c_vp]),
    "dgcnn_rows_absmax": (_c_int, [_c_vp, _c_int, _c_int, _c_int, _c_vp, _c_vp]),
    "dgcnn_cls_fwd": (_c_int, [_c_vp, _c_vp, _c_int, _c_int, _c_int, _c_vp, _c_vp]),
    "dgcnn_cls_bwd_workspace_bytes": (_c_sz, [_c_int, _c_int, _c_int]),
    "dgcnn_cls_bwd": (_c_int, [_c_vp, _c_vp, _c_vp, _c_int, _c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_sz, _c_vp]),
    "dgcnn_embed_max_fwd_workspace_bytes": (_c_sz, [_c_int, _c_int, _c_int]),
    "dgcnn_embed_max_fwd": (_c_int, [_c_vp, _c_int, _c_vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_vp, _c_vp,
                                     ctypes.POINTER(BnParams), _c_int, _c_f, _c_f, _c_f, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_sz,
                                     _c_vp]),
    "dgcnn_embed_fwd_workspace_bytes": (_c_sz, [_c_int, _c_int, _c_int, _c_int]),
    "dgcnn_embed_fwd": (_c_int, [_c_vp, _c_int, _c_vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_vp, ctypes.POINTER(BnParams),
                                 _c_int, _c_f, _c_f, _c_f, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_sz, _c_vp]),
    "dgcnn_embed_bwd_workspace_bytes": (_c_sz, [_c_int, _c_int, _c_int, _c_int]),
    "dgcnn_embed_bwd": (_c_int, [_c_vp, _c_int, _c_vp, _c_vp, _c_vp, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_int, _c_int, _c_int,
                                 _c_int, _c_int, _c_f, _c_vp, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_sz, _c_vp]),
    "dgcnn_rows_stats_workspace_bytes": (_c_sz, [_c_int, _c_int]),
    "dgcnn_rows_stats": (_c_int, [_c_vp, _c_int, _c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_sz, _c_vp]),
    "dgcnn_embed_max_bwd_prepare": (_c_int, [_c_vp, _c_vp, _c_vp, _c_int, _c_int, _c_int, _c_int, _c_f, _c_vp, _c_vp, _c_vp,
                                             _c_vp, _c_vp, _c_vp]),
    "dgcnn_embed_max_bwd_scatter": (_c_int, [_c_vp, _c_vp, _c_vp, _c_int, _c_int, _c_int, _c_int, _c_vp, _c_vp]),
    "dgcnn_head_tail_saved_bytes": (_c_sz, [_c_ht_p]),
    "dgcnn_head_tail_fwd_workspace_bytes": (_c_sz, [_c_ht_p]),
    "dgcnn_head_tail_bwd_workspace_bytes": (_c_sz, [_c_ht_p]),
    "dgcnn_head_tail_fwd": (_c_int, [_c_ht_p, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_sz, _c_vp]),
    "dgcnn_head_tail_bwd": (_c_int, [_c_ht_p, _c_vp, _c_vp, _c_vp, _c_vp, ctypes.POINTER(HeadTailGrads), _c_vp, _c_sz, _c_vp]),
    "dgcnn_dropout_mask": (_c_int, [ctypes.c_uint64, _c_int, _c_int, _c_f, _c_vp, _c_vp]),
    "dgcnn_model_saved_bytes": (_c_sz, [_c_md_p]),
    "dgcnn_model_fwd_workspace_bytes": (_c_sz, [_c_md_p]),
    "dgcnn_model_bwd_workspace_bytes": (_c_sz, [_c_md_p]),
    "dgcnn_model_fwd": (_c_int, [_c_md_p, _c_vp, _c_i64, _c_i64, _c_i64, _c_vp, _c_i64, _c_i64, _c_i64, _c_vp, _c_vp, _c_sz, _c_vp]),
    "dgcnn_model_bwd": (_c_int, [_c_md_p, _c_vp, _c_i64, _c_i64, _c_i64, _c_vp, _c_i64, _c_i64, _c_i64, _c_vp,
                                 ctypes.POINTER(ModelGrads), _c_vp, _c_sz, _c_int, _c_vp]),
    "dgcnn_comm_rccl_version": (_c_int, []),
    "dgcnn_comm_unique_id": (_c_int, [_c_vp]),
    "dgcnn_comm_init": (_c_int, [ctypes.POINTER(_c_vp), _c_vp, _c_int, _c_int]),
    "dgcnn_comm_destroy": (_c_int, [_c_vp]),
    "dgcnn_comm_world": (_c_int, [_c_vp]),
    "dgcnn_comm_rank": (_c_int, [_c_vp]),
    "dgcnn_comm_allreduce_f32": (_c_int, [_c_vp, _c_vp, _c_sz, _c_int, _c_vp]),
    "dgcnn_comm_broadcast_f32": (_c_int, [_c_vp, _c_vp, _c_sz, _c_int, _c_vp]),
    "dgcnn_seg_loss_workspace_bytes": (_c_sz, [_c_int, _c_int]),
    "dgcnn_seg_loss": (_c_int, [_c_vp, _c_int, _c_int, _c_int, _c_i64, _c_i64, _c_i64, _c_vp, _c_vp, _c_f, _c_vp, _c_vp,
                                _c_vp, _c_vp, _c_sz, _c_vp]),
    "dgcnn_seg_loss_labels": (_c_int, [_c_vp, _c_int, _c_int, _c_int, _c_i64, _c_i64, _c_i64, _c_vp, _c_int, _c_vp, _c_f, _c_vp, _c_vp,
                                       _c_vp, _c_vp, _c_sz, _c_vp]),
    "dgcnn_seg_predict": (_c_int, [_c_vp, _c_int, _c_int, _c_int, _c_i64, _c_i64, _c_i64, _c_f, _c_int, _c_vp, _c_vp, _c_vp]),
    "dgcnn_adam_step": (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_sz, _c_f, _c_f, _c_f, _c_f, _c_f, _c_int, _c_vp]),
    "dgcnn_cast_f64_f32": (_c_int, [_c_vp, _c_sz, _c_vp]),
    "dgcnn_profile_enable": (_c_int, [_c_int]),
    "dgcnn_profile_resume": (_c_int, []),
    "dgcnn_profile_regions": (_c_int, []),
    "dgcnn_profile_region_name": (ctypes.c_char_p, [_c_int]),
    "dgcnn_profile_mean_ms": (_c_f, [_c_int]),
    "dgcnn_profile_samples": (_c_int, [_c_int]),
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
