"""CPU tests of the host side: the C-ABI library loads and exports every declared symbol (no
compute calls without a GPU), the module mirrors the reference's surface, and the compute entry
points refuse to run without the HIP device instead of falling back."""
import ctypes
import json
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def test_library_exports_every_declared_symbol():
    from ahn3_dgcnn_pytorch_amd import _lib
    header = open(os.path.join(ROOT, "include", "dgcnn_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(dgcnn_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name)
    assert lib.dgcnn_abi_version() >= 4


def test_ctypes_struct_layout_matches_c():
    from ahn3_dgcnn_pytorch_amd import _lib
    # natural alignment on x86-64: 6 int32 + 3 float = 36 -> pad to 40, then pointers
    assert ctypes.sizeof(_lib.BnParams) == 40  # four parameter pointers + num_batches_tracked (ABI version 4)
    assert _lib.EdgeConvDesc.w1.offset == 40
    assert ctypes.sizeof(_lib.EdgeConvDesc) == 40 + 8 + 40 + 8 + 40
    assert ctypes.sizeof(_lib.EdgeConvGrads) == 56


def test_workspace_queries_need_no_gpu():
    from ahn3_dgcnn_pytorch_amd import _lib
    lib = _lib.load()
    assert lib.dgcnn_knn_workspace_bytes(8, 64, 7000, 30) >= 8 * 7040 * 64 * 4
    d = _lib.EdgeConvDesc()
    d.B, d.N, d.C, d.k, d.n_stage, d.training = 8, 7000, 64, 30, 2, 1
    assert lib.dgcnn_edgeconv_saved_bytes(d) >= 3 * 56000 * 64 * 4
    assert lib.dgcnn_edgeconv_bwd_workspace_bytes(d) >= 56000 * 30 * 64 * 4


def test_backward_limit_on_points_is_reported_by_the_workspace_queries():
    """every backward builds the reverse neighbour lists (N ints of LDS per wave): N <= 40000 is a limit of the
    backward entry points, reported where the caller sizes its workspace -- not after the workspace has been carved"""
    from ahn3_dgcnn_pytorch_amd import _lib
    lib = _lib.load()
    d = _lib.EdgeConvDesc()
    d.B, d.N, d.C, d.k, d.n_stage, d.training = 1, 40001, 64, 30, 1, 0
    assert lib.dgcnn_edgeconv_fwd_workspace_bytes(d) > 0          # the forward has no such limit
    assert lib.dgcnn_edgeconv_bwd_workspace_bytes(d) == 0
    assert b"40000" in lib.dgcnn_last_error()
    assert lib.dgcnn_graph_feature_bwd_workspace_bytes(1, 40001, 4, 30) == 0
    assert b"40000" in lib.dgcnn_last_error()
    d.N = 40000
    assert lib.dgcnn_edgeconv_bwd_workspace_bytes(d) > 0
    assert lib.dgcnn_graph_feature_bwd_workspace_bytes(1, 40000, 4, 30) > 0
    buf = ctypes.create_string_buffer(64)
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.dgcnn_graph_feature_bwd(p, p, 1, 40001, 4, 30, p, p, 1 << 30, None) != 0
    assert b"40000" in lib.dgcnn_last_error()


def test_argument_errors_are_reported_not_crashed():
    from ahn3_dgcnn_pytorch_amd import _lib
    lib = _lib.load()
    rc = lib.dgcnn_knn(None, 1, 10, 4, 3, None, None, 0, None)
    assert rc != 0 and b"null" in lib.dgcnn_last_error()
    buf = ctypes.create_string_buffer(64)
    p = ctypes.cast(buf, ctypes.c_void_p)
    rc = lib.dgcnn_knn(p, 1, 10, 4, 11, p, p, 1 << 20, None)
    assert rc != 0 and b"out of range" in lib.dgcnn_last_error()


def test_module_surface_matches_reference():
    from ahn3_dgcnn_pytorch_amd import DGCNN
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    with open(os.path.join(GOLD, "state_dict.json")) as f:
        spec = json.load(f)
    m = DGCNN(5, 4, 30, dropout=0.5, emb_dims=1024, cuda=False)
    sd = m.state_dict()
    assert list(sd.keys()) == list(spec["entries"].keys()) and len(sd) == 89
    for k, (shape, dtype) in spec["entries"].items():
        assert list(sd[k].shape) == shape and str(sd[k].dtype) == "torch." + dtype
    assert sum(p.numel() for p in m.parameters()) == spec["n_parameters"]
    assert len(list(m.parameters())) == spec["n_parameter_tensors"]
    for attr, val in spec["attributes"].items():
        assert getattr(m, attr) == val
    assert [n for n, _ in m.named_children()] == ["bn%d" % i for i in range(1, 9)] + \
        ["conv%d" % i for i in range(1, 9)] + ["dp1", "conv9"]
    assert DGCNN(5, 4, 30, ropout=0.25).dp1.p == 0.25  # train.py:103's keyword


def test_no_cpu_fallback():
    from ahn3_dgcnn_pytorch_amd import DGCNN, _lib, get_graph_feature, knn, ops
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    x = torch.rand(1, 4, 64)
    with pytest.raises(_lib.HipExtensionError):
        DGCNN(5, 4, 8)(x)
    with pytest.raises(_lib.HipExtensionError):
        knn(x, 4)
    with pytest.raises(_lib.HipExtensionError):
        get_graph_feature(x, k=4)
    with pytest.raises(_lib.HipExtensionError):
        ops.knn_pm(x.transpose(1, 2).contiguous(), 4)
    with pytest.raises(_lib.HipExtensionError):
        DGCNN(5, 4, 8, cuda=True)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "ahn3-dgcnn.pytorch_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), fn
                assert "torch_ref" not in src and "knn_exact(" not in src, fn


@pytest.mark.timeout(180)
def test_bench_never_reports_fewer_ranks_than_asked():
    """bench.py --gpus N: a launcher environment with another WORLD_SIZE is refused (non-zero, no JSON line); without
    one, N > 1 starts N ranks as a child launcher, and with no HIP device every rank fails loudly -- there is no
    1-rank or CPU fallback."""
    import subprocess
    import sys
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert p.returncode == 2 and "refusing" in p.stderr and "{" not in p.stdout
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=170)
    assert p.returncode != 0 and "needs a HIP device" in p.stderr and "{" not in p.stdout


def test_flat_adam_is_a_torch_optimizer_class():
    from ahn3_dgcnn_pytorch_amd.optim import FlatAdam
    assert issubclass(FlatAdam, torch.optim.Optimizer)
    if not torch.cuda.is_available():
        from ahn3_dgcnn_pytorch_amd import _lib
        with pytest.raises(_lib.HipExtensionError):     # no CPU fallback
            FlatAdam([torch.nn.Parameter(torch.zeros(4))])


def test_rccl_binding_behind_the_c_abi_loads_without_a_gpu():
    """dgcnn_comm_*: RCCL is bound at run time (the copy the process already holds); making an id needs no device"""
    from ahn3_dgcnn_pytorch_amd import _lib
    lib = _lib.load()
    assert lib.dgcnn_comm_rccl_version() >= 20000
    buf = ctypes.create_string_buffer(128)
    assert lib.dgcnn_comm_unique_id(buf) == 0
    assert any(b != 0 for b in buf.raw)
    assert lib.dgcnn_comm_allreduce_f32(None, None, 0, 0, None) != 0 and b"null" in lib.dgcnn_last_error()


def test_model_structs_match_the_header(tmp_path):
    """dgcnn_model_desc / dgcnn_model_grads (arrays of pointers, nested BatchNorm structs): sizes and member offsets as the
    C compiler lays out include/dgcnn_hip.h against the ctypes mirrors; and the shape limits are reported by the queries"""
    import subprocess
    from ahn3_dgcnn_pytorch_amd import _lib
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "dgcnn_hip.h"\nint main(void) {\n'
                   '  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(dgcnn_model_desc), offsetof(dgcnn_model_desc, slope), '
                   'offsetof(dgcnn_model_desc, seed), offsetof(dgcnn_model_desc, w), offsetof(dgcnn_model_desc, bn), '
                   'offsetof(dgcnn_model_desc, eps), offsetof(dgcnn_model_desc, momentum), offsetof(dgcnn_model_desc, idx), '
                   'sizeof(dgcnn_model_grads));\n  return 0;\n}\n')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(v) for v in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    D = _lib.ModelDesc
    want = [ctypes.sizeof(D), D.slope.offset, D.seed.offset, D.w.offset, D.bn.offset, D.eps.offset, D.momentum.offset, D.idx.offset,
            ctypes.sizeof(_lib.ModelGrads)]
    assert got == want
    lib = _lib.load()
    d = _lib.ModelDesc()
    d.B, d.N, d.F, d.k, d.num_classes, d.emb, d.c7, d.c8, d.training = 8, 7000, 4, 30, 5, 1024, 512, 256, 1
    assert lib.dgcnn_model_saved_bytes(d) > 56000 * 192 * 4 and lib.dgcnn_model_fwd_workspace_bytes(d) > 0
    assert lib.dgcnn_model_bwd_workspace_bytes(d) > 56000 * 30 * 64 * 4
    d.k = 65
    assert lib.dgcnn_model_saved_bytes(d) == 0 and b"k=65" in lib.dgcnn_last_error()
    d.k, d.N = 30, 40001
    assert lib.dgcnn_model_fwd_workspace_bytes(d) > 0 and lib.dgcnn_model_bwd_workspace_bytes(d) == 0
    assert b"40000" in lib.dgcnn_last_error()
