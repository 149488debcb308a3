"""GPU parity of the fused kNN kernel (through the C ABI) against oracle/knn_exact.c.

Gate (gate 1): neighbour indices BIT-EXACT against the oracle's specified
arithmetic (fp32 FMA chain in channel order, lower index wins ties, nearest first)."""
import numpy as np
import pytest
import torch

from oracle import knn_exact as ke

pytestmark = pytest.mark.gpu


def _run(x_bcn, k, device):
    from ahn3_dgcnn_pytorch_amd import ops
    x = torch.from_numpy(x_bcn).to(device)
    x_pm = ops.to_point_major(x)
    idx = ops.knn_pm(x_pm, k)
    torch.cuda.synchronize()
    return idx.cpu().numpy()


@pytest.mark.parametrize("B,C,N,k", [
    (2, 4, 256, 8), (2, 64, 256, 30), (1, 4, 1000, 30), (3, 3, 130, 20), (2, 9, 777, 40),
    (1, 64, 2000, 64), (2, 128, 300, 30), (1, 6, 64, 64), (1, 5, 33, 33), (2, 64, 1, 1),
    (2, 16, 1500, 50), (1, 32, 4100, 64), (2, 64, 1100, 41),   # 40 < k <= 64 with 128 threshold classes (fp16 path)
    (1, 4, 3000, 64), (2, 8, 900, 50), (2, 3, 260, 41),        # ... and the fp32 path for small C
])
def test_knn_bit_exact_random(B, C, N, k, device):
    rng = np.random.default_rng(1234 + B * 7 + C * 13 + N)
    x = rng.standard_normal((B, C, N)).astype(np.float32)
    want = ke.knn_exact(x, k)
    got = _run(x, k, device)
    assert got.dtype == np.int32 and got.shape == (B, N, k)
    np.testing.assert_array_equal(got, want)


def test_knn_block_like_coordinates(device):
    # block-local x,y in [0,30), z, agl in [0,20): the value range of the real tiles
    rng = np.random.default_rng(7)
    x = rng.random((2, 4, 3000)).astype(np.float32) * np.array([30, 30, 20, 20], np.float32)[None, :, None]
    np.testing.assert_array_equal(_run(x, 30, device), ke.knn_exact(x, 30))


def test_knn_duplicate_points_lower_index_wins(device):
    # second half of the cloud is a copy of the first: every distance is tied at least twice.
    rng = np.random.default_rng(3)
    half = rng.standard_normal((1, 8, 250)).astype(np.float32)
    x = np.concatenate([half, half], axis=2)
    want = ke.knn_exact(x, 20)
    got = _run(x, 20, device)
    np.testing.assert_array_equal(got, want)
    # the fixed tie-break: point i and its copy i+250 share a list that starts (i, i+250)
    assert (got[0, :250, 0] == np.arange(250)).all() and (got[0, :250, 1] == np.arange(250) + 250).all()
    assert (got[0, 250:, 0] == np.arange(250)).all()


def test_knn_all_points_identical(device):
    x = np.ones((1, 4, 200), np.float32)
    got = _run(x, 30, device)
    np.testing.assert_array_equal(got, np.broadcast_to(np.arange(30, dtype=np.int32), (1, 200, 30)))


def test_knn_integer_grid_many_ties(device):
    # small integer coordinates: all arithmetic exact, massive ties -> pure tie-break test
    rng = np.random.default_rng(11)
    x = rng.integers(0, 4, size=(2, 3, 500)).astype(np.float32)
    np.testing.assert_array_equal(_run(x, 30, device), ke.knn_exact(x, 30))


def test_knn_full_size_bit_exact(device):
    # BASELINE.json configs[1] size: B=8, N=7000, k=30, C=64 (the oracle takes ~1-2 s on 8 cores)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((8, 64, 7000)).astype(np.float32)
    got = _run(x, 30, device)
    want = ke.knn_exact(x, 30)
    np.testing.assert_array_equal(got, want)
    assert (got[:, :, 0] == np.arange(7000)[None]).all()  # self is neighbour 0 on tie-free data


@pytest.mark.parametrize("C", [4, 64])
def test_knn_stress_size_bit_exact(C, device):
    # BASELINE.json configs[4]: N=32768, k=64 (one batch element; C=4 is block 1's search in the raw coordinates --
    # the fp32 path -- and C=64 the feature-space search of blocks 2 and 3 -- the fp16 filter path)
    rng = np.random.default_rng(5 + C)
    if C == 4:
        x = rng.random((1, 4, 32768)).astype(np.float32) * np.array([30, 30, 20, 20], np.float32)[None, :, None]
    else:
        x = rng.standard_normal((1, C, 32768)).astype(np.float32)
    np.testing.assert_array_equal(_run(x, 64, device), ke.knn_exact(x, 64))


# ---- the fp16 filter + exact refine path (C >= 16, k <= 40, N >= 256) must never change the result -------------
def _feature_cases():
    rng = np.random.default_rng(99)
    B, C, N = 2, 64, 900
    base = rng.standard_normal((B, C, N)).astype(np.float32)
    cases = {
        "offset_1000": base + 1000.0,                       # |mean| >> spread: cancellation in the oracle's fp32 pd
        "tiny_1e-4": base * 1e-4,                           # fp16 subnormal range after centring
        "huge_1e4": base * 1e4,                             # centred norms near the fp16 limit
        "overflow_1e6": base * 1e6,                         # does not fit fp16 -> exact fp32 kernel fallback
        "one_wide_channel": base * np.concatenate([[3000.0], np.ones(C - 1)]).astype(np.float32)[None, :, None],
        "post_activation_like": np.maximum(base, 0.2 * base) + 3.0,  # LeakyReLU outputs with a common shift
        "duplicates": np.concatenate([base[:, :, :450], base[:, :, :450]], axis=2),
        "clustered": (base * 0.01 + rng.integers(0, 4, size=(B, 1, N)).astype(np.float32) * 5.0),
        "constant": np.full((B, C, N), 2.5, np.float32),
        # few distinct coordinate values: the distances take few distinct values, so many exact ties sit on the k-th
        # boundary and the survivor lists are long (rows are compacted while later, lower-index ties still arrive)
        "quantized_ties": np.round(base[:, :, :] * 0.6).astype(np.float32) * 0.5 + rng.integers(0, 2, size=(B, 1, N)).astype(np.float32),
        "quantized_ties_sparse": (np.abs(base) > 1.8).astype(np.float32),
    }
    mixed = base.copy()
    mixed[0] *= 1e-3
    mixed[1] = mixed[1] * 50.0 - 7.0
    cases["mixed_scales_per_batch_element"] = mixed
    return cases


@pytest.mark.parametrize("name", sorted(_feature_cases().keys()))
def test_knn_f16_filter_path_is_exact(name, device):
    x = _feature_cases()[name]
    for k in (8, 30, 40, 64):
        np.testing.assert_array_equal(_run(x, k, device), ke.knn_exact(x, k), err_msg="%s k=%d" % (name, k))


def test_knn_f16_path_c16_c32_c128(device):
    rng = np.random.default_rng(5)
    for C in (16, 24, 32, 100, 128):
        x = (rng.standard_normal((2, C, 700)) * rng.uniform(0.1, 10.0, size=(1, C, 1)) + 2.0).astype(np.float32)
        np.testing.assert_array_equal(_run(x, 20, device), ke.knn_exact(x, 20), err_msg="C=%d" % C)


def test_knn_small_c_filter_path_is_exact(device):
    # C <= 8 (raw coordinates of the first block): C <= 4 through the fp16 streams on two-piece operands, C = 8 through the fp32
    # pre-pass + fp32 filter, both followed by the exact refine; offsets, scales, duplicates
    rng = np.random.default_rng(21)
    for C in (1, 2, 3, 4, 8):
        base = rng.standard_normal((2, C, 1200)).astype(np.float32)
        cases = {
            "plain": base,
            "offset_1000": base + 1000.0,
            "utm_like": base * 10.0 + np.array([5.0e5, 4.0e6, 50.0, 3.0, 0, 0, 0, 0], np.float32)[None, :C, None],
            "tiny": base * 1e-5,
            # centred coordinates below 2^-3: the low fp16 piece of the two-piece operands is subnormal or zero there (an
            # absolute rounding of 2^-25 per element, covered by the linear term of knn_f16_eps -- ADVICE round 3)
            "scale_1e-1": base * 1e-1,
            "scale_1e-2": base * 1e-2,
            "scale_1e-3": base * 1e-3,
            "duplicates": np.concatenate([base[:, :, :600], base[:, :, :600]], axis=2),
            "clustered": base * 0.01 + rng.integers(0, 3, size=(2, 1, 1200)).astype(np.float32) * 7.0,
        }
        for name, x in cases.items():
            x = np.ascontiguousarray(x, dtype=np.float32)
            for k in (8, 30, 40, 64):
                np.testing.assert_array_equal(_run(x, k, device), ke.knn_exact(x, k), err_msg="C=%d %s k=%d" % (C, name, k))


@pytest.mark.parametrize("C", [4, 64])
def test_knn_large_n_bit_exact(C, device):
    # one big block: N = 20 000 (157 row groups, both candidate halves, list segments)
    rng = np.random.default_rng(3 + C)
    x = rng.standard_normal((1, C, 20000)).astype(np.float32)
    np.testing.assert_array_equal(_run(x, 30, device), ke.knn_exact(x, 30))


def _random_case(rng):
    C = int(rng.choice([3, 4, 5, 8, 9, 16, 24, 32, 64, 100, 128]))
    N = int(rng.choice([257, 300, 512, 777, 1000, 1500, 2048]))
    B = int(rng.choice([1, 2, 3]))
    k = min(int(rng.choice([1, 8, 20, 30, 32, 33, 40, 41, 64])), N)
    x = rng.standard_normal((B, C, N)).astype(np.float32)
    mode = str(rng.choice(["plain", "scale", "offset", "quant", "cluster", "sparse", "dups", "chanscale"]))
    if mode == "scale":
        x *= np.float32(10.0 ** rng.uniform(-4, 4))
    elif mode == "offset":
        x += np.float32(10.0 ** rng.uniform(0, 4))
    elif mode == "quant":  # exact ties
        x = np.round(x * rng.uniform(0.3, 2.0)).astype(np.float32) * np.float32(rng.choice([0.25, 0.5, 1.0]))
    elif mode == "cluster":
        x = x * np.float32(10.0 ** rng.uniform(-3, -1)) + rng.integers(0, rng.integers(2, 6), size=(B, 1, N)).astype(np.float32) * np.float32(rng.uniform(1, 10))
    elif mode == "sparse":
        x = (np.abs(x) > rng.uniform(1.0, 2.2)).astype(np.float32) * np.float32(rng.uniform(0.5, 3))
    elif mode == "dups":
        x[:, :, N // 2:] = x[:, :, : N - N // 2]
    elif mode == "chanscale":
        x *= (10.0 ** rng.uniform(-2, 2, size=(1, C, 1))).astype(np.float32)
    return np.ascontiguousarray(x, dtype=np.float32), k, "B=%d C=%d N=%d k=%d %s" % (B, C, N, k, mode)


def test_knn_randomised_shapes_scales_and_ties(device):
    # 150 random configurations (every kernel path: fp32 / fp16 filters, both refine capacities, the in-loop kernel for
    # k > 40) against the oracle; scripts/fuzz_knn.py runs the same generator for thousands of cases
    rng = np.random.default_rng(2026)
    for it in range(150):
        x, k, what = _random_case(rng)
        np.testing.assert_array_equal(_run(x, k, device), ke.knn_exact(x, k), err_msg="case %d: %s" % (it, what))


@pytest.mark.parametrize("B,C,N,k,mode", [(6, 4, 7000, 30, "plain"), (9, 64, 7000, 20, "mixed"), (12, 16, 4096, 8, "dups"),
                                          (24, 32, 2048, 32, "quant"), (8, 64, 7000, 30, "quant"), (7, 3, 8192, 16, "offset")])
def test_knn_balanced_filter_split_is_exact(B, C, N, k, mode, device):
    """Shapes at which the filter's workgroups take BALANCED runs of (row group, candidate chunk) units (knn.hip,
    knn_filter_balance: the uniform split would not fill the chip's workgroup slots a whole number of times): a run may cross
    a row-group boundary, a row group's list has three or four segments of unequal length.  Ties, duplicates (long lists), a
    batch element outside the fp16 range (flagged: the exact pass inside the refine) -- indices bit-exact as everywhere."""
    rng = np.random.default_rng(100 * B + C)
    x = rng.standard_normal((B, C, N)).astype(np.float32)
    if mode == "mixed":
        x[1] *= 1e-3
        x[2] = x[2] * 50.0 - 7.0
        x[3] *= 3.0e4          # centred features beyond the fp16 range: the whole batch element goes to the exact pass
    elif mode == "dups":
        x[:, :, N // 2:] = x[:, :, : N - N // 2]
    elif mode == "quant":
        x = np.round(x * 1.5).astype(np.float32) * np.float32(0.5)
    elif mode == "offset":
        x = x * 10.0 + np.array([5.0e5, 4.0e6, 50.0], np.float32)[None, :C, None]
    x = np.ascontiguousarray(x, dtype=np.float32)
    np.testing.assert_array_equal(_run(x, k, device), ke.knn_exact(x, k))


def test_knn_k_out_of_range(device):
    from ahn3_dgcnn_pytorch_amd import ops
    x = torch.zeros((1, 10, 4), device=device)
    with pytest.raises(RuntimeError, match="out of range"):
        ops.knn_pm(x, 11)
