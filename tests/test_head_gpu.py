"""Head products on the fp16 matrix cores (dgcnn_rows_gemm: conv6 .. conv8 forward and input gradients, SURVEY.md 8f row
N1) against a float64 product: the split-operand evaluation has to stay at fp32's own error level for every operand
scale."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _check(x, w, add=None, rows_per_add=0, tol=2e-6):
    from ahn3_dgcnn_pytorch_amd import ops
    dev = torch.device("cuda:0")
    got = ops.rows_gemm(x.to(dev), w.to(dev), add.to(dev) if add is not None else None, rows_per_add).cpu().double()
    want = x.double() @ w.double().t()
    scale = x.double().abs() @ w.double().abs().t()     # fp32 accumulation error scales with the sum of magnitudes
    if add is not None:
        P = x.shape[0]
        rows = torch.arange(P) // rows_per_add
        want = want + add.double()[rows]
        scale = scale + add.double().abs()[rows]
    assert got.shape == want.shape
    rel = ((got - want).abs() / scale.clamp_min(1e-300)).max().item()
    assert rel <= tol, rel
    return got


@pytest.mark.parametrize("P,K,N", [(1000, 192, 1024), (7000, 192, 512), (513, 512, 256), (128, 256, 512), (77, 512, 192),
                                   (300, 192, 192), (1, 32, 5), (4097, 64, 130)])
def test_rows_gemm_matches_float64(P, K, N):
    g = torch.Generator().manual_seed(P + K + N)
    _check(torch.randn(P, K, generator=g) + 0.3, torch.randn(N, K, generator=g) / K ** 0.5)


@pytest.mark.parametrize("sx,sw", [(1e-8, 1.0), (1e6, 1e-3), (1.0, 1e8), (3e-5, 7e-6), (1e15, 1e-15)])
def test_rows_gemm_operand_scales(sx, sw):
    # gradients of a mean loss are ~1e-8, weights can be anything: the exact power-of-two scales keep the relative error
    g = torch.Generator().manual_seed(3)
    _check(torch.randn(700, 256, generator=g) * sx, torch.randn(200, 256, generator=g) * sw)


def test_rows_gemm_wide_dynamic_range_within_an_operand():
    # one huge entry next to ordinary ones: the ordinary ones keep an absolute error relative to the LARGEST entry
    g = torch.Generator().manual_seed(4)
    x = torch.randn(500, 192, generator=g)
    x[17, 5] = 3.0e4
    w = torch.randn(256, 192, generator=g)
    _check(x, w, tol=2e-6)


def test_rows_gemm_strided_weight_and_per_batch_add():
    # conv7 of the head: W7 = [W7g | W7c] (model.py:134), the 192-wide slice as a strided view, the pooled feature's
    # share added per batch element
    g = torch.Generator().manual_seed(5)
    B, N = 3, 333
    cat = torch.randn(B * N, 192, generator=g)
    w7 = torch.randn(512, 1216, generator=g) / 30.0
    gterm = torch.randn(B, 512, generator=g)
    _check(cat, w7[:, 1024:], gterm, N)


def test_rows_gemm_zero_operand_and_determinism():
    from ahn3_dgcnn_pytorch_amd import ops
    dev = torch.device("cuda:0")
    x = torch.zeros(300, 64)
    w = torch.randn(96, 64, generator=torch.Generator().manual_seed(6))
    assert float(ops.rows_gemm(x.to(dev), w.to(dev)).abs().max()) == 0.0
    x = torch.randn(5000, 192, generator=torch.Generator().manual_seed(7)).to(dev)
    wd = torch.randn(1024, 192, generator=torch.Generator().manual_seed(8)).to(dev)
    assert torch.equal(ops.rows_gemm(x, wd), ops.rows_gemm(x, wd))


def test_linear_rows_autograd_matches_torch():
    from ahn3_dgcnn_pytorch_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(9)
    B, N = 2, 400
    x = torch.randn(B * N, 512, generator=g)
    w = torch.randn(256, 512, generator=g) / 20.0
    add = torch.randn(B, 256, generator=g)
    R = torch.randn(B * N, 256, generator=g)
    xr, wr, ar = (t.clone().double().requires_grad_(True) for t in (x, w, add))
    yr = xr @ wr.t() + ar.repeat_interleave(N, dim=0)
    (yr * R.double()).sum().backward()
    xd, wd, ad = (t.to(dev).requires_grad_(True) for t in (x, w, add))
    yd = ops.linear_rows(xd, wd, ad, N)
    (yd * R.to(dev)).sum().backward()
    for got, want in ((yd, yr), (xd.grad, xr.grad), (wd.grad, wr.grad), (ad.grad, ar.grad)):
        err = (got.detach().cpu().double() - want.detach()).abs().max().item()
        assert err <= 1e-5 * want.detach().abs().max().item()


@pytest.mark.parametrize("P,K,NC", [(1000, 256, 5), (4097, 256, 2), (333, 64, 8), (7, 512, 3), (20000, 256, 5)])
def test_classifier_rows_matches_float64(P, K, NC):
    # conv9 (model.py:141, :191): the streaming forward and the one-pass backward against float64
    from ahn3_dgcnn_pytorch_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(P + K + NC)
    x = torch.randn(P, K, generator=g) + 0.2
    w = torch.randn(NC, K, generator=g) / K ** 0.5
    R = torch.randn(P, NC, generator=g)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = xr @ wr.t()
    (yr * R.double()).sum().backward()
    xd, wd = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True)
    yd = ops.classifier_rows(xd, wd)
    (yd * R.to(dev)).sum().backward()
    for got, want, scale in ((yd, yr, (x.double().abs() @ w.double().abs().t())),
                             (xd.grad, xr.grad, (R.double().abs() @ w.double().abs())),
                             (wd.grad, wr.grad, (R.double().abs().t() @ x.double().abs()))):
        rel = ((got.detach().cpu().double() - want.detach()).abs() / scale.clamp_min(1e-300)).max().item()
        assert rel <= 2e-6, rel
    xd2, wd2 = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True)
    (ops.classifier_rows(xd2, wd2) * R.to(dev)).sum().backward()
    assert torch.equal(xd2.grad, xd.grad) and torch.equal(wd2.grad, wd.grad)     # fixed order


def test_classifier_rows_many_classes_uses_the_library_product():
    from ahn3_dgcnn_pytorch_amd import ops
    dev = torch.device("cuda:0")
    x, w = torch.randn(100, 64).to(dev), torch.randn(13, 64).to(dev)
    torch.testing.assert_close(ops.classifier_rows(x, w), x @ w.t(), rtol=1e-5, atol=1e-5)


def test_rows_gemm_rows_far_below_the_largest_keep_the_absolute_bound():
    # ONE power-of-two scale per tensor (include/dgcnn_hip.h): rows 1e-8 of the others are represented to about 2^-39 of
    # the tensor's largest entry, absolute -- not to fp32's relative accuracy (dgcnn_head_tail_* scales per 128-row tile)
    from ahn3_dgcnn_pytorch_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(8)
    x = torch.randn(600, 192, generator=g)
    x[100:200] *= 1e-8
    w = torch.randn(256, 192, generator=g)
    got = ops.rows_gemm(x.to(dev), w.to(dev)).cpu().double()
    want = x.double() @ w.double().t()
    big = torch.ones(600, dtype=torch.bool)
    big[100:200] = False
    bound = 2.0 ** -36 * x.abs().max().double() * w.double().abs().sum(1)      # per output column: the small rows' absolute bound
    assert ((got - want).abs()[~big] <= bound.unsqueeze(0)).all()
    rel = (got - want).abs()[big] / (x.double().abs() @ w.double().abs().t())[big]
    assert rel.max().item() <= 2e-6
