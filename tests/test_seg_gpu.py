"""Parity of the steps either side of DGCNN.forward with their CPU restatement
(oracle/seg_ref.py): loss within 1e-6, d loss / d logits within 1e-5 of its scale 1 / (B N), predictions exact,
Adam within 1e-6 after ten steps, float64 -> float32 exact."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _case(B, C, N, seed, scale=3.0):
    g = torch.Generator().manual_seed(seed)
    z = torch.randn(B, C, N, generator=g) * scale
    lab = torch.randint(0, C, (B, N), generator=g)
    msk = (torch.rand(B, N, generator=g) < 0.3)
    return z, lab, msk


@pytest.mark.parametrize("B,C,N", [(2, 5, 1000), (1, 5, 257), (3, 8, 64), (2, 13, 300), (1, 32, 130), (2, 2, 77)])
@pytest.mark.parametrize("masked", [False, True])
@pytest.mark.parametrize("eps", [0.2, 0.0])
def test_seg_loss_matches_reference_ops(B, C, N, masked, eps):
    from ahn3_dgcnn_pytorch_amd import ops
    from oracle import seg_ref
    z, lab, msk = _case(B, C, N, 11 * B + C + N)
    zr = z.clone().requires_grad_(True)
    loss_r, pred_r = seg_ref.step_loss(zr, lab, msk if masked else None, eps=eps)
    loss_r.backward()
    zd = z.to(_dev()).requires_grad_(True)
    loss, pred = ops.seg_loss(zd, lab.double().to(_dev()), msk.to(_dev()) if masked else None, smoothing=eps)
    (loss * 1.0).backward()
    assert abs(loss.item() - loss_r.item()) <= 1e-6 * max(1.0, abs(loss_r.item()))
    assert torch.equal(pred.cpu(), pred_r)
    scale = 1.0 / (B * N)
    assert (zd.grad.cpu() - zr.grad).abs().max().item() <= 1e-5 * scale


def test_seg_loss_reads_point_major_logits_in_place():
    from ahn3_dgcnn_pytorch_amd import ops
    from oracle import seg_ref
    z, lab, _ = _case(2, 5, 500, 3)
    z_pm = z.permute(0, 2, 1).contiguous().to(_dev())          # [B, N, C] memory
    view = z_pm.permute(0, 2, 1).requires_grad_(True)           # [B, C, N] view of it
    loss, pred = ops.seg_loss(view, lab.to(_dev()))
    loss.backward()
    zr = z.clone().requires_grad_(True)
    loss_r, pred_r = seg_ref.step_loss(zr, lab)
    loss_r.backward()
    assert abs(loss.item() - loss_r.item()) <= 1e-6
    assert torch.equal(pred.cpu(), pred_r)
    assert (view.grad.cpu() - zr.grad).abs().max().item() <= 1e-5 / 1000


def test_seg_loss_scales_with_upstream_gradient():
    from ahn3_dgcnn_pytorch_amd import ops
    z, lab, _ = _case(1, 5, 300, 5)
    a = z.to(_dev()).requires_grad_(True)
    b = z.to(_dev()).requires_grad_(True)
    ops.seg_loss(a, lab.to(_dev()))[0].backward()
    (ops.seg_loss(b, lab.to(_dev()))[0] * 4.0).backward()
    assert torch.allclose(b.grad, 4.0 * a.grad, rtol=1e-6, atol=0)


@pytest.mark.parametrize("C,conf", [(5, 0.0), (5, 0.6), (13, 0.3)])
def test_seg_predict_matches_reference_ops(C, conf):
    from ahn3_dgcnn_pytorch_amd import ops
    from oracle import seg_ref
    z, _, _ = _case(2, C, 700, 21 + C, scale=1.5)
    pred_r, val_r = seg_ref.predict_step(z, conf)
    pred, prob = ops.seg_predict(z.to(_dev()), conf, unclassified=seg_ref.UNCLASSIFIED)
    # a probability within rounding of the threshold may fall either side: exclude those points
    safe = (val_r - conf).abs() > 1e-6
    assert torch.equal(pred.cpu()[safe], pred_r[safe])
    assert (prob.cpu() - val_r).abs().max().item() <= 1e-6
    if conf > 0:
        assert (pred.cpu() == seg_ref.UNCLASSIFIED).any()


def test_label_out_of_range_poisons_the_loss():
    from ahn3_dgcnn_pytorch_amd import ops
    z, lab, _ = _case(1, 5, 100, 9)
    lab[0, 17] = 5
    loss, _ = ops.seg_loss(z.to(_dev()), lab.to(_dev()))
    assert torch.isnan(loss).item()


def test_bad_arguments_raise():
    from ahn3_dgcnn_pytorch_amd import _lib, ops
    z = torch.randn(1, 40, 10, device=_dev())
    with pytest.raises(_lib.HipExtensionError):
        ops.seg_predict(z)                                   # more than 32 classes
    with pytest.raises(_lib.HipExtensionError):
        ops.seg_loss(torch.randn(1, 5, 10), torch.zeros(1, 10))   # CPU tensor: no fallback


@pytest.mark.parametrize("set_to_none", [True, False])
def test_flat_adam_matches_torch_adam(set_to_none):
    from ahn3_dgcnn_pytorch_amd.optim import FlatAdam
    g = torch.Generator().manual_seed(0)
    shapes = [(64, 8, 1, 1), (64,), (64,), (5, 256, 1), (1000, 3)]
    ref = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes]
    mine = [torch.nn.Parameter(p.detach().clone().to(_dev())) for p in ref]
    opt_r = torch.optim.Adam(ref, lr=1e-3, weight_decay=1e-4)          # train.py:118
    opt = FlatAdam(mine, lr=1e-3, weight_decay=1e-4)
    for step in range(10):
        opt_r.zero_grad()
        opt.zero_grad(set_to_none=set_to_none)
        for pr, pm in zip(ref, mine):
            gr = torch.randn(pr.shape, generator=g) * (10.0 ** (step % 3 - 1))
            pr.grad = gr.clone()
            if set_to_none:
                assert pm.grad is None
                pm.grad = gr.to(_dev())            # what backward does when .grad is None
            else:
                pm.grad.add_(gr.to(_dev()))        # ... and when it is a (zeroed) view of the flat buffer
        opt_r.step()
        opt.step()
    for pr, pm in zip(ref, mine):
        assert (pm.detach().cpu() - pr.detach()).abs().max().item() <= 1e-6
    assert all(pm.data_ptr() >= opt.flat.data_ptr() for pm in mine)     # the parameters are views of the flat buffer


def _adam_pair(shapes, seed=0):
    from ahn3_dgcnn_pytorch_amd.optim import FlatAdam
    g = torch.Generator().manual_seed(seed)
    ref = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes]
    mine = [torch.nn.Parameter(p.detach().clone().to(_dev())) for p in ref]
    return g, ref, mine, torch.optim.Adam(ref, lr=1e-3, weight_decay=1e-4), FlatAdam(mine, lr=1e-3, weight_decay=1e-4)


def _feed(g, ref, mine, skip=()):
    for i, (pr, pm) in enumerate(zip(ref, mine)):
        if i in skip:
            pr.grad = pm.grad = None
            continue
        gr = torch.randn(pr.shape, generator=g)
        pr.grad, pm.grad = gr.clone(), gr.to(_dev())


def test_flat_adam_is_an_optimizer_the_reference_schedulers_drive():
    """train.py:121-124 wraps the optimizer in CosineAnnealingLR (default) / StepLR: the learning rate is read from
    param_groups[0]['lr'] at every step."""
    from torch.optim.lr_scheduler import CosineAnnealingLR, StepLR
    for make in (lambda o: CosineAnnealingLR(o, 6, eta_min=1e-3), lambda o: StepLR(o, 2, 0.5)):
        g, ref, mine, opt_r, opt = _adam_pair([(64, 8, 1, 1), (64,), (300, 7)])
        assert isinstance(opt, torch.optim.Optimizer) and len(opt.param_groups) == 1
        opt.param_groups[0]["lr"] = opt_r.param_groups[0]["lr"] = 0.05
        sch_r, sch = make(opt_r), make(opt)
        for _ in range(6):
            _feed(g, ref, mine)
            opt_r.step()
            opt.step()
            sch_r.step()
            sch.step()
            assert opt.param_groups[0]["lr"] == pytest.approx(opt_r.param_groups[0]["lr"])
        for pr, pm in zip(ref, mine):
            assert (pm.detach().cpu() - pr.detach()).abs().max().item() <= 2e-5


def test_flat_adam_state_dict_interchanges_with_torch_adam():
    """train.py:131 / :306-309: `optimizer_state_dict` of an existing .t7 checkpoint loads into FlatAdam and a
    checkpoint FlatAdam wrote loads into torch.optim.Adam."""
    import io
    shapes = [(64, 8, 1, 1), (64,), (64,), (5, 256, 1)]
    g, ref, mine, opt_r, opt = _adam_pair(shapes)
    for _ in range(3):
        _feed(g, ref, mine)
        opt_r.step()
        opt.step()
    sd_r, sd = opt_r.state_dict(), opt.state_dict()
    assert sorted(sd.keys()) == sorted(sd_r.keys()) and sorted(sd["state"].keys()) == sorted(sd_r["state"].keys())
    assert set(sd_r["param_groups"][0].keys()) <= set(sd["param_groups"][0].keys())
    for i in sd_r["state"]:
        assert set(sd["state"][i].keys()) == set(sd_r["state"][i].keys())
        assert float(sd["state"][i]["step"]) == float(sd_r["state"][i]["step"]) == 3.0
        for key in ("exp_avg", "exp_avg_sq"):
            assert (sd["state"][i][key].cpu() - sd_r["state"][i][key]).abs().max().item() <= 1e-6
    # torch.save / torch.load round trip in both directions, then three more steps side by side
    def through_file(obj):
        buf = io.BytesIO()
        torch.save(obj, buf)
        buf.seek(0)
        return torch.load(buf, weights_only=False)
    g2, ref2, mine2, opt_r2, opt2 = _adam_pair(shapes)
    with torch.no_grad():
        for p_new, p_old in zip(ref2, mine):
            p_new.copy_(p_old.cpu())
        for p_new, p_old in zip(mine2, ref):
            p_new.copy_(p_old.to(_dev()))
    opt_r2.load_state_dict(through_file(sd))       # FlatAdam -> torch.optim.Adam
    opt2.load_state_dict(through_file(sd_r))       # torch.optim.Adam -> FlatAdam
    assert opt2.steps == 3
    for _ in range(3):
        _feed(g, ref, mine)
        for a, b in zip(ref2, mine):
            a.grad = b.grad.cpu()
        for a, b in zip(mine2, ref):
            a.grad = b.grad.to(_dev())
        for o in (opt_r, opt, opt_r2, opt2):
            o.step()
    for pr, pm, pr2, pm2 in zip(ref, mine, ref2, mine2):
        assert (pm.detach().cpu() - pr.detach()).abs().max().item() <= 2e-6
        assert (pr2.detach() - pr.detach()).abs().max().item() <= 2e-6
        assert (pm2.detach().cpu() - pr.detach()).abs().max().item() <= 2e-6


def test_flat_adam_skips_parameters_without_gradient_and_detects_moved_storage():
    g, ref, mine, opt_r, opt = _adam_pair([(32, 4), (16,), (8, 8), (5,)])
    for it in range(4):
        _feed(g, ref, mine, skip=(1,) if it % 2 else ())      # torch leaves a grad-less parameter and its step alone
        opt_r.step()
        opt.step()
    for pr, pm in zip(ref, mine):
        assert (pm.detach().cpu() - pr.detach()).abs().max().item() <= 2e-6
    assert float(opt.state_dict()["state"][1]["step"]) == float(opt_r.state_dict()["state"][1]["step"]) == 2.0
    mine[2].data = mine[2].data.clone()                        # what model.to() / .float() does to a parameter
    _feed(g, ref, mine)
    with pytest.raises(RuntimeError, match="flat buffer"):
        opt.step()


def test_points_to_float_is_exact():
    from ahn3_dgcnn_pytorch_amd import ops
    g = torch.Generator().manual_seed(1)
    x = torch.rand(3, 1001, 4, generator=g, dtype=torch.float64) * 30.0
    y = ops.points_to_float(x.to(_dev()))
    assert torch.equal(y.cpu(), x.float())


def test_training_step_with_device_loss_and_flat_adam():
    """fwd -> seg_loss -> bwd -> FlatAdam.step on a small model equals the same step with the reference's loss ops
    and torch.optim.Adam (both on the HIP model: the point is the loss + optimizer path)."""
    import copy
    from ahn3_dgcnn_pytorch_amd import DGCNN, ops
    from ahn3_dgcnn_pytorch_amd.optim import FlatAdam
    from oracle import seg_ref
    from param_fill import fill_module
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(4)
    x = torch.rand(2, 4, 300, generator=g).to(_dev())
    lab = torch.randint(0, 5, (2, 300), generator=g).to(_dev())
    a = DGCNN(5, 4, 8, dropout=0.0, emb_dims=1024, cuda=True)   # conv7 fixes emb_dims at 1024 (model.py:134)
    fill_module(a)
    b = copy.deepcopy(a)
    a.train()
    b.train()
    opt_a = torch.optim.Adam(a.parameters(), lr=1e-3, weight_decay=1e-4)
    opt_b = FlatAdam(b.parameters(), lr=1e-3, weight_decay=1e-4)
    # Step 1 starts from identical weights: same loss, same gradients, same update.  Adam's first update is
    # lr * sign(g) for every weight, so after it a weight whose gradient is rounding noise may sit 2 lr apart in the two
    # runs, and max-over-k / max-over-points winners near a tie may change: from step 2 on only the loss is compared.
    def one_step(tol):
        opt_a.zero_grad()
        la, _ = seg_ref.step_loss(a(x), lab)
        la.backward()
        opt_b.zero_grad()
        lb, _ = ops.seg_loss(b(x), lab)
        lb.backward()
        assert abs(la.item() - lb.item()) <= tol * max(1.0, abs(la.item()))
        grads = [(n, pa.grad.clone(), pb.grad.clone()) for (n, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters())]
        opt_a.step()
        opt_b.step()
        return grads

    # (some gradients are zero in exact arithmetic -- e.g. bn6.bias: a shift of the pooled feature is removed by bn7 --
    # and pure rounding noise in both runs, which Adam turns into +-lr steps: compare above a noise floor only)
    grads = one_step(1e-6)
    for (n, ga, gb), (_, pa), (_, pb) in zip(grads, a.named_parameters(), b.named_parameters()):
        floor = 1e-6 * ga.numel() ** 0.5
        assert (ga - gb).norm().item() <= 1e-3 * max(ga.norm().item(), floor), n
        solid = ga.abs() > 1e-3          # well above the weight-decay term (1e-4 |w|) and the noise
        if solid.any():
            assert (pa - pb).abs()[solid].max().item() <= 2e-5, n
    one_step(2e-4)


def test_model_input_is_a_view_of_point_major_memory():
    from ahn3_dgcnn_pytorch_amd import blocks
    pts = torch.rand(2, 100, 4, dtype=torch.float64)
    x = blocks.to_model_input(pts)
    assert x.shape == (2, 4, 100) and x.dtype == torch.float32 and x.is_cuda
    assert x.permute(0, 2, 1).is_contiguous()
    assert torch.equal(x.cpu(), pts.permute(0, 2, 1).float())


def test_classify_block_matches_reference_steps():
    import numpy as np
    from ahn3_dgcnn_pytorch_amd import DGCNN, blocks
    from oracle import seg_ref
    from param_fill import fill_module
    m = DGCNN(5, 4, 8, dropout=0.5, emb_dims=1024, cuda=True)
    fill_module(m)
    rs = np.random.RandomState(3)
    X = rs.rand(400, 4) * np.array([30.0, 30.0, 20.0, 20.0]) + np.array([1000.0, 2000.0, 0.0, 0.0])
    X0 = X.copy()
    pred = blocks.classify_block(m, X, min_class_confidence=0.3)
    assert np.array_equal(X, X0) and pred.shape == (400, 1)
    # the same steps with the reference's tensor ops on the model's logits
    Xl = X - np.array([X[:, 0].min(), X[:, 1].min(), 0.0, 0.0])
    m.eval()
    with torch.no_grad():
        logits = m(torch.tensor(Xl[:, :, np.newaxis]).permute(2, 1, 0).float().cuda()).cpu()
    pred_r, val_r = seg_ref.predict_step(logits, 0.3)
    safe = ((val_r - 0.3).abs() > 1e-5).reshape(-1).numpy()
    assert np.array_equal(pred.reshape(-1)[safe], pred_r.reshape(-1).numpy()[safe])


def test_classify_blocks_stacks_in_block_order():
    import numpy as np
    from ahn3_dgcnn_pytorch_amd import DGCNN, blocks
    from param_fill import fill_module
    m = DGCNN(5, 4, 8, dropout=0.5, emb_dims=1024, cuda=True)
    fill_module(m)
    rs = np.random.RandomState(5)
    tiles = [rs.rand(n, 4) * 30.0 for n in (120, 64, 200)]
    pts, preds = blocks.classify_blocks(m, tiles, 0.25)
    assert pts.shape == (384, 4) and preds.shape == (384, 1)
    assert np.array_equal(pts[120:184], tiles[1])
    assert np.array_equal(preds[120:184], blocks.classify_block(m, tiles[1], 0.25))
