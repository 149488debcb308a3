"""GPU parity of the DGCNN host mirror (EdgeConv blocks on HIP kernels, head on torch) against the
golden vectors captured from the reference (tests/golden/, made by gen_golden.py).

Tolerances: logits / activations 1e-4 absolute (BASELINE.json north_star); gradients 1e-4 of each
tensor's largest entry.  The 1e-4 gates are asserted with the reference's neighbour graphs injected
(SURVEY.md 8c gate 3); the end-to-end runs (own kNN) are checked for graph agreement as well."""
import json
import os

import numpy as np
import pytest
import torch

from param_fill import cotangent, fill_module

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    return np.load(os.path.join(GOLD, name))


def small_model(train, device):
    from ahn3_dgcnn_pytorch_amd import DGCNN
    g = gold("model_small.npz")
    m = DGCNN(int(g["num_classes"]), 4, int(g["k"]), dropout=0.0, emb_dims=1024, cuda=True)
    fill_module(m)
    m.train(train)
    return g, m, torch.from_numpy(g["x"]).to(device)


def idx_list(g, prefix, device):
    return [torch.from_numpy(g["%s_idx%d" % (prefix, i)].astype(np.int32)).to(device) for i in (1, 2, 3)]


@pytest.mark.parametrize("depth", range(1, 11))
def test_eval_all_depths_with_reference_graph(depth, device):
    g, m, x = small_model(False, device)
    with torch.no_grad():
        y = m(x, depth=depth, idx=idx_list(g, "eval", device))
    assert y.is_contiguous()
    step = int(g["eval_d%d_step" % depth])
    want = g["eval_d%d" % depth]
    got = y.cpu().numpy()[:, ::step]
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-4)


def test_eval_end_to_end_own_knn(device):
    g, m, x = small_model(False, device)
    with torch.no_grad():
        y, used = m(x, return_idx=True)
    # block 1 searches in the raw coordinates: identical lists; blocks 2/3 search in computed
    # features (1e-6 apart between implementations): identical neighbour SETS on this tie-free input
    np.testing.assert_array_equal(used[0].cpu().numpy(), g["eval_idx1"])
    for i in (1, 2):
        a = np.sort(used[i].cpu().numpy(), axis=-1)
        b = np.sort(g["eval_idx%d" % (i + 1)].astype(np.int32), axis=-1)
        assert (a == b).all(-1).mean() == 1.0
    np.testing.assert_allclose(y.cpu().numpy(), g["eval_d10"], rtol=0, atol=1e-4)


def test_train_step_matches_reference(device):
    g, m, x = small_model(True, device)
    idx = idx_list(g, "train", device)
    y = m(x, idx=idx)
    np.testing.assert_allclose(y.detach().cpu().numpy(), g["train_d10"], rtol=0, atol=1e-4)
    R = cotangent(tuple(y.shape)).to(device)
    (y * R).sum().backward()
    sd = m.state_dict()
    for key in g.files:
        if key.startswith("train_after/"):
            np.testing.assert_allclose(sd[key.split("/", 1)[1]].cpu().numpy(), g[key], rtol=1e-5, atol=1e-6)
    for i in range(1, 9):
        assert int(sd["bn%d.num_batches_tracked" % i]) == 1
    params = dict(m.named_parameters())
    checked = 0
    for key in g.files:
        if key.startswith("grad/"):
            want, got = g[key], params[key[5:]].grad.cpu().numpy()
        elif key.startswith("grad_sample/"):
            want, got = g[key], params[key[12:]].grad.cpu().numpy().reshape(-1)[::97]
        else:
            continue
        scale = float(np.abs(want).max())
        err = float(np.abs(got - want).max())
        # 2e-5 floor: a few head gradients vanish analytically (|g| ~ 1e-6 while their siblings are O(1..100))
        assert err <= 1e-4 * scale + 2e-5, "%s: err %.3g scale %.3g" % (key, err, scale)
        checked += 1
    assert checked == 25


@pytest.mark.parametrize("depth", [1, 2, 3])
def test_train_block_outputs(depth, device):
    g, m, x = small_model(True, device)
    y = m(x, depth=depth, idx=idx_list(g, "train", device))
    np.testing.assert_allclose(y.detach().cpu().numpy(), g["train_d%d" % depth], rtol=0, atol=1e-4)


def test_readme_config_logits(device):
    # BASELINE.json configs[0]: DGCNN(5, 4, 30, emb_dims=1024, cuda=False) on a [1, 4, 1000] tensor
    from ahn3_dgcnn_pytorch_amd import DGCNN
    g = gold("model_readme.npz")
    m = DGCNN(5, 4, 30, dropout=0.5, emb_dims=1024, cuda=False)
    fill_module(m)
    m.eval()
    x_cpu = torch.from_numpy(g["x"])
    idx = [torch.from_numpy(g["idx%d" % i].astype(np.int32)).to(device) for i in (1, 2, 3)]
    with torch.no_grad():
        y = m(x_cpu, idx=idx)
        y_own, used = m(x_cpu, return_idx=True)
    assert y.device.type == "cpu" and y.shape == (1, 5, 1000)  # CPU in -> CPU out, like the reference
    np.testing.assert_allclose(y.numpy(), g["logits"], rtol=0, atol=1e-4)
    # end to end with the HIP kNN: report/limit the fraction of points off by more than 1e-4
    # (agreement with PyTorch-CPU's index lists is a RATE: fp32 near-ties order differently, SURVEY.md 8c gate 2)
    rows_equal = (used[0].cpu().numpy() == g["idx1"]).all(-1)
    assert rows_equal.mean() >= 0.99
    frac = float((np.abs(y_own.numpy() - g["logits"]).max(axis=1) > 1e-4).mean())
    assert frac <= 0.01, "fraction of points whose logits differ by > 1e-4 end-to-end: %.4f" % frac
    # strided callers: train.py:176 / test.py:119 hand a permuted view of [B, N, F]
    xs = x_cpu.permute(0, 2, 1).contiguous().permute(0, 2, 1)
    assert xs.stride() == (4000, 1, 4)
    with torch.no_grad():
        ys = m(xs.to(device), idx=idx)
    assert ys.is_cuda
    np.testing.assert_allclose(ys.cpu().numpy(), g["logits"], rtol=0, atol=1e-4)


@pytest.mark.timeout(900)
def test_full_model_eval_at_benchmark_size(device):
    """BASELINE.json configs[2]: full DGCNN eval forward at B=8, N=7000, k=30 on block-like coordinates (x, y in
    [0, 30), z, AGL in [0, 20)).  SURVEY.md 8c at full size:
      gate 3  with the CPU restatement's own neighbour graphs injected, logits within 1e-4 everywhere;
      gate 1  end to end, the graph of every block is bit-exact under the fixed arithmetic GIVEN the features the
              block was fed (the graph is "dynamic": a function of computed features);
      gate 2  every end-to-end difference to the reference is explained by the graphs: the CPU restatement run on
              the HIP path's graphs reproduces the HIP logits within 1e-4 everywhere.  (A per-point fraction against
              the reference's own graphs is not a usable gate: one changed neighbour list that moves a winner of the
              global max-pool at model.py:175 shifts every point of its batch element -- measured here: 50 % of the
              points, i.e. 4 of the 8 batch elements, with block-1 lists equal on > 99 % of the rows.)"""
    from ahn3_dgcnn_pytorch_amd import DGCNN
    from oracle import knn_exact as ke
    from oracle import torch_ref
    B, N, k = 8, 7000, 30
    gen = torch.Generator().manual_seed(8)
    x = (torch.rand(B, N, 4, generator=gen) * torch.tensor([30.0, 30.0, 20.0, 20.0])).permute(0, 2, 1)  # train.py:176 strides
    ref = torch_ref.RefDGCNN(5, 4, k, dropout=0.5)
    fill_module(ref)
    ref.eval()
    torch.set_num_threads(min(16, torch.get_num_threads()))
    with torch.no_grad():
        want, used_ref = ref(x, return_idx=True)
    m = DGCNN(5, 4, k, dropout=0.5, emb_dims=1024, cuda=True)
    fill_module(m)
    m.eval()
    xd = x.to(device)
    with torch.no_grad():
        got = m(xd, idx=[u.to(torch.int32).to(device) for u in used_ref])
        own, used = m(xd, return_idx=True)
        feats = [xd] + [m(xd, depth=d, idx=used) for d in (1, 2)]      # what each block's search was fed
    err = (got.cpu() - want).abs()
    assert float(err.max()) <= 1e-4, "max |logit - reference| = %.3g" % float(err.max())
    for bi in range(3):
        np.testing.assert_array_equal(used[bi].cpu().numpy(), ke.knn_exact(feats[bi].cpu().contiguous().numpy(), k),
                                      err_msg="block %d graph" % (bi + 1))
    rows_equal = (used[0].cpu().numpy() == used_ref[0].numpy()).all(-1).mean()
    assert rows_equal >= 0.99, rows_equal
    with torch.no_grad():
        explained = ref(x, idx=[u.cpu().long() for u in used])
    err2 = (own.cpu() - explained).abs()
    assert float(err2.max()) <= 1e-4, "max |HIP logit - CPU restatement on the HIP graphs| = %.3g" % float(err2.max())


def test_predict_style_input_and_ropout_alias(device):
    # test.py:304-306 builds X[N, F, 1] -> permute(2, 1, 0): fp32 [1, F, N] with strides (1, 1, F)
    from ahn3_dgcnn_pytorch_amd import DGCNN
    m = DGCNN(5, 4, 20, ropout=0.3, emb_dims=1024, cuda=True)
    assert abs(m.dp1.p - 0.3) < 1e-12
    fill_module(m)
    m.eval()
    X = torch.rand(500, 4, 1)
    v = X.permute(2, 1, 0).float()
    with torch.no_grad():
        a = m(v.to(device))
        b = m(v.contiguous().to(device))
    assert torch.equal(a, b)


def test_state_dict_roundtrip_with_reference_keys(device, tmp_path):
    from ahn3_dgcnn_pytorch_amd import DGCNN
    with open(os.path.join(GOLD, "state_dict.json")) as f:
        spec = json.load(f)
    m = DGCNN(5, 4, 30, cuda=True)
    sd = m.state_dict()
    assert list(sd.keys()) == list(spec["entries"].keys())
    for k, (shape, dtype) in spec["entries"].items():
        assert list(sd[k].shape) == shape and str(sd[k].dtype) == "torch." + dtype
    for k in spec["aliased_bn_keys"]:  # bnN.* and convN.1.* are the same storage (model.py:107-139)
        assert sd[k].data_ptr() == sd["conv" + k[2] + ".1" + k[3:]].data_ptr()
    # checkpoint interchange in the reference's format (train.py:301-312)
    path = str(tmp_path / "model.t7")
    torch.save({"epoch": 3, "model_state_dict": {k: v.cpu() for k, v in sd.items()}}, path)
    m2 = DGCNN(5, 4, 30, cuda=True)
    m2.load_state_dict(torch.load(path)["model_state_dict"])
    for k in sd:
        assert torch.equal(m2.state_dict()[k], sd[k])
    assert m.k == 30 and m.num_classes == 5 and m.num_features == 4 and m.use_cuda is True


def test_determinism_forward(device):
    g, m, x = small_model(False, device)
    with torch.no_grad():
        a = m(x)
        b = m(x)
    assert torch.equal(a, b)


def _near_tie_census(ref, x, idx, gap=2e-6):
    """Counts, in the CPU restatement's own fp32 forward (train mode), the places where the gradient routing is decided
    by less than `gap` -- there the reference's own result is rounding-dependent and two correct implementations may
    differ by O(1) in ONE contribution:
      edge  : (point, channel) maxima over k whose two best activations are closer than gap   (of 3 x P x 64)
      pool  : (batch element, channel) maxima over the points with the same property          (of B x 1024)
      kink  : pre-activations of a LeakyReLU within gap of zero (derivative 1 or 0.2)         (of all activations)
    Returns the relative-L2 allowance sqrt(sum n_i / M_i) those could explain, and the counts."""
    counts = {"edge": [0, 0], "pool": [0, 0], "kink": [0, 0]}
    hooks = []

    def bn_hook(_m, _inp, out):
        counts["kink"][0] += int((out.abs() < gap).sum())
        counts["kink"][1] += out.numel()

    def edge_hook(_m, _inp, out):          # output of the last stage of a block: [B, 64, N, k]
        top2 = out.topk(2, dim=-1)[0]
        counts["edge"][0] += int(((top2[..., 0] - top2[..., 1]) < gap).sum())
        counts["edge"][1] += top2[..., 0].numel()

    def pool_hook(_m, _inp, out):          # conv6's activated output [B, 1024, N]
        top2 = out.topk(2, dim=-1)[0]
        counts["pool"][0] += int(((top2[..., 0] - top2[..., 1]) < gap).sum())
        counts["pool"][1] += top2[..., 0].numel()

    for i in range(1, 9):
        hooks.append(getattr(ref, "bn%d" % i).register_forward_hook(bn_hook))
    for name in ("conv2", "conv4", "conv5"):
        hooks.append(getattr(ref, name).register_forward_hook(edge_hook))
    hooks.append(ref.conv6.register_forward_hook(pool_hook))
    y = ref(x, idx=idx)
    for h in hooks:
        h.remove()
    # one flipped decision replaces one unit contribution by another: 2 n / M of the squared norm
    allowance = sum(2.0 * n / max(1, m) for n, m in counts.values()) ** 0.5
    return y, allowance, {k_: v[0] for k_, v in counts.items()}


@pytest.mark.timeout(1500)
def test_full_model_train_step_at_benchmark_size(device):
    """The composition the block- and head-level full-size tests leave open (model.py:167-191): full DGCNN, train mode
    (batch statistics in all eight BatchNorms), forward AND backward at B=8, N=7000, k=30 with the CPU restatement's own
    neighbour graphs injected; dropout 0 so that both sides see the same network.  Covers the three gradient fan-ins
    into x1 / x2 / x3 (block b+1's edge features, the cat of :167 and the cat of :180), the one-call conv6 backward and
    the size-selected kernels of the head at the size bench.py times.
      logits           within 1e-4 everywhere (north_star);
      running stats    as nn.BatchNorm updates them;
      all 25 gradients in relative L2 within 5e-4 + three times what a 1e-6 relative perturbation of the reference's own
                       weights does to its gradients (the measured effect of the near-ties of max-over-k and of the
                       LeakyReLU kinks, which cannot be avoided at this size; inputs with a near-tie of the max over the
                       points are skipped)."""
    from ahn3_dgcnn_pytorch_amd import DGCNN
    from oracle import torch_ref
    B, N, k = 8, 7000, 30
    torch.set_num_threads(min(32, max(1, (os.cpu_count() or 8))))
    # a near-tie of the max over the POINTS (model.py:175) moves one of only B x 1024 large contributions (1.6e-2 of every
    # upstream gradient: measured with seed 21, which has one): the seed is advanced until the input has none; the
    # near-ties of max-over-k and of the LeakyReLU kinks cannot be avoided at this size and are priced
    for seed in range(21, 27):
        gen = torch.Generator().manual_seed(seed)
        x = (torch.rand(B, N, 4, generator=gen) * torch.tensor([30.0, 30.0, 20.0, 20.0])).permute(0, 2, 1)
        ref = torch_ref.RefDGCNN(5, 4, k, dropout=0.0)
        fill_module(ref)
        ref.eval()
        with torch.no_grad():
            _, used = ref(x, return_idx=True)      # the restatement's own graphs (eval statistics: any fixed graph serves)
        ref.train()
        want, allowance, counts = _near_tie_census(ref, x, used)
        if counts["pool"] == 0:
            break
        del want
    assert counts["pool"] == 0, "no input without a near-tie of the max over the points found (%s)" % counts
    assert allowance <= 2e-2, "input rejected: near-tie allowance %.3g (%s)" % (allowance, counts)
    R = cotangent(tuple(want.shape), tag="fullstep")
    (want * R).sum().backward()
    # What the counted near-ties do to each gradient is measured, not estimated: the reference itself is evaluated a second time
    # with its convolution weights perturbed by 1e-6 relative (the size of the forward differences between two correct fp32
    # implementations: smoke() reports |y - oracle| ~ 3e-6; the contract allows 1e-4).  The distance between its two gradient
    # sets is what decisions taken by less than that are worth at this size -- the gradient of a first-layer parameter collects
    # the flips of all three blocks: bn1.weight moves by 1.2e-2 between the HIP path and the fp32 restatement with 489 near-ties
    # (0.5e-2 between the restatement and its own float64 evaluation, whose roundings differ by 1e-7 only).  Gate: the HIP
    # gradients are within 5e-4 + THREE times that distance of the reference's.
    import copy
    ref2 = copy.deepcopy(ref)
    for p in ref2.parameters():
        p.grad = None
    with torch.no_grad():
        for i in range(1, 9):
            w = getattr(ref2, "conv%d" % i)[0].weight
            flip = torch.where(torch.arange(w.numel()).reshape(w.shape) % 2 == 0, 1.0, -1.0)
            w.mul_(1.0 + 1e-6 * flip)
    ref2.train()
    y2 = ref2(x, idx=used)
    (y2 * R).sum().backward()
    self_rel, denom = {}, {}
    p32 = dict(ref.named_parameters())
    # (bn6.bias has NO gradient in train mode -- a constant added to the pooled feature is removed by bn7's batch mean --, so
    # norms are floored at 1e-4 of the largest RMS gradient of the model: there the gate becomes an absolute one)
    rms = max(float(p.grad.norm()) / p.grad.numel() ** 0.5 for p in ref.parameters())
    for name, p in ref2.named_parameters():
        denom[name] = max(float(p32[name].grad.norm()), 1e-4 * rms * p.grad.numel() ** 0.5)
        self_rel[name] = float((p32[name].grad - p.grad).norm()) / denom[name]
    del ref2, y2
    assert max(self_rel.values()) <= 1.5e-2, "input rejected: a 1e-6 perturbation moves the reference's gradients by %.3g" % max(self_rel.values())

    m = DGCNN(5, 4, k, dropout=0.0, emb_dims=1024, cuda=True)
    fill_module(m)
    m.train()
    got = m(x.to(device), idx=[u.to(torch.int32).to(device) for u in used])
    err = float((got.detach().cpu() - want.detach()).abs().max())
    assert err <= 1e-4, "max |logit - reference| = %.3g" % err
    (got * R.to(device)).sum().backward()
    torch.cuda.synchronize()
    sd, sd_ref = m.state_dict(), ref.state_dict()
    for i in range(1, 9):
        for what in ("running_mean", "running_var"):
            key = "bn%d.%s" % (i, what)
            np.testing.assert_allclose(sd[key].cpu().numpy(), sd_ref[key].numpy(), rtol=2e-5, atol=2e-6, err_msg=key)
        assert int(sd["bn%d.num_batches_tracked" % i]) == 1
    checked, worst = 0, (0.0, "", 0.0)
    ref_params = dict(ref.named_parameters())
    for name, p in m.named_parameters():
        g_ref = ref_params[name].grad
        assert p.grad is not None and g_ref is not None, name
        rel = float((p.grad.cpu() - g_ref).norm()) / denom[name]
        tol = 5e-4 + 3.0 * self_rel[name]
        worst = max(worst, (rel / tol, name, rel))
        assert rel <= tol, "%s: relative L2 %.3g > %.3g (reference under a 1e-6 weight perturbation: %.3g; near-ties %s)" % (
            name, rel, tol, self_rel[name], counts)
        checked += 1
    assert checked == 25
    print("full-model train step: logits %.2e, worst gradient %s %.2e (%.2f of its tolerance; near-ties %s)" % (
        err, worst[1], worst[2], worst[0], counts))


@pytest.mark.timeout(1500)
def test_full_model_eval_at_stress_size(device):
    """BASELINE.json configs[4] shape, one batch element: full DGCNN eval forward at N=32768, k=64 (the N at which the
    reference's B x N x N temporaries are 4.3 GB each: the CPU restatement is run on INJECTED graphs, which skips them).
      gate 1  block 1's graph (raw coordinates) is bit-exact under the fixed arithmetic;
      gate 2  "explained": the CPU restatement run on the HIP path's own three graphs reproduces the HIP logits within
              1e-4 everywhere."""
    from ahn3_dgcnn_pytorch_amd import DGCNN
    from oracle import knn_exact as ke
    from oracle import torch_ref
    B, N, k = 1, 32768, 64
    gen = torch.Generator().manual_seed(31)
    x = (torch.rand(B, N, 4, generator=gen) * torch.tensor([30.0, 30.0, 20.0, 20.0])).permute(0, 2, 1)
    m = DGCNN(5, 4, k, dropout=0.5, emb_dims=1024, cuda=True)
    fill_module(m)
    m.eval()
    with torch.no_grad():
        own, used = m(x.to(device), return_idx=True)
    np.testing.assert_array_equal(used[0].cpu().numpy(), ke.knn_exact(x.contiguous().numpy(), k), err_msg="block 1 graph")
    ref = torch_ref.RefDGCNN(5, 4, k, dropout=0.5)
    fill_module(ref)
    ref.eval()
    torch.set_num_threads(min(32, max(1, (os.cpu_count() or 8))))
    with torch.no_grad():
        explained = ref(x, idx=[u.cpu().long() for u in used])
    err = float((own.cpu() - explained).abs().max())
    assert err <= 1e-4, "max |HIP logit - CPU restatement on the HIP graphs| = %.3g" % err


@pytest.mark.parametrize("F_,k,N,B", [(4, 9, 260, 3), (3, 20, 131, 2), (6, 33, 300, 1), (4, 6, 130, 18)])   # (the CLI's default is 3 features, k = 20)
def test_model_level_call_against_the_per_layer_path(F_, k, N, B, device, monkeypatch):
    """DGCNN.forward at depth 10 runs as dgcnn_model_fwd / _bwd (one call each way).  The ways through the same kernels
    must agree: (1) own kNN, the reverse lists prepared by riders in the forward (training mode); (0) DGCNN_NO_FWD_LISTS=1: the
    backward builds them -- the three graphs in ONE array -> in one go; (2) the same graphs injected as three separate tensors:
    all bit-identical to (1) (the lists are sorted by edge id whoever builds them); (3) DGCNN_PER_LAYER=1: the per-layer entry points composed in Python, where the
    gradient fan-in of x1 / x2 / x3 is added by autograd in another order: equal within rounding."""
    from ahn3_dgcnn_pytorch_amd import DGCNN
    g = torch.Generator().manual_seed(12)
    x = (torch.rand(B, N, F_, generator=g) * 20.0).permute(0, 2, 1).to(device)
    m = DGCNN(5, F_, k, dropout=0.0, emb_dims=1024, cuda=True)
    fill_module(m)
    m.train()
    R = cotangent((B, 5, N), tag="paths").to(device)

    def run(idx):
        m.zero_grad(set_to_none=True)
        y, used = m(x, idx=idx, return_idx=True)
        (y * R).sum().backward()
        return y.detach().clone(), used, {n: p.grad.detach().clone() for n, p in m.named_parameters()}

    y1, used, g1 = run(None)   # (training forward: the list passes of the three blocks ride in the forward's / backward's launches)
    # (0) the backward builds the lists itself: the three graphs in ONE array -> one merged build; three tensors -> per block
    monkeypatch.setenv("DGCNN_NO_FWD_LISTS", "1")
    y0, used0, g0 = run(None)
    assert used0[1].data_ptr() == used0[0].data_ptr() + 4 * used0[0].numel()       # one array: the merged build ran
    y0b, _, g0b = run([u.clone() for u in used])
    monkeypatch.delenv("DGCNN_NO_FWD_LISTS")
    for ya, ga in ((y0, g0), (y0b, g0b)):
        assert torch.equal(y1, ya)
        for n in g1:
            assert torch.equal(g1[n], ga[n]), n
    y2, _, g2 = run([u.clone() for u in used])
    assert torch.equal(y1, y2)
    for n in g1:
        assert torch.equal(g1[n], g2[n]), n
    monkeypatch.setenv("DGCNN_PER_LAYER", "1")
    y3, _, g3 = run([u.clone() for u in used])
    assert float((y3 - y1).abs().max()) <= 2e-5
    for n in g1:
        scale = float(g1[n].abs().max())
        assert float((g3[n] - g1[n]).abs().max()) <= 1e-4 * scale + 2e-5, n
