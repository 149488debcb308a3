"""Generate the golden vectors under tests/golden/ by IMPORTING THE REFERENCE on CPU.

Run in the build container only (the reference never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py

What it does
  1. imports /root/reference/model.py (knn :25, get_graph_feature :43, DGCNN :85) with the one
     shim SURVEY.md D1 prescribes: the stray `dim9=` keyword of model.py:147 is swallowed;
  2. loads the closed-form parameter fill of tests/param_fill.py into the reference;
  3. records inputs and the reference's outputs as small .npz fixtures (data only);
  4. asserts, while doing so, that oracle/torch_ref.py is BIT-EQUAL to the imported reference
     on every recorded quantity and that oracle/knn_exact.c reproduces the reference's index
     lists on the tie-free kNN inputs -- this is what pins the two oracles.

The fixtures record the torch version they were made with (`torch_version`).
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

import model as ref  # noqa: E402  (the reference)

from oracle import knn_exact as ke  # noqa: E402
from oracle import torch_ref  # noqa: E402
from param_fill import cotangent, fill_module  # noqa: E402

_orig_ggf = ref.get_graph_feature
ref.get_graph_feature = lambda x, k=20, use_cuda=True, dim9=False: _orig_ggf(x, k=k, use_cuda=use_cuda)  # D1 shim

torch.manual_seed(0)
torch.set_num_threads(8)
META = {"torch_version": torch.__version__, "numpy_version": np.__version__}


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, torch_version=np.array(torch.__version__), **arrays)
    print("wrote %s (%.1f KB)" % (name, os.path.getsize(path) / 1024))


def npf(t):
    return t.detach().cpu().numpy()


def sub(t, step):
    """strided subsample along the channel axis (axis 1) for large tensors."""
    return npf(t)[:, ::step]


def min_gap(x_bcn, k):
    """smallest gap between consecutive pd values among the best k+1 of every row (oracle arithmetic)."""
    _, pd = ke.knn_exact(x_bcn, min(k + 1, x_bcn.shape[2]), return_pd=True)
    return float(np.min(pd[:, :, :-1] - pd[:, :, 1:]))


def boundary_gap(x_bcn, k):
    _, pd = ke.knn_exact(x_bcn, k + 1, return_pd=True)
    return float(np.min(pd[:, :, k - 1] - pd[:, :, k]))


# ---------------------------------------------------------------------------------------------
# G1 / G2: knn and get_graph_feature on tie-free inputs
def gen_knn():
    out = {}
    for tag, C, dist in (("c4", 4, "uniform"), ("c64", 64, "normal")):
        best = None
        for seed in range(200):
            g = torch.Generator().manual_seed(seed)
            x = torch.rand(2, C, 256, generator=g) if dist == "uniform" else torch.randn(2, C, 256, generator=g)
            gap = min_gap(x.numpy(), 30)
            if best is None or gap > best[0]:
                best = (gap, seed, x)
            if gap > 1e-5:
                break
        gap, seed, x = best
        print("knn %s: seed %d min top-31 gap %.3g" % (tag, seed, gap))
        assert gap > 2e-6, "no tie-free seed found"
        out["x_" + tag] = x.numpy()
        out["gap_" + tag] = np.array(gap)
        for k in (8, 30):
            idx = ref.knn(x, k)
            assert torch.equal(idx, torch_ref.pairwise_knn(x, k))
            mine = ke.knn_exact(x.numpy(), k)
            assert np.array_equal(mine, idx.numpy()), "oracle/knn_exact.c disagrees with the reference on tie-free data"
            out["idx_%s_k%d" % (tag, k)] = idx.numpy().astype(np.int16)
        gf = ref.get_graph_feature(x, k=8, use_cuda=False)
        assert torch.equal(gf, torch_ref.edge_features(x, k=8))
        assert gf.shape == (2, 2 * C, 256, 8) and gf.stride() == (256 * 8 * 2 * C, 1, 8 * 2 * C, 2 * C)
        out["ggf_%s_k8" % tag] = npf(gf) if C == 4 else npf(gf)[:, :, ::16]
    save("knn_ggf.npz", **out)


# ---------------------------------------------------------------------------------------------
def block_like_input(B, N, seed):
    g = torch.Generator().manual_seed(seed)
    scale = torch.tensor([30.0, 30.0, 20.0, 20.0]).view(1, 4, 1)
    return torch.rand(B, 4, N, generator=g) * scale


def build_pair(num_classes, F, k, dropout):
    r = ref.DGCNN(num_classes, F, k, dropout=dropout, emb_dims=1024, cuda=False)
    m = torch_ref.RefDGCNN(num_classes, F, k, dropout=dropout, emb_dims=1024)
    fill_module(r)
    fill_module(m)
    return r, m


def layer_inputs(m, x):
    """inputs of the three kNN searches (x, x1, x2) as produced by the oracle in its current mode."""
    with torch.no_grad():
        x1 = m(x, depth=1)
        x2 = m(x, depth=2)
    return [x, x1, x2]


def premax_top2_gap(m, x):
    """smallest gap between the two largest activations entering any of the three max-over-k (current mode)."""
    gaps = []
    with torch.no_grad():
        cur = x
        for names in torch_ref.EDGE_BLOCKS:
            e = torch_ref.edge_features(cur, k=m.k)
            for nm in names:
                e = getattr(m, nm)(e)
            top2 = e.topk(2, dim=-1)[0]
            gaps.append(float((top2[..., 0] - top2[..., 1]).min()))
            cur = e.max(dim=-1)[0]
    return min(gaps)


# G3 / G4 / G5: small model, every depth, eval and train, gradients
def gen_model_small():
    # max-over-k routes the whole gradient of a (point, channel) to ONE edge; where the two best activations
    # differ by a few ulp the winner depends on rounding (inside the reference too), so the fixture input is
    # chosen free of such near-ties (and of near-ties at the k-th neighbour) in both modes
    B, F, N, k, ncls = 2, 4, 128, 8, 5
    best = None
    for seed in range(300):
        x = block_like_input(B, N, 100 + seed)
        r, m = build_pair(ncls, F, k, 0.0)
        crit = []
        for mode in ("eval", "train"):
            getattr(m, mode)()
            crit += [boundary_gap(npf(t), k) / max(1e-30, float(t.var()) * t.shape[1]) / 1e-5 for t in layer_inputs(m, x)]
            r2, m2 = build_pair(ncls, F, k, 0.0)
            getattr(m2, mode)()
            crit.append(premax_top2_gap(m2, x) / 2e-6)
        score = min(crit)
        if best is None or score > best[0]:
            best = (score, seed, x)
        if score >= 1.0:
            break
    score, seed, x = best
    print("model_small: seed %d score %.3g (>= 1: boundary gap >= 1e-5 rel, top-2 activation gap >= 2e-6)" % (seed, score))
    assert score >= 0.5
    out = {"x": npf(x), "rel_boundary_gap": np.array(score), "k": np.array(k), "num_classes": np.array(ncls)}

    # ---- eval mode, all depths
    r, m = build_pair(ncls, F, k, 0.0)
    r.eval(), m.eval()
    with torch.no_grad():
        for d in range(1, 11):
            yr = r(x, depth=d)
            ym, used = m(x, depth=d, return_idx=True)
            assert torch.equal(yr, ym), "torch_ref differs from the reference at depth %d (eval)" % d
            assert yr.is_contiguous()
            step = {5: 16, 7: 16, 8: 8, 9: 4}.get(d, 1)
            out["eval_d%d" % d] = sub(yr, step)
            out["eval_d%d_step" % d] = np.array(step)
        for i, u in enumerate(used):
            out["eval_idx%d" % (i + 1)] = u.numpy().astype(np.int16)

    # ---- train mode (dropout = 0): block outputs, logits, running statistics, gradients
    r, m = build_pair(ncls, F, k, 0.0)
    r.train(), m.train()
    for d in (1, 2, 3):
        r2, m2 = build_pair(ncls, F, k, 0.0)
        r2.train(), m2.train()
        yr, ym = r2(x, depth=d), m2(x, depth=d)
        assert torch.equal(yr, ym), "torch_ref differs from the reference at depth %d (train)" % d
        out["train_d%d" % d] = npf(yr)
    R = cotangent((B, ncls, N))
    yr = r(x)
    ym, used = m(x, return_idx=True)
    assert torch.equal(yr, ym)
    (yr * R).sum().backward()
    (ym * R).sum().backward()
    out["train_d10"] = npf(yr)
    for i, u in enumerate(used):
        out["train_idx%d" % (i + 1)] = u.numpy().astype(np.int16)
    sdr, sdm = r.state_dict(), m.state_dict()
    for key in sdr:
        assert torch.equal(sdr[key], sdm[key]), key
        if "running" in key and key.startswith("bn"):
            out["train_after/" + key] = npf(sdr[key])
    pr, pm = dict(r.named_parameters()), dict(m.named_parameters())
    assert list(pr) == list(pm) and len(pr) == 25
    for key in pr:
        # the reference's own backward is not run-to-run bit-stable on CPU (its IndexBackward
        # scatter-adds with atomics: observed spread up to 1.2e-6 of the tensor's max), so the
        # gradient check is relative, not bitwise
        scale = float(pr[key].grad.abs().max())
        assert float((pr[key].grad - pm[key].grad).abs().max()) <= 5e-6 * scale, key
        g = npf(pr[key].grad)
        edge = key.split(".")[0] in ("conv1", "conv2", "conv3", "conv4", "conv5", "bn1", "bn2", "bn3", "bn4", "bn5")
        if edge or g.size <= 4096:
            out["grad/" + key] = g
        else:  # large head tensors: strided sample + two norms
            out["grad_sample/" + key] = g.reshape(-1)[::97]
            out["grad_l2/" + key] = np.array(np.sqrt((g.astype(np.float64) ** 2).sum()))
    save("model_small.npz", **out)


# configs[0] of BASELINE.json: the README example (README.md:71-90)
def gen_model_readme():
    g = torch.Generator().manual_seed(2024)
    x = torch.rand(1, 4, 1000, generator=g)
    r, m = build_pair(5, 4, 30, 0.5)
    r.eval(), m.eval()
    with torch.no_grad():
        yr = r(x)
        ym, used = m(x, return_idx=True)
    assert torch.equal(yr, ym) and yr.shape == (1, 5, 1000)
    out = {"x": npf(x), "logits": npf(yr)}
    for i, u in enumerate(used):
        out["idx%d" % (i + 1)] = u.numpy().astype(np.int16)
    # strided callers: train.py:176 / test.py:119 hand a permuted view of [B, N, F]
    xs = x.permute(0, 2, 1).contiguous().permute(0, 2, 1)
    assert xs.stride() == (4000, 1, 4)
    with torch.no_grad():
        assert torch.equal(r(xs), yr)
    save("model_readme.npz", **out)


# G6: state_dict contract
def gen_state_dict():
    r = ref.DGCNN(5, 4, 30, dropout=0.5, emb_dims=1024, cuda=False)
    sd = r.state_dict()
    spec = {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()}
    alias = sorted(k for k in sd if k.startswith("bn") and sd[k].data_ptr() == sd["conv" + k[2] + ".1" + k[3:]].data_ptr())
    with open(os.path.join(HERE, "state_dict.json"), "w") as f:
        json.dump({"meta": META, "entries": spec, "aliased_bn_keys": alias,
                   "n_parameters": sum(p.numel() for p in r.parameters()),
                   "n_parameter_tensors": len(list(r.parameters())),
                   "attributes": {"k": r.k, "num_classes": r.num_classes, "use_cuda": r.use_cuda,
                                  "num_features": r.num_features}}, f, indent=1)
    print("wrote state_dict.json (%d entries)" % len(spec))


# G7: duplicate points -- documents that the reference's tie order is arbitrary (not a parity target)
def gen_dup():
    g = torch.Generator().manual_seed(5)
    half = torch.randn(1, 8, 250, generator=g)
    x = torch.cat([half, half], dim=2)
    idx = ref.knn(x, 4)
    mine = ke.knn_exact(x.numpy(), 4)
    same_set = np.mean([set(a) == set(b) for a, b in zip(idx.numpy().reshape(-1, 4), mine.reshape(-1, 4))])
    print("dup: reference rows equal to fixed-tie-break rows: %.3f; as sets: %.3f" % (
        float((idx.numpy() == mine).all(-1).mean()), same_set))
    save("dup_points.npz", x=npf(x), idx_reference_k4=idx.numpy().astype(np.int16), idx_fixed_tiebreak_k4=mine.astype(np.int16))


def _ref_nodes(path, first, last, inside=None):
    """the statements of a reference file whose source lines lie in [first, last] (taken from the outermost
    block that holds whole statements in that range), as an executable ast.Module -- the reference's own code, run here, never stored"""
    import ast
    tree = ast.parse(open(path).read())
    best = None
    for node in ast.walk(tree):
        for field in ("body", "orelse"):
            body = getattr(node, field, None)
            if not isinstance(body, list):
                continue
            picked = [st for st in body if isinstance(st, ast.stmt) and first <= st.lineno and st.end_lineno <= last]
            if picked and best is None:     # ast.walk is breadth-first: the outermost block that holds such lines
                best = picked
    assert best, "no statements found in %s:%d-%d" % (path, first, last)
    return ast.Module(body=best, type_ignores=[])


def gen_seg():
    """Pins oracle/seg_ref.py: util.py / train.py / test.py cannot be imported here
    (prettytable, tensorboard, h5py, laspy are absent), so the reference's OWN statements are taken out of the files
    with `ast` and executed on fixed inputs: cal_loss (util.py:17-34), the training step's softmax / mask / criterion /
    arg-max lines (train.py:180-197, :201), the validation step's (train.py:268-271) and the inference post-step
    (test.py:122-127)."""
    import torch.nn.functional as F
    from oracle import seg_ref
    ns = {"torch": torch, "F": F, "np": np}
    exec(compile(_ref_nodes("/root/reference/util.py", 17, 34), "util.py", "exec"), ns)
    cal_loss = ns["cal_loss"]
    g = torch.Generator().manual_seed(21)
    B, C, N = 2, 5, 96
    logits = torch.randn(B, C, N, generator=g) * 2.0
    labels = torch.randint(0, C, (B, N), generator=g).double()          # collate_pcs hands float64 labels
    mask = (torch.rand(B, N, generator=g) < 0.35).double()
    out = {"logits": npf(logits), "labels": npf(labels), "mask": npf(mask)}

    # training step, masked (train.py:180-197 and :201; loss.backward()/opt.step() at :198-199 are left out)
    step = _ref_nodes("/root/reference/train.py", 180, 197)
    tail = _ref_nodes("/root/reference/train.py", 201, 201)
    z = logits.clone().requires_grad_(True)
    env = dict(ns, labels_pred=z, labels=labels.clone(), mask=mask, num_classes=C, criterion=cal_loss)
    exec(compile(step, "train.py", "exec"), env)
    exec(compile(tail, "train.py", "exec"), env)
    env["loss"].backward()
    out.update(train_loss=npf(env["loss"]), train_pred=env["pred"].numpy(), train_grad=npf(z.grad))
    zr = logits.clone().requires_grad_(True)
    l2, p2 = seg_ref.step_loss(zr, labels, mask)
    l2.backward()
    assert abs(float(l2) - float(env["loss"])) <= 1e-7 and torch.equal(p2, env["pred"])
    assert float((zr.grad - z.grad).abs().max()) <= 1e-9

    # validation step, no mask (train.py:268-271; :267 is the model call)
    val = _ref_nodes("/root/reference/train.py", 268, 271)
    z = logits.clone().requires_grad_(True)
    env = dict(ns, labels_pred=z, labels=labels.clone(), num_classes=C, criterion=cal_loss)
    exec(compile(val, "train.py", "exec"), env)
    env["loss"].backward()
    out.update(val_loss=npf(env["loss"]), val_pred=env["pred"].numpy(), val_grad=npf(z.grad))
    zr = logits.clone().requires_grad_(True)
    l2, p2 = seg_ref.step_loss(zr, labels, None)
    l2.backward()
    assert abs(float(l2) - float(env["loss"])) <= 1e-7 and torch.equal(p2, env["pred"])
    assert float((zr.grad - z.grad).abs().max()) <= 1e-9

    # cal_loss itself on raw rows, smoothing on and off (util.py:17-34)
    rows = torch.randn(64, 7, generator=g)
    gold = torch.randint(0, 7, (64,), generator=g)
    out.update(rows=npf(rows), gold=gold.numpy(), cal_loss_smooth=npf(cal_loss(rows, gold, smoothing=True)),
               cal_loss_plain=npf(cal_loss(rows, gold, smoothing=False)))
    assert abs(float(seg_ref.cal_loss(rows, gold, True)) - float(out["cal_loss_smooth"])) <= 1e-7
    assert abs(float(seg_ref.cal_loss(rows, gold, False)) - float(out["cal_loss_plain"])) <= 1e-7

    # inference post-step (test.py:123-127): seg_pred = model(data) is the input
    post = _ref_nodes("/root/reference/test.py", 123, 127)
    for conf in (0.0, 0.3, 0.6):
        env = dict(ns, seg_pred=logits.clone(), softmax=F.softmax, min_class_confidence=conf, UNCLASSIFIED=31)
        exec(compile(post, "test.py", "exec"), env)
        out["predict_pred_%d" % int(conf * 10)] = env["pred"].numpy()
        out["predict_vals_%d" % int(conf * 10)] = npf(env["vals"])
        p2, v2 = seg_ref.predict_step(logits, conf)
        assert torch.equal(p2, env["pred"]) and torch.equal(v2, env["vals"])
    save("seg_steps.npz", **out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "seg":
        gen_seg()
        sys.exit(0)
    gen_knn()
    gen_model_small()
    gen_model_readme()
    gen_state_dict()
    gen_dup()
    gen_seg()
