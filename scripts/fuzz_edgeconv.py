"""dev helper: randomised GPU-vs-torch_ref comparison of the EdgeConv block (forward within 1e-4, gradients within
1e-4 of each tensor's maximum when the input has no near-tie of the two best activations)."""
import os, sys, time, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_edgeconv_gpu as T
from oracle import knn_exact as ke
from param_fill import cotangent

dev = torch.device("cuda:0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = skipped = 0
for it in range(n_cases):
    B = int(rng.choice([1, 2, 3, 16])); C = int(rng.choice([3, 4, 9, 16, 64])); N = int(rng.choice([40, 100, 257, 400, 700]))
    k = int(rng.choice([3, 8, 16, 20, 30, 40])); k = min(k, N - 1)
    n_stage = int(rng.choice([1, 2])); train = bool(rng.integers(0, 2))
    graph = str(rng.choice(["knn", "random", "hub"]))
    x = torch.from_numpy(rng.standard_normal((B, C, N)).astype(np.float32) * np.float32(10.0 ** rng.uniform(-1, 1)))
    if graph == "knn":
        idx = ke.knn_exact(x.numpy(), k)
    else:
        idx = np.stack([np.stack([rng.choice(N, size=k, replace=False) for _ in range(N)]) for _ in range(B)]).astype(np.int32)
        if graph == "hub":
            idx[:, :, 0] = 0
            for b in range(B):
                for i in range(N):  # keep the rows duplicate-free
                    if (idx[b, i, 1:] == 0).any():
                        idx[b, i, 1:][idx[b, i, 1:] == 0] = (set(range(1, N)) - set(idx[b, i].tolist())).pop()
    base = T.make_stages(2 * C, n_stage)
    # near-tie criterion relative to the activation scale: a top-2 gap of a few ulp of the largest activation can flip
    # the winner of max-over-k (on either side), which reroutes a whole gradient entry
    with torch.no_grad():
        e = __import__('oracle.torch_ref', fromlist=['x']).edge_features(x, k=k, idx=torch.from_numpy(idx.astype(np.int64)))
        for st in copy.deepcopy(base):
            st.train(train); e = st(e)
        amax = float(e.abs().max())
    tie_free = T.premax_top2_gap(x, idx, base, train) >= max(2e-6, 2e-6 * amax)
    R = cotangent((B, 64, N), tag="fuzz%d" % it)
    y0, gx0, st0 = T.run_oracle(x, idx, copy.deepcopy(base), train, R)
    y1, gx1, st1 = T.run_hip(x, idx, base, train, R, dev)
    what = "case %d: B=%d C=%d N=%d k=%d stages=%d train=%s graph=%s tie_free=%s" % (it, B, C, N, k, n_stage, train, graph, tie_free)
    errs = []
    ef = float((y1 - y0).abs().max())
    if ef > 1e-4: errs.append("fwd %.3g" % ef)
    if tie_free:
        pairs = [("gx", gx1, gx0)] + [("gw%d" % i, s1[0].weight.grad.cpu(), s0[0].weight.grad) for i, (s0, s1) in enumerate(zip(st0, st1))] + \
                [("gbnw%d" % i, s1[1].weight.grad.cpu(), s0[1].weight.grad) for i, (s0, s1) in enumerate(zip(st0, st1))]
        for name, a, b in pairs:
            tol = 1e-4 * max(1e-6, float(b.abs().max()))
            e = float((a - b).abs().max())
            if e > tol: errs.append("%s %.3g > %.3g" % (name, e, tol))
    else:
        skipped += 1
    if errs:
        bad += 1
        print("MISMATCH", what, errs, flush=True)
print("%d cases, %d mismatching, %d forward-only (near-ties)" % (n_cases, bad, skipped))
