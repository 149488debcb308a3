"""dev helper: the model-level training step with the reverse lists prepared by riders in the forward (default) against the same
step with DGCNN_NO_FWD_LISTS=1 (the backward builds them): logits and all 25 gradients must be bit-identical -- the lists are
sorted by edge id whoever builds them.  Random shapes around the mode boundaries (B = 15 / 16: riders for all three passes /
counting pass only; N up to 8192: the riders' tables live in 32 KB of LDS).   usage: fuzz_model_lists.py [seed] [cases]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from param_fill import fill_module, cotangent
from ahn3_dgcnn_pytorch_amd import DGCNN

dev = torch.device("cuda:0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bad = 0
for it in range(n_cases):
    B = int(rng.choice([1, 2, 3, 5, 15, 16, 17]))
    N = int(rng.choice([128, 131, 200, 333, 512, 1000])) if B > 3 else int(rng.choice([128, 300, 1000, 4096, 8192, 8200]))
    F = int(rng.choice([1, 3, 4, 6, 8]))
    k = int(rng.choice([1, 5, 20, 32, 33, 40, 64]))
    k = min(k, N)
    g = torch.Generator().manual_seed(1000 + it)
    x = (torch.rand(B, N, F, generator=g) * 20.0).permute(0, 2, 1).to(dev)
    m = DGCNN(5, F, k, dropout=0.0, emb_dims=1024, cuda=True)
    fill_module(m)
    m.train()
    R = cotangent((B, 5, N), tag="fuzzlists%d" % it).to(dev)

    def run():
        m.zero_grad(set_to_none=True)
        y, used = m(x, return_idx=True)
        (y * R).sum().backward()
        torch.cuda.synchronize()
        return y.detach().clone(), {n: p.grad.detach().clone() for n, p in m.named_parameters()}

    os.environ.pop("DGCNN_NO_FWD_LISTS", None)
    y1, g1 = run()
    os.environ["DGCNN_NO_FWD_LISTS"] = "1"
    y0, g0 = run()
    os.environ.pop("DGCNN_NO_FWD_LISTS", None)
    what = "case %d: B=%d N=%d F=%d k=%d" % (it, B, N, F, k)
    errs = [] if torch.equal(y1, y0) else ["logits"]
    errs += [n for n in g1 if not torch.equal(g1[n], g0[n])]
    if errs:
        bad += 1
        print("MISMATCH", what, errs[:5], flush=True)
    else:
        print("ok", what, flush=True)
print("%d cases, %d mismatching" % (n_cases, bad))
