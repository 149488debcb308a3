import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, copy
from test_edgeconv_gpu import make_stages, run_oracle, run_hip
from oracle import knn_exact as ke, torch_ref
from param_fill import cotangent
dev = torch.device("cuda:0")
B, C, N, k, n_stage, train = 1, 64, 40, 32, 2, False
g = torch.Generator().manual_seed(0)
x = torch.randn(B, C, N, generator=g)
idx = ke.knn_exact(x.numpy(), k)
R = cotangent((B, 64, N), tag="edge")
base = make_stages(2 * C, n_stage)
y0, gx0, st0 = run_oracle(x, idx, copy.deepcopy(base), train, R)
y1, gx1, st1 = run_hip(x, idx, base, train, R, dev)
print("fwd max err", (y0 - y1).abs().max().item())
d = (gx0 - gx1).abs()
print("gx max err", d.max().item(), "n>1e-3:", int((d > 1e-3).sum()))
pts = torch.nonzero(d.max(dim=1)[0][0] > 1e-3).flatten().tolist()
print("points with gx diff", pts)
for si, (s0, s1) in enumerate(zip(st0, st1)):
    for nm, a, b in (("w", s0[0].weight.grad, s1[0].weight.grad), ("g", s0[1].weight.grad, s1[1].weight.grad), ("b", s0[1].bias.grad, s1[1].bias.grad)):
        print("stage", si, nm, "err", (a - b.cpu()).abs().max().item(), "scale", a.abs().max().item())
# analyse ties in the oracle's pre-max activations (fp32 CPU)
for st in base: st.eval()
with torch.no_grad():
    e = torch_ref.edge_features(x, k=k, idx=torch.from_numpy(idx.astype(np.int64)))
    for st in base: e = st(e)
    top2 = e.topk(2, dim=-1)[0]
    gap = (top2[..., 0] - top2[..., 1])
    print("min top-2 gap", gap.min().item(), "n gaps==0:", int((gap == 0).sum()), "n<1e-6:", int((gap < 1e-6).sum()))
    z = torch.nonzero(gap[0] == 0)
    print("tie locations (channel, point):", z[:10].tolist())
    for c, p in z[:3].tolist():
        print(c, p, e[0, c, p].tolist())
