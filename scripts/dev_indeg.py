"""dev helper: in-degree distribution of the kNN graph on the bench input (hubness)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ahn3_dgcnn_pytorch_amd import ops
dev = torch.device("cuda:0")
B, N, C, k = 8, 7000, 64, 30
g = torch.Generator().manual_seed(1000)
x = torch.randn(B, N, C, generator=g).to(dev)
idx = ops.knn_pm(x, k).long()
for b in range(2):
    deg = torch.bincount(idx[b].reshape(-1), minlength=N).float()
    q = torch.quantile(deg, torch.tensor([0.5, 0.9, 0.99, 0.999], device=dev))
    print("batch %d: mean %.1f max %d  zero %.1f%%  median %.0f p90 %.0f p99 %.0f p99.9 %.0f  >128: %d  >256: %d  >512: %d" % (
        b, deg.mean(), int(deg.max()), 100 * (deg == 0).float().mean(), *q.tolist(), int((deg > 128).sum()), int((deg > 256).sum()), int((deg > 512).sum())))
