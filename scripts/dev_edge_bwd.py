"""dev helper: time one EdgeConv block fwd+bwd (single stage), print per-phase wall times."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from ahn3_dgcnn_pytorch_amd import ops
dev = torch.device("cuda:0")
B, N, C, k = 8, 7000, 64, 30
st = nn.Sequential(nn.Conv2d(2 * C, 64, 1, bias=False), nn.BatchNorm2d(64), nn.LeakyReLU(0.2)).to(dev).train()
x = torch.randn(B, N, C, device=dev, requires_grad=True)
idx = ops.knn_pm(x, k)
cot = torch.randn(B, N, 64, device=dev)
def step():
    x.grad = None
    y = ops.edgeconv_pm(x, idx, st[0].weight, st[1], training=True)
    y.backward(cot)
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.time()
for _ in range(20): step()
torch.cuda.synchronize(); print("edgeconv fwd+bwd (no knn): %.3f ms" % ((time.time() - t0) / 20 * 1e3))
