"""dev helper: a few launches of the kNN kernel at the bench shape (for rocprofv3 --pmc runs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ahn3_dgcnn_pytorch_amd import ops
dev = torch.device("cuda:0")
C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
x = torch.randn(8, 7000, C, device=dev, generator=torch.Generator(device=dev).manual_seed(0))
for _ in range(4):
    idx = ops.knn_pm(x, 30)
torch.cuda.synchronize()
print("done", int(idx.sum()))
