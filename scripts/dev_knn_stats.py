"""dev helper: survivor statistics of knn_select16_kernel (library built with -DKNN_STATS: results are garbage)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ahn3_dgcnn_pytorch_amd import ops
dev = torch.device("cuda:0")
x = torch.randn(8, 7000, 64, device=dev, generator=torch.Generator(device=dev).manual_seed(0))
idx = ops.knn_pm(x, 30)
torch.cuda.synchronize()
nw = 8 * 220
st = idx.reshape(-1)[: nw * 4].reshape(nw, 4).float()
print("per wave (32 rows): survivors mean %.1f max %d | batches %.1f | compactions %.2f | stalls/attempt-code max %d" % (
    st[:, 0].mean(), int(st[:, 0].max()), st[:, 1].mean(), st[:, 2].mean(), int(st[:, 3].max())))
print("survivors per row: %.1f" % (st[:, 0].mean() / 32))
