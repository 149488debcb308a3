"""dev helper: survivor-list statistics of the fp16 filter (reads lcount / rbflag out of the kNN workspace)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ahn3_dgcnn_pytorch_amd import _lib
dev = torch.device("cuda:0")
B, N, C, k = 8, 7000, int(sys.argv[1]) if len(sys.argv) > 1 else 64, 30
x = torch.randn(B, N, C, generator=torch.Generator().manual_seed(1000)).to(dev)  # the bench input
lib = _lib.load()
idx = torch.empty((B, N, k), dtype=torch.int32, device=dev)
nb = lib.dgcnn_knn_workspace_bytes(B, C, N, k)
ws = torch.zeros((nb,), dtype=torch.uint8, device=dev)
_lib.check(lib.dgcnn_knn(_lib.ptr(x), B, N, C, k, _lib.ptr(idx), _lib.ptr(ws), nb, _lib.stream_ptr(dev)), "knn")
torch.cuda.synchronize()
npad = (N + 127) // 128 * 128
cp = 4 if C <= 4 else 8 if C <= 8 else (C + 15) // 16 * 16
al = lambda v: (v + 255) // 256 * 256
off = al(B * npad * cp * 4) + 4 * al(B * npad * 4) + al(B * cp * 4) + al(B * npad * cp * 2)
nrb = B * (npad // 32)
lcount = ws[off:off + nrb * 16].view(torch.int32).reshape(nrb, 4).sum(1)
off += al(nrb * 16) + al(nrb * 8)
rbflag = ws[off:off + nrb * 4].view(torch.int32)
print("row blocks", nrb, "flagged", int(rbflag.sum()), "lcount mean %.1f max %d (per row %.1f)" % (
    lcount.float().mean(), int(lcount.max()), lcount.float().mean() / 32))
print("flagged blocks:", torch.nonzero(rbflag).flatten()[:20].tolist())
