"""Full DGCNN (3 EdgeConv blocks + the head, all on the HIP kernels of this repository) timings:
BASELINE.json configs[2] (eval forward, B=8 N=7000 k=30) and a synthetic training step (configs[3] shape per GPU).
Not the headline metric (bench.py is); recorded in DESIGN.md."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from ahn3_dgcnn_pytorch_amd import DGCNN

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 7000
k = int(sys.argv[3]) if len(sys.argv) > 3 else 30
mode = sys.argv[4] if len(sys.argv) > 4 else "all"   # all | eval | train (to profile one of them)
m = DGCNN(5, 4, k, dropout=0.5, emb_dims=1024, cuda=True)
g = torch.Generator().manual_seed(0)
x = (torch.rand(B, N, 4, generator=g) * torch.tensor([30.0, 30.0, 20.0, 20.0])).to(dev).permute(0, 2, 1)  # strided, as train.py:176
labels = torch.randint(0, 5, (B, N), generator=g).to(dev)


def timeit(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


if mode in ("all", "eval"):
    m.eval()
    with torch.no_grad():
        t = timeit(lambda: m(x))
    print("eval forward  B=%d N=%d k=%d: %.2f ms  (%.2f M points/s)" % (B, N, k, t * 1e3, B * N / t / 1e6), flush=True)
    with torch.no_grad():
        t3 = timeit(lambda: m(x, depth=3))
    print("  of which the three EdgeConv blocks (depth=3 exit): %.2f ms" % (t3 * 1e3), flush=True)
if mode == "eval":
    sys.exit(0)

m.train()
from ahn3_dgcnn_pytorch_amd import ops
from ahn3_dgcnn_pytorch_amd.optim import FlatAdam
opt = FlatAdam(m.parameters(), lr=1e-3, weight_decay=1e-4)  # train.py:118 on one flat buffer


def step():
    opt.zero_grad()
    loss, _pred = ops.seg_loss(m(x), labels)       # softmax -> cal_loss -> arg-max (train.py:267-271) in one kernel
    loss.backward()
    opt.step()


t = timeit(step)
print("train step (fwd + loss + bwd + Adam)  B=%d N=%d k=%d: %.2f ms  (%.2f M points/s)" % (B, N, k, t * 1e3, B * N / t / 1e6))
