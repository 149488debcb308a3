"""dev helper: host-side cost of one bench step (enqueue only) against its GPU time."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import bench
from ahn3_dgcnn_pytorch_amd import ops
from ahn3_dgcnn_pytorch_amd.parallel import FlatGradAllReduce
from param_fill import cotangent
dev = torch.device("cuda:0")
B, N, k, C = 8, 7000, 30, 64
stages = bench.make_stages(C, 1, dev)
params = [p for st in stages for p in st.parameters()]
sync = FlatGradAllReduce(params)
x_pm = torch.randn(B, N, C, generator=torch.Generator().manual_seed(1000)).to(dev).requires_grad_(True)
cot = cotangent((B, 64, N), tag="bench").to(dev).transpose(1, 2).contiguous()

def step():
    sync.zero()
    x_pm.grad = None
    idx = ops.knn_pm(x_pm, k)
    y = ops.edgeconv_pm(x_pm, idx, stages[0][0].weight, stages[0][1], None, None, training=True, slope=0.2)
    y.backward(cot)
    sync.sync()

for _ in range(5): step()
torch.cuda.synchronize()
n = 200
t0 = time.perf_counter()
for _ in range(n): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.3f ms/step; with final sync %.3f ms/step" % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(100): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
