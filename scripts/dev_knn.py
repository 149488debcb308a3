"""dev helper: time the kNN kernel on the GPU box (not part of the test suite)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ahn3_dgcnn_pytorch_amd import ops

dev = torch.device("cuda:0")
for (B, C, N, k) in [(8, 64, 7000, 30), (8, 4, 7000, 30), (8, 64, 7000, 20), (8, 64, 7000, 40), (2, 64, 32768, 64)]:
    x = torch.randn(B, N, C, device=dev)
    for _ in range(3):
        idx = ops.knn_pm(x, k)
    torch.cuda.synchronize()
    t0 = time.time()
    n = 10
    for _ in range(n):
        idx = ops.knn_pm(x, k)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / n
    gf = 2.0 * B * N * N * C / 1e9
    print("knn B=%d C=%d N=%d k=%d: %.3f ms  (%.1f TFLOP/s of distance work)" % (B, C, N, k, dt * 1e3, gf / dt / 1e3), flush=True)
