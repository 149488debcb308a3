"""dev helper: rows_wgrad against the library GEMM for the head's weight-gradient shapes."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ahn3_dgcnn_pytorch_amd import ops
dev = torch.device("cuda:0")
P = 56000


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for M, N in ((256, 512), (512, 192), (192, 192), (64, 64)):
    a = torch.randn(P, M, device=dev)
    b = torch.randn(P, N, device=dev)
    u1, u2 = t(lambda: ops.rows_wgrad(a, b)), t(lambda: torch.mm(a.t(), b))
    err = (ops.rows_wgrad(a, b) - torch.mm(a.t(), b)).abs().max().item()
    print("[%d,P]x[P,%d]: rows_wgrad %7.1f us (%5.1f TF/s)   library %7.1f us   max diff %.2e" % (M, N, u1, 2.0 * P * M * N / u1 / 1e6, u2, err), flush=True)
