"""dev helper: dgcnn_rows_gemm against torch.mm (hipBLASLt fp32) on the head's shapes at B=8, N=7000."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ahn3_dgcnn_pytorch_amd import ops
dev = torch.device("cuda:0")
P = int(sys.argv[1]) if len(sys.argv) > 1 else 56000


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for K, N, what in ((192, 1024, "conv6 fwd"), (192, 512, "conv7 fwd"), (512, 256, "conv8 fwd"), (256, 512, "conv8 dgrad"),
                   (512, 192, "conv7 dgrad"), (192, 192, "cat . M")):
    x = torch.randn(P, K, device=dev)
    w = torch.randn(N, K, device=dev)
    a = t(lambda: ops.rows_gemm(x, w))
    b = t(lambda: torch.mm(x, w.t()))
    fl = 2.0 * P * K * N
    print("%-12s P=%d K=%4d N=%4d  rows_gemm %7.1f us (%5.1f TF/s)   torch.mm %7.1f us (%5.1f TF/s)" % (
        what, P, K, N, a, fl / a / 1e6, b, fl / b / 1e6), flush=True)

for M, N, what in ((256, 512, "dW8"), (512, 192, "dW7c"), (192, 192, "cat^T cat")):
    a = torch.randn(P, M, device=dev) * 1e-5
    b = torch.randn(P, N, device=dev)
    t16 = t(lambda: ops.rows_wgrad(a, b))
    os.environ["DGCNN_WGRAD_F32"] = "1"
    fl = 2.0 * P * M * N
    print("%-12s P=%d M=%4d N=%4d  rows_wgrad %7.1f us (%5.1f TF/s)" % (what, P, M, N, t16, fl / t16 / 1e6), flush=True)
