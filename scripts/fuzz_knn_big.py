import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from oracle import knn_exact as ke
from ahn3_dgcnn_pytorch_amd import ops
dev = torch.device("cuda:0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 77)
bad = 0
for it in range(24):
    C = int(rng.choice([4, 9, 64, 64, 64, 128])); N = int(rng.choice([4097, 5000, 7000, 7001, 8192, 9000]))
    B = int(rng.choice([1, 3, 5, 8, 9])); k = int(rng.choice([20, 30, 32, 40, 50, 64]))
    x = rng.standard_normal((B, C, N)).astype(np.float32)
    mode = str(rng.choice(["plain", "cluster", "quant", "offset"]))
    if mode == "cluster": x = x * np.float32(0.02) + rng.integers(0, 4, size=(B, 1, N)).astype(np.float32) * np.float32(3.0)
    elif mode == "quant": x = np.round(x).astype(np.float32) * np.float32(0.5)
    elif mode == "offset": x += np.float32(100.0)
    x = np.ascontiguousarray(x)
    want = ke.knn_exact(x, k)
    got = ops.knn_pm(ops.to_point_major(torch.from_numpy(x).to(dev)), k).cpu().numpy()
    n = int((got != want).sum())
    print("case %d B=%d C=%d N=%d k=%d %s -> %s" % (it, B, C, N, k, mode, "ok" if n == 0 else "MISMATCH %d" % n), flush=True)
    bad += n > 0
print("bad", bad)
