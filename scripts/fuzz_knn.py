"""dev helper: randomised GPU-vs-oracle comparison of the kNN index lists (bit-exact gate) over shapes, scales,
offsets, quantisation (exact ties) and cluster structure.  Prints every mismatching configuration.

    python scripts/fuzz_knn.py [seed] [cases] [channel counts, e.g. 1,2,3,4]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import knn_exact as ke
from ahn3_dgcnn_pytorch_amd import ops

dev = torch.device("cuda:0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 80
channels = [int(c) for c in sys.argv[3].split(",")] if len(sys.argv) > 3 else [3, 4, 5, 8, 9, 16, 24, 32, 64, 100, 128]
bad = 0
t0 = time.time()
for it in range(n_cases):
    C = int(rng.choice(channels))
    N = int(rng.choice([257, 300, 512, 777, 1000, 1500, 2048, 3000]))
    B = int(rng.choice([1, 2, 3]))
    k = int(rng.choice([1, 8, 20, 30, 32, 33, 40, 41, 64])); k = min(k, N)
    x = rng.standard_normal((B, C, N)).astype(np.float32)
    mode = rng.choice(["plain", "scale", "offset", "quant", "cluster", "sparse", "dups", "chanscale"])
    if mode == "scale": x *= np.float32(10.0 ** rng.uniform(-4, 4))
    elif mode == "offset": x += np.float32(10.0 ** rng.uniform(0, 4))
    elif mode == "quant": x = np.round(x * rng.uniform(0.3, 2.0)).astype(np.float32) * np.float32(rng.choice([0.25, 0.5, 1.0]))
    elif mode == "cluster": x = x * np.float32(10.0 ** rng.uniform(-3, -1)) + rng.integers(0, rng.integers(2, 6), size=(B, 1, N)).astype(np.float32) * np.float32(rng.uniform(1, 10))
    elif mode == "sparse": x = (np.abs(x) > rng.uniform(1.0, 2.2)).astype(np.float32) * np.float32(rng.uniform(0.5, 3))
    elif mode == "dups": x[:, :, N // 2:] = x[:, :, : N - N // 2]
    elif mode == "chanscale": x *= (10.0 ** rng.uniform(-2, 2, size=(1, C, 1))).astype(np.float32)
    x = np.ascontiguousarray(x, dtype=np.float32)
    want = ke.knn_exact(x, k)
    got = ops.knn_pm(ops.to_point_major(torch.from_numpy(x).to(dev)), k).cpu().numpy()
    nbad = int((got != want).sum())
    if nbad:
        bad += 1
        print("MISMATCH case %d: B=%d C=%d N=%d k=%d mode=%s -> %d elements" % (it, B, C, N, k, mode, nbad), flush=True)
print("%d cases, %d with mismatches, %.1f s" % (n_cases, bad, time.time() - t0))
