"""Times the head's tail (dgcnn_head_tail_fwd / _bwd: conv7 .. conv9 with their BatchNorms, LeakyReLUs and dropout) at a
training step's shape.   usage: bench_head_tail.py [B] [N] [iters]     (under rocprofv3 --kernel-trace --stats for the kernels)"""
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ahn3_dgcnn_pytorch_amd import ops  # noqa: E402
from test_head_tail_gpu import make_layers  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 7000
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device("cuda:0")
P = B * N
w7, w8, w9, bn7, bn8 = make_layers(0)
bn7, bn8 = bn7.to(dev).train(), bn8.to(dev).train()
g = torch.Generator().manual_seed(1)
cat = torch.randn(P, 192, generator=g).to(dev).requires_grad_(True)
add7 = torch.randn(B, 512, generator=g).to(dev).requires_grad_(True)
w7, w8, w9 = [t.to(dev).requires_grad_(True) for t in (w7, w8, w9)]
cot = torch.randn(P, 5, generator=g).to(dev) / P
amax = ops.rows_absmax(cat.detach())


def step():
    for t in (cat, add7, w7, w8, w9, bn7.weight, bn7.bias, bn8.weight, bn8.bias):
        t.grad = None
    y = ops.head_tail_rows(cat, add7, w7[:, 1024:], bn7, w8, bn8, w9, B, N, True, 0.2, 0.5, amax)
    y.backward(cot)


for _ in range(3):
    step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    step()
e1.record()
torch.cuda.synchronize()
print("head tail fwd+bwd B=%d N=%d: %.3f ms per step" % (B, N, e0.elapsed_time(e1) / iters))
