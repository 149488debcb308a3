"""dev helper: the CPU baseline (oracle/torch_ref.py EdgeConv block fwd+bwd) at several thread counts."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import bench
print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count())
for nt in (8, 16, 32, 64, 128):
    os.sched_setaffinity(0, set(list(os.sched_getaffinity(0))[:max(nt, 1)])) if False else None
    torch.set_num_threads(nt)
    from oracle import torch_ref
    from param_fill import cotangent
    stages = bench.make_stages(64, 1, torch.device("cpu"))
    x = torch.randn(2, 64, 7000, generator=torch.Generator().manual_seed(0))
    R = cotangent((2, 64, 7000), tag="bench")
    def step():
        xr = x.clone().requires_grad_(True)
        y = torch_ref.edgeconv_block(xr, 30, stages)
        (y * R).sum().backward()
    step()
    t0 = time.perf_counter(); step(); dt = time.perf_counter() - t0
    print("threads %3d: %.2f s per step of B=2 -> %.0f points/s" % (nt, dt, 2 * 7000 / dt), flush=True)
