"""bench.py -- EdgeConv fwd+bwd throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of synthetic point features that is already
resident in HBM: kNN graph build in 64-d feature space -> EdgeConv block (conv 128->64 + BatchNorm
(batch statistics) + LeakyReLU(0.2) [+ conv 64->64 + BN + LReLU with --stages 2] -> max over k)
forward, then its backward for a fixed cotangent (gradients w.r.t. the features and every
parameter).  Workload = BASELINE.json configs[1]: B=8, N=7000, k=30, C=64 per GPU.  With N GPUs
every rank runs that workload on its own shard (weak scaling: the batch elements are independent
units) and the step ends with one RCCL all-reduce of the flat parameter-gradient buffer.

Rank 0 prints ONE JSON line (the contract of the build prompt) with two extra objects:
  roofline     -- the dominant kernel (knn_kernel): algorithmic FLOPs / its HIP-event time, against
                  the fp32 MFMA peak of MI355X_MICROARCH.md
  cpu_baseline -- oracle/torch_ref.py (the reference's own PyTorch-CPU op sequence) timed on the
                  host cores of the same box on a bounded sample of the same workload
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md, "Peak FP32 (matrix)"


def make_edge_stage(cin, cout=64):
    """parameter container of one edge-MLP stage: Conv2d 1x1 (no bias) + BatchNorm2d + LeakyReLU(0.2)
    (reference model.py:116-130)"""
    import torch.nn as nn
    return nn.Sequential(nn.Conv2d(cin, cout, kernel_size=1, bias=False), nn.BatchNorm2d(cout),
                         nn.LeakyReLU(negative_slope=0.2))


def make_stages(C, n_stage, device):
    from param_fill import fill_tensor
    stages = [make_edge_stage(2 * C)] + ([make_edge_stage(64)] if n_stage == 2 else [])
    for i, st in enumerate(stages):
        st.load_state_dict({n: fill_tensor("bench%d.%s" % (i, n), v) for n, v in st.state_dict().items()})
        st.train()
        st.to(device)
    return stages


def cpu_baseline(B, N, k, C, n_stage, steps=2):
    """The reference's CPU path (restated op-for-op in oracle/torch_ref.py: materialised BxNxN
    distances, topk, gather/repeat/cat, Conv2d, BatchNorm2d, LeakyReLU, max) fwd+bwd on the host."""
    from oracle import torch_ref
    from param_fill import cotangent
    nthreads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    torch.set_num_threads(nthreads)
    stages = make_stages(C, n_stage, torch.device("cpu"))
    x = torch.randn(B, C, N, generator=torch.Generator().manual_seed(0))
    R = cotangent((B, 64, N), tag="bench")

    def step():
        xr = x.clone().requires_grad_(True)
        y = torch_ref.edgeconv_block(xr, k, stages)
        (y * R).sum().backward()

    step()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    dt = time.perf_counter() - t0
    return {"value": B * N * steps / dt, "unit": "points/s", "cores": nthreads, "kind": "port",
            "sample": "B=%d of the workload's 8 batch elements (same N=%d, k=%d, C=%d, %d-stage block), "
                      "%d fwd+bwd steps after 1 warm-up, %.1f s" % (B, N, k, C, n_stage, steps, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--stages", type=int, default=1, choices=(1, 2),
                    help="1: conv5-shaped block (128->64); 2: conv3+conv4-shaped block (128->64->64)")
    ap.add_argument("--batch", type=int, default=8, help="batch elements per GPU")
    ap.add_argument("--points", type=int, default=7000)
    ap.add_argument("--k", type=int, default=30)
    ap.add_argument("--channels", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-batch", type=int, default=4)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the EdgeConv path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    from ahn3_dgcnn_pytorch_amd import ops
    from ahn3_dgcnn_pytorch_amd.parallel import FlatGradAllReduce
    from param_fill import cotangent

    B, N, k, C = args.batch, args.points, args.k, args.channels
    stages = make_stages(C, args.stages, dev)
    params = [p for st in stages for p in st.parameters()]
    sync = FlatGradAllReduce(params)
    g = torch.Generator().manual_seed(1000 + rank)
    x_pm = torch.randn(B, N, C, generator=g).to(dev).requires_grad_(True)   # resident in HBM
    cot = cotangent((B, 64, N), tag="bench").to(dev).transpose(1, 2).contiguous()  # [B, N, 64]
    w2 = stages[1][0].weight if args.stages == 2 else None
    bn2 = stages[1][1] if args.stages == 2 else None
    knn_events = []

    def step(timed):
        sync.zero()
        x_pm.grad = None
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        idx = ops.knn_pm(x_pm, k)
        if timed:
            e1.record()
            knn_events.append((e0, e1))
        y = ops.edgeconv_pm(x_pm, idx, stages[0][0].weight, stages[0][1], w2, bn2, training=True, slope=0.2)
        y.backward(cot)
        sync.sync()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    knn_ms = sum(a.elapsed_time(b) for a, b in knn_events) / max(1, len(knn_events))
    if rank == 0:
        flops = 2.0 * B * N * N * C  # FLOPs_kNN = 2 B N^2 C per launch
        achieved = flops / (knn_ms * 1e-3) / 1e12
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "knn_traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as f:
                traffic = json.load(f).get("hbm_bytes_per_launch")
        out = {
            "metric": "EdgeConv fwd+bwd points/sec (B=8,N=7000,k=30)",
            "value": world * B * N * args.steps / dt,
            "unit": "points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "single EdgeConv block fwd+bwd (kNN graph build + conv 128->64%s + BN(train) + "
                                   "LeakyReLU + max-over-k), BASELINE.json configs[1]" %
                                   (" + conv 64->64 + BN + LeakyReLU" if args.stages == 2 else ""),
                       "batch_per_gpu": B, "points": N, "k": k, "channels": C, "stages": args.stages,
                       "parallelism": "dp%d (batch elements sharded, one all-reduce of the flat grad buffer)" % world},
            "roofline": {"bound": "mfma", "kernel": "knn_kernel (fused fp32-MFMA distances + streaming top-k)",
                         "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": traffic,
                         "kernel_ms": knn_ms, "algorithmic_flops_per_launch": flops},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(min(args.cpu_sample_batch, B), N, k, C, args.stages)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
