"""bench.py -- EdgeConv fwd+bwd throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload block|train]

`--gpus N` with N > 1 and no launcher environment: this process starts the N ranks itself (one process per GPU, as a
`python -m torch.distributed.run` child, BEFORE it touches the GPU) and relays their output.  Under the driver's own
launcher (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...`) it is one of the
ranks.  It never falls back to fewer ranks: WORLD_SIZE != --gpus exits non-zero.

Workloads
  block (default; BASELINE.json configs[1], the configuration the metric is quoted on): one step = kNN graph build in
        64-d feature space -> EdgeConv block (conv 128->64 + BatchNorm(batch statistics) + LeakyReLU(0.2) [+ conv
        64->64 + BN + LReLU with --stages 2] -> max over k) forward, then its backward for a fixed cotangent, on
        B=8, N=7000, k=30, C=64 per GPU, resident in HBM.  With N GPUs every rank runs that workload on its own shard
        (weak scaling: batch elements are independent units) and the step ends with one RCCL all-reduce of the flat
        parameter-gradient buffer.  The same line carries `train_step`: configs[3] measured in the same run (below).
  train (BASELINE.json configs[3]): one step = the reference's training step on synthetic blocks -- float64 [B, N, 4]
        points -> fp32 (train.py:175-176) -> full DGCNN forward (model.py:144-193) -> softmax + cal_loss + arg-max
        (train.py:180,197,201) -> backward -> ONE RCCL all-reduce of the 980 864-float gradient buffer -> Adam step
        (train.py:118,199); global B=32 sharded B/N per GPU (strong scaling), N=7000, k=30, dropout 0.5.

Rank 0 prints ONE JSON line (the contract of the build prompt) with
  roofline     -- the dominant kernel (largest of the profiled kernels): algorithmic flops or bytes / its HIP-event time
                  (events recorded inside the C ABI on the launch stream).  `bound` is "mfma" (vs the dense fp16 or fp32
                  MFMA peak), "hbm" (vs 8 TB/s) or "l2-gather" (row gathers served by the XCD's L2: vs the guide's
                  L2-resident gathered-row rate); gather kernels also carry `hbm_frac` = PMC-counter HBM bytes / time /
                  8 TB/s.  The others follow in `kernels`.
  cpu_baseline -- oracle/torch_ref.py (the reference's own PyTorch-CPU op sequence) timed on the host cores of the same
                  box on a bounded sample of the same workload.
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_F16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md, dense fp16/bf16 MFMA peak
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md, "HBM3E peak BW" (spec; ~6.3 TB/s achievable)
PEAK_L2_GATHER_GBS = 18800.0   # MI355X_MICROARCH.md, "Indexed rows": rows shared by every workgroup, served by the
                               # XCD's L2: 16.8-18.8 TB/s chip-wide
                               # (scripts/microbench/row_gather.hip reaches 20-28 TB/s for random 256-byte rows of a
                               # 1.8 MB table per XCD with waves that stream thousands of rows; the three gather passes of
                               # the step run at 60-77 G L2 requests/s (6.5-7.8 TB/s of rows) and scale with their
                               # request count: DESIGN.md section 5, profiles/r2_60_l2_requests.txt)
TRAFFIC_FILE = os.path.join("profiles", "hbm_traffic.json")


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", choices=("block", "train"), default="block")
    ap.add_argument("--profile-every", type=int, default=0,
                    help="record the per-kernel HIP events on every n-th timed step (an event record costs GPU time: some "
                         "twenty records per sampled step); 0 = one (up to 30 steps) or three sampled steps, spread over the timed region")
    ap.add_argument("--stages", type=int, default=1, choices=(1, 2),
                    help="block workload: 1 = conv5-shaped block (128->64); 2 = conv3+conv4-shaped block (128->64->64)")
    ap.add_argument("--batch", type=int, default=None,
                    help="block: batch elements PER GPU (default 8); train: GLOBAL batch (default 32)")
    ap.add_argument("--points", type=int, default=7000)
    ap.add_argument("--k", type=int, default=30)
    ap.add_argument("--channels", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train-leg", action="store_true", help="block workload: skip the configs[3] `train_step` object")
    ap.add_argument("--train-steps", type=int, default=20, help="timed steps of the `train_step` leg of the block workload")
    ap.add_argument("--cpu-sample-batch", type=int, default=None)
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="nccl = RCCL over xGMI (the product path); gloo only to rehearse the N-rank code on a box "
                         "with fewer GPUs than ranks (together with --same-device)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--no-overlap", action="store_true", help="train workload, N > 1: one all-reduce after the backward instead of "
                                                             "starting the head's share inside it")
    ap.add_argument("--graph-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-graph-leg", action="store_true", help="block workload: skip the hipGraph replay of the step (a child process)")
    ap.add_argument("--no-model-legs", action="store_true", help="block workload: skip the `eval_forward` (configs[2]) and `stress` (configs[4]) objects")
    ap.add_argument("--no-stress-cpu", action="store_true", help="skip the CPU baseline of the stress leg (B=1, N=32768: tens of GB of host memory)")
    ap.add_argument("--exchange", default="torch", choices=("torch", "abi"),
                    help="the gradient all-reduce: torch.distributed's RCCL binding (default) or the library's own, behind the C "
                         "ABI (dgcnn_comm_allreduce_f32; the id is handed out over the torch.distributed group)")
    return ap.parse_args()


# ----------------------------------------------------------------------------------------------------------------------
def self_launch(args):
    """No launcher environment and --gpus N > 1: start the N ranks as a child `torch.distributed.run` (one process per
    GPU).  Nothing in this process has touched the GPU, and it does not exec: it waits and exits with the child's code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


def init_ranks(args):
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%d: refusing to report a %d-rank number as a %d-GPU one"
              % (args.gpus, world, world, args.gpus), file=sys.stderr)
        raise SystemExit(2)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the EdgeConv path has no CPU fallback)")
    ndev = torch.cuda.device_count()
    dev_index = 0 if args.same_device else local_rank
    if dev_index >= ndev:
        raise SystemExit("bench.py: rank %d needs GPU %d but only %d are visible" % (rank, dev_index, ndev))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ):
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group("gloo")
        if dist.get_world_size() != args.gpus:
            raise SystemExit(2)
    return rank, world, dev


_COMM = None


def make_comm(args, rank, world):
    """--exchange abi: one RCCL communicator behind the C ABI per process (kept for both workloads of a run)"""
    global _COMM
    if args.exchange != "abi":
        return None
    if _COMM is None:
        from ahn3_dgcnn_pytorch_amd.parallel import HipComm
        _COMM = HipComm(rank, world)
    return _COMM


def fence():
    import torch
    import torch.distributed as dist
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
        torch.cuda.synchronize()


def max_over_ranks(value, dev):
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sample_stride(args, steps):
    """every how many timed steps the per-kernel events are recorded"""
    if args.profile_every > 0:
        return args.profile_every
    # one sampled step in a short run, three otherwise: an event record is a barrier packet between two kernels (about 8 us
    # each, 20-odd per sampled step of the block workload: two sampled steps of twenty cost the mean 3 %)
    return max(1, -(-steps // (1 if steps <= 30 else 3)))


def timed_steps(step, n, every, dev, warmup=0):
    """`warmup` untimed steps, then EXACTLY n steps between two fences; `step(sampled)` with sampled=True on every `every`-th.
    The warm-up runs HERE, behind the garbage collection and directly in front of the first fence: with the collection (tens
    of milliseconds of host time) or the creation of the profiling events between warm-up and region, the region started on a
    GPU that had idled -- its first steps ran below the clocks of a loaded chip and a 20-step mean read 5 % above the per-step
    median.  Returns the wall
    time (max over ranks) and this rank's host timeline of the region: `host_enqueue_ms` = the time the host took to
    enqueue the n steps (when it is close to `region_total_ms` the host, not the GPU, paced the region), `region_total_ms`
    = enqueue + the wait of the final fence."""
    import gc
    gc.collect()
    gc.disable()   # (a generation-2 collection inside the region is a 10 ms host stall: twenty 0.5 ms steps then read 1.1 ms)
    marks = []
    debug = bool(os.environ.get("DGCNN_BENCH_DEBUG"))
    try:
        for _ in range(warmup):
            step(False)
        fence()
        t0 = time.perf_counter()
        for i in range(n):
            step(every > 0 and i % every == every // 2)   # (the middle of each stride: never the first step behind the fence)
            if debug:
                marks.append(time.perf_counter() - t0)
        t1 = time.perf_counter() - t0
        fence()
        dt = time.perf_counter() - t0
    finally:
        gc.enable()
    if debug:
        sys.stderr.write("timed region: enqueue %.3f ms, total %.3f ms; host marks (ms): %s\n" % (
            t1 * 1e3, dt * 1e3, " ".join("%.2f" % (m * 1e3) for m in marks)))
    return max_over_ranks(dt, dev), {"host_enqueue_ms": t1 * 1e3, "region_total_ms": dt * 1e3}


def per_step_ms(step, n):
    """a second, separate pass (not the one `value` comes from): every step bracketed by events -> the distribution"""
    import torch
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    torch.cuda.synchronize()
    ev[0].record()
    for i in range(n):
        step(False)
        ev[i + 1].record()
    torch.cuda.synchronize()
    return [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]


def region_times(lib):
    return {lib.dgcnn_profile_region_name(r).decode(): (float(lib.dgcnn_profile_mean_ms(r)), int(lib.dgcnn_profile_samples(r)))
            for r in range(lib.dgcnn_profile_regions())}


def kernel_entries(regions, work, traffic):
    """one roofline entry per profiled kernel that ran; largest first"""
    out = []
    for name, (ms, n) in regions.items():
        if n == 0 or ms <= 0 or name not in work:
            continue
        bound, amount, peak, note = work[name]
        flops = bound == "mfma"
        ach = amount / (ms * 1e-3) / (1e12 if flops else 1e9)
        e = {"kernel": name, "bound": bound, "achieved": ach, "peak": peak, "unit": "TFLOP/s" if flops else "GB/s",
             "frac": ach / peak, "traffic": None, "kernel_ms": ms, "samples": n,
             "algorithmic_%s_per_launch" % ("flops" if flops else "bytes"): amount, "note": note}
        t = traffic.get(name)
        if t is not None:
            e["traffic"] = t
            e["traffic_source"] = TRAFFIC_FILE + " (rocprofv3 --pmc passes of this workload, committed; not re-measured in this run)"
            e["hbm_frac"] = t / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS
        out.append(e)
    out.sort(key=lambda d: -d["kernel_ms"])
    return out


def load_traffic(key):
    path = os.path.join(ROOT, TRAFFIC_FILE)
    if not os.path.exists(path):
        return {}
    with open(path) as f:
        return json.load(f).get(key, {})


# ----------------------------------------------------------------------------------------------------------------------
# workload "block": BASELINE.json configs[1]
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


REF_THREADS = 8   # the reference's own default: torch.set_num_threads(8), train.py:36


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or "unknown"


def cpu_threads(want=None):
    import torch
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # measured on the MI355X box's host (256 hardware threads): 8 / 16 / 32 / 64 / 128 / 256 threads give
    # 20.6 / 20.4 / 25.1 / 19.0 / 12.0 / 3.9 k points/s on the block, so the baseline's `value` is taken at its best
    # setting, 32; the reference's own default of 8 (train.py:36) is reported beside it
    n = min(avail, want or 32)
    torch.set_num_threads(n)
    return n, avail


def cpu_timed(step, steps):
    step()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    return time.perf_counter() - t0


def cpu_both_thread_counts(step, steps, units_per_step):
    """`step` timed at the best thread count (-> value, cores) and at the reference's default of 8 threads"""
    import torch
    best, avail = cpu_threads()
    dt_best = cpu_timed(step, steps)
    ref, _ = cpu_threads(REF_THREADS)
    dt_ref = dt_best if ref == best else cpu_timed(step, steps)
    cpu_threads()
    return {"value": units_per_step * steps / dt_best, "unit": "points/s", "cores": best, "kind": "port",
            "cpu_model": cpu_model(), "hardware_threads": avail,
            "value_at_reference_threads": units_per_step * steps / dt_ref, "reference_threads": ref,
            "torch": torch.__version__, "seconds": dt_best + (0.0 if ref == best else dt_ref)}


def cpu_baseline_block(B, N, k, C, n_stage, steps=2):
    """The reference's CPU path (restated op-for-op in oracle/torch_ref.py: materialised BxNxN
    distances, topk, gather/repeat/cat, Conv2d, BatchNorm2d, LeakyReLU, max) fwd+bwd on the host."""
    import torch
    from oracle import torch_ref
    from param_fill import cotangent
    stages = make_stages(C, n_stage, torch.device("cpu"))
    x = torch.randn(B, C, N, generator=torch.Generator().manual_seed(0))
    R = cotangent((B, 64, N), tag="bench")

    def step():
        xr = x.clone().requires_grad_(True)
        y = torch_ref.edgeconv_block(xr, k, stages)
        (y * R).sum().backward()

    out = cpu_both_thread_counts(step, steps, B * N)
    out["sample"] = ("B=%d of the workload's 8 batch elements (same N=%d, k=%d, C=%d, %d-stage block), %d fwd+bwd steps after 1 "
                     "warm-up at each thread count (%d = best measured on this host class, %d = the reference's default, "
                     "train.py:36), %.1f s in all" % (B, N, k, C, n_stage, steps, out["cores"], out["reference_threads"], out["seconds"]))
    return out


def run_block(args, rank, world, dev):
    import torch
    from ahn3_dgcnn_pytorch_amd import _lib, ops
    from ahn3_dgcnn_pytorch_amd.parallel import FlatGradAllReduce
    from param_fill import cotangent

    B, N, k, C = args.batch or 8, args.points, args.k, args.channels
    stages = make_stages(C, args.stages, dev)
    params = [p for st in stages for p in st.parameters()]
    sync = FlatGradAllReduce(params, comm=make_comm(args, rank, world))
    g = torch.Generator().manual_seed(1000 + rank)
    x_pm = torch.randn(B, N, C, generator=g).to(dev).requires_grad_(True)   # resident in HBM
    cot = cotangent((B, 64, N), tag="bench").to(dev).transpose(1, 2).contiguous()  # [B, N, 64]
    w2 = stages[1][0].weight if args.stages == 2 else None
    bn2 = stages[1][1] if args.stages == 2 else None
    knn_events = []
    lib = _lib.load()

    def step(sampled):
        sync.zero()
        x_pm.grad = None
        if sampled:
            lib.dgcnn_profile_resume()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        idx = ops.knn_pm(x_pm, k)
        if sampled:
            e1.record()
            knn_events.append((e0, e1))
        y = ops.edgeconv_pm(x_pm, idx, stages[0][0].weight, stages[0][1], w2, bn2, training=True, slope=0.2)
        y.backward(cot)
        sync.sync()
        if sampled:
            lib.dgcnn_profile_enable(0)

    lib.dgcnn_profile_enable(1)  # create the events, reset the sample rings ...
    lib.dgcnn_profile_enable(0)  # ... recording is switched on for the sampled steps only
    step(False)                  # (first call: code objects, allocator pools; NOT one of the counted warm-up steps)
    dt, host = timed_steps(step, args.steps, sample_stride(args, args.steps), dev, args.warmup)
    knn_ms = sum(a.elapsed_time(b) for a, b in knn_events) / max(1, len(knn_events))
    regions = region_times(lib)
    dist_ms = per_step_ms(step, min(args.steps, 50))
    if os.environ.get("DGCNN_BENCH_DEBUG"):
        sys.stderr.write("per-step ms (separate pass): %s\n" % " ".join("%.3f" % v for v in dist_ms[:12]))
    if rank != 0:
        return None
    P, E = B * N, B * N * k
    knn_flops = 2.0 * B * N * N * C  # FLOPs_kNN = 2 B N^2 C
    conv2 = 2.0 * E * 64 * 64
    # Algorithmic work of one launch of each large kernel.
    work = {
        "knn_bound16s_kernel": ("mfma", knn_flops, PEAK_F16_MFMA_TFLOPS,
                                "all B N^2 pairs once on fp16 MFMA (threshold pre-pass), 2 B N^2 C flops; the kernel executes "
                                "(C + 16) / C of that: one extra 16-wide K slab carries the norms, the threshold and the per-pair "
                                "slack, so that the accumulator is the tested quantity (DESIGN.md section 5)"),
        "knn_filter16_kernel": ("mfma", knn_flops, PEAK_F16_MFMA_TFLOPS,
                                "all B N^2 pairs once on fp16 MFMA (exact filter), 2 B N^2 C flops; executed (C + 16) / C of that "
                                "(augmented K slab, as the pre-pass)"),
        "knn_refine_kernel": ("l2-gather", E * (4.0 * C + 4 + 4 + 4), PEAK_L2_GATHER_GBS,
                              "exact fp32 distance of at least the k winners of every row: one C-float row, its norm and "
                              "its list entry per winner, the index written; rows come from L2 (latency / VALU chain bound)"),
        "edge_reduce_kernel": ("l2-gather", E * (256.0 + 4) + P * (256.0 * 3 + 64), PEAK_L2_GATHER_GBS,
                               "per edge one fp32 [64] row of TP (1.8 MB table per batch element = one XCD's L2) and its "
                               "index; per point TQ read, usel, usum, hitj written"),
        "edge_rgather_kernel": ("l2-gather", (E * (256.0 + 4 + 32) if args.stages == 1 else E * (512.0 + 4)) + P * (8.0 + 512),
                                PEAK_L2_GATHER_GBS,
                                "per in-edge one fp32 [64] row of TQ (two-stage: also one row of dv1), its reverse-list entry "
                                "and (one-stage) its hit record; per point list bounds, TP read, dR written"),
        "edge_mlp_fwd_kernel": ("mfma", conv2, PEAK_F16_MFMA_TFLOPS,
                                "conv 64->64 per edge, 2 E 64 64 flops; executed as 3 split-fp16 products on fp16 MFMA"),
        "edge_mlp_bwd_kernel": ("mfma", 2 * conv2, PEAK_F16_MFMA_TFLOPS,
                                "dgrad + wgrad of conv 64->64 per edge, 2 x 2 E 64 64 flops (the kernel also recomputes the "
                                "forward product); each executed as 3 split-fp16 products on fp16 MFMA"),
    }
    traffic = load_traffic("hbm_bytes_per_launch") if args.stages == 1 and (B, N, k, C) == (8, 7000, 30, 64) else {}
    kernels = kernel_entries(regions, work, traffic)
    out = {
        "metric": "EdgeConv fwd+bwd points/sec (B=8,N=7000,k=30)",
        "value": world * B * N * args.steps / dt,
        "unit": "points/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "ms_per_step_median": statistics.median(dist_ms),
        "ms_per_step_p10_p90": [sorted(dist_ms)[len(dist_ms) // 10], sorted(dist_ms)[(len(dist_ms) * 9) // 10]],
        "host_enqueue_ms": host["host_enqueue_ms"],
        "region_total_ms": host["region_total_ms"],
        "host_timeline_note": "rank 0's host clock over the timed region: the %d steps were enqueued in host_enqueue_ms; "
                              "region_total_ms adds the final fence (= ms_per_step x steps on one rank).  enqueue ~ total means "
                              "the host paced the region, enqueue << total means the GPU did" % args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32 (results); the kNN pre-filter runs on f16 MFMA and is followed by an exact f32 refine "
                 "(indices bit-exact vs the f32 oracle); the first convolution's forward on f32 MFMA, "
                 "its input and weight gradients%s as split-f16 MFMA products (hi.hi + hi.lo + lo.hi, 2^-22 relative)" %
                 (" and the second convolution" if args.stages == 2 else ""),
        "data": "synthetic",
        "config": {"workload": "single EdgeConv block fwd+bwd (kNN graph build + conv 128->64%s + BN(train) + "
                               "LeakyReLU + max-over-k), BASELINE.json configs[1]" %
                               (" + conv 64->64 + BN + LeakyReLU" if args.stages == 2 else ""),
                   "batch_per_gpu": B, "points": N, "k": k, "channels": C, "stages": args.stages,
                   "parallelism": "dp%d (batch elements sharded, one all-reduce of the flat grad buffer)" % world,
                   "launch": "eager", "ranks": world, "backend": args.backend if world > 1 else None,
                   "exchange": args.exchange},
        "roofline": kernels[0] if kernels else None,
        "kernels": kernels[1:],
        "knn": {"ms": knn_ms, "algorithmic_flops": knn_flops, "tflops": knn_flops / (knn_ms * 1e-3) / 1e12 if knn_ms else None,
                "note": "whole dgcnn_knn call (prep + fp16 threshold pass + fp16 filter + exact fp32 refine); the "
                        "distance products run on fp16 MFMA (so this exceeds the 157 TF f32 peak), the result is bit-exact f32"},
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_block(min(args.cpu_sample_batch or 8, B), N, k, C, args.stages)
        out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    return out


# ----------------------------------------------------------------------------------------------------------------------
# workload "train": BASELINE.json configs[3]
def synthetic_blocks(B, N, seed):
    """what collate_pcs (data.py:169-182) hands to the training loop for .npy blocks [N, 5]: float64 points [B, N, 4]
    (block-local x, y in [0, 30) m, z, AGL in [0, 20) m) and float64 labels [B, N] in 0..4"""
    import torch
    g = torch.Generator().manual_seed(seed)
    pts = torch.rand(B, N, 4, generator=g, dtype=torch.float64) * torch.tensor([30.0, 30.0, 20.0, 20.0], dtype=torch.float64)
    lab = torch.randint(0, 5, (B, N), generator=g).double()
    return pts, lab


def cpu_baseline_train(B, N, k, steps=1):
    """the reference's training step on the host: RefDGCNN (model.py:144-193 op for op) forward, softmax + cal_loss
    (train.py:180,197), backward, torch.optim.Adam step (train.py:118,199)"""
    import torch
    from oracle import seg_ref, torch_ref
    torch.manual_seed(0)
    m = torch_ref.RefDGCNN(5, 4, k, dropout=0.5)
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=1e-4)
    pts, lab = synthetic_blocks(B, N, 0)

    def step():
        x = pts.permute(0, 2, 1).float()
        opt.zero_grad()
        loss, _ = seg_ref.step_loss(m(x), lab)
        loss.backward()
        opt.step()

    out = cpu_both_thread_counts(step, steps, B * N)
    out["sample"] = ("B=%d of the workload's blocks (same N=%d, k=%d, full DGCNN training step), %d step(s) after 1 warm-up at each "
                     "thread count (%d and the reference's %d), %.1f s in all" % (
                         B, N, k, steps, out["cores"], out["reference_threads"], out["seconds"]))
    return out


def cpu_baseline_eval(B, N, k, scale_to, steps=1):
    """the reference's inference forward on the host (RefDGCNN.eval(), model.py:144-193), on B batch elements; the
    GPU leg runs `scale_to` of them: batch elements are independent in eval mode, so points/s carries over
    (where the reference formulation does not fit at the full batch, time B=1 and say so)"""
    import torch
    from oracle import torch_ref
    torch.manual_seed(0)
    m = torch_ref.RefDGCNN(5, 4, k, dropout=0.5)
    m.eval()
    pts, _ = synthetic_blocks(B, N, 0)
    x = pts.permute(0, 2, 1).float()

    def step():
        with torch.no_grad():
            m(x)

    out = cpu_both_thread_counts(step, steps, B * N)
    out["sample"] = ("B=%d of the leg's %d batch elements (points/s of B=%d; x%d in batch = the same rate: eval-mode batch elements "
                     "are independent), N=%d, k=%d, full DGCNN eval forward with the materialised B x N x N distances, %d "
                     "forward(s) after 1 warm-up at each thread count (%d and the reference's %d), %.1f s in all" % (
                         B, scale_to, B, scale_to // B, N, k, steps, out["cores"], out["reference_threads"], out["seconds"]))
    return out


def run_train(args, rank, world, dev, steps, warmup, with_cpu, global_batch=None):
    import torch
    import torch.distributed as dist
    from ahn3_dgcnn_pytorch_amd import DGCNN, _lib, blocks, ops
    from ahn3_dgcnn_pytorch_amd.optim import FlatAdam
    from ahn3_dgcnn_pytorch_amd.parallel import FlatGradAllReduce, broadcast, shard_batch

    Bg, N, k = (global_batch or (args.batch if args.workload == "train" and args.batch else 32)), args.points, args.k
    if Bg % world:
        raise SystemExit("bench.py: global batch %d is not divisible by %d ranks" % (Bg, world))
    Bl = Bg // world
    torch.manual_seed(0)                                  # the same initial weights on every rank
    model = DGCNN(5, 4, k, dropout=0.5, emb_dims=1024, cuda=True)
    model.train()
    # (overlap: with more than one rank the all-reduce of conv6 .. conv9 -- 97 % of the bytes -- starts inside the backward,
    # behind the head's part, and runs under the three blocks' backward; torch.distributed exchange only)
    sync = FlatGradAllReduce(model.parameters(), comm=make_comm(args, rank, world), overlap=not args.no_overlap)
    opt = FlatAdam(model.parameters(), lr=1e-3, weight_decay=1e-4, grad_buffer=sync.flat)   # train.py:118
    if dist.is_initialized() and world > 1:
        broadcast(opt.flat, 0)
    pts, lab = synthetic_blocks(Bg, N, 0)
    pts = shard_batch(pts, rank, world).contiguous().to(dev)     # this rank's blocks, resident in HBM
    lab = shard_batch(lab, rank, world).contiguous().to(dev)
    torch.manual_seed(1234 + rank)                        # dropout masks differ per rank
    lib = _lib.load()
    ar_events = []

    def step(sampled):
        if sampled:
            lib.dgcnn_profile_resume()
        x = blocks.to_model_input(pts)                    # float64 -> fp32, [B, 4, N] view (train.py:175-176)
        opt.zero_grad()                                   # train.py:178
        loss, _pred = ops.seg_loss(model(x), lab)         # train.py:179-201 (softmax, cal_loss, arg-max) in one kernel
        loss.backward()                                   # train.py:198
        if sampled:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        sync.sync()                                       # ONE all-reduce of the flat 3.92 MB gradient buffer
        if sampled:
            e1.record()
            ar_events.append((e0, e1))
        opt.step()                                        # train.py:199
        if sampled:
            lib.dgcnn_profile_enable(0)

    lib.dgcnn_profile_enable(1)
    lib.dgcnn_profile_enable(0)
    step(False)                  # (first call: code objects, allocator pools; not one of the counted warm-up steps)
    dt, host = timed_steps(step, steps, sample_stride(args, steps), dev, warmup)
    regions = region_times(lib)
    ar_ms = sum(a.elapsed_time(b) for a, b in ar_events) / max(1, len(ar_events))
    dist_ms = per_step_ms(step, min(steps, 20))
    ranks = dist.get_world_size() if dist.is_initialized() else 1
    if rank != 0:
        return None
    E = Bl * N * k
    conv2 = 2.0 * E * 64 * 64
    work = {
        "edge_mlp_fwd_kernel": ("mfma", conv2, PEAK_F16_MFMA_TFLOPS,
                                "conv2 / conv4 per edge, 2 E 64 64 flops per launch; executed as 3 split-fp16 products"),
        "edge_mlp_bwd_kernel": ("mfma", 2 * conv2, PEAK_F16_MFMA_TFLOPS,
                                "dgrad + wgrad of conv2 / conv4 per edge, 2 x 2 E 64 64 flops per launch (the kernel also "
                                "recomputes the forward product); each executed as 3 split-fp16 products on fp16 MFMA"),
    }
    # the head's products (head_fused.hip): fp32 rows in and out, so their algorithmic work is bytes --
    # what each has to read and write once -- against the HBM peak; conv6 stores nothing and is a matrix product
    Pl, F_, C6, C7, C8 = Bl * N, 192, 1024, 512, 256
    work.update({
        "head_conv6_kernel": ("mfma", 2.0 * Pl * F_ * C6, PEAK_F16_MFMA_TFLOPS,
                              "conv6 + bn6 statistics + max over the points, 2 P 192 1024 flops (3 split-fp16 products executed); "
                              "the [P, 1024] product is never stored"),
        "head_conv7_kernel": ("hbm", 4.0 * Pl * (F_ + C7), PEAK_HBM_GBS, "reads cat [P, 192], writes u7 [P, 512]; BatchNorm-7 sums in the epilogue"),
        "head_conv8_kernel": ("hbm", 4.0 * Pl * (C7 + C8), PEAK_HBM_GBS, "reads u7 (BatchNorm-7 + LeakyReLU at the operand load), writes u8 [P, 256]"),
        "head_dz7_kernel": ("hbm", 4.0 * Pl * (2 * C8 + 2 * C7), PEAK_HBM_GBS,
                            "reads dv8, u8 (du8 formed at the load) and u7, writes dv7 [P, 512]; BatchNorm-7 backward sums in the epilogue"),
        "head_dcat_kernel": ("hbm", 4.0 * Pl * (2 * C7 + F_), PEAK_HBM_GBS, "reads dv7, u7, writes dL/dcat [P, 192]"),
        "head_dw8_kernel": ("hbm", 4.0 * Pl * (2 * C8 + C7), PEAK_HBM_GBS, "reads dv8, u8, u7 once (a contraction over the points: [256, 512] out)"),
        "head_dw7_kernel": ("hbm", 4.0 * Pl * (2 * C7 + F_), PEAK_HBM_GBS, "reads dv7, u7, cat once ([512, 192] out)"),
    })
    head_traffic = {}
    tpath = os.path.join(ROOT, "profiles", "r3_80_head_tail_traffic.json")
    if (Bl, N) == (32, 7000) and os.path.exists(tpath):   # PMC passes of this shape (scripts/pmc_head_tail.sh), committed
        with open(tpath) as f:
            kk = json.load(f)["kernels"]
        for region, pat in (("head_conv7_kernel", "hrow_kernel<1"), ("head_conv8_kernel", "hgemm_kernel<1, 1"),
                            ("head_dz7_kernel", "hgemm_kernel<2, 2"), ("head_dcat_kernel", "hgemm_kernel<2, 0"),
                            ("head_dw8_kernel", "hwgrad_kernel<2, 1"), ("head_dw7_kernel", "hwgrad_kernel<2, 0")):
            for name, e in kk.items():
                if name.startswith(pat):
                    head_traffic[region] = e["hbm_MB"] * 1e6
    kernels = kernel_entries(regions, work, head_traffic)
    for e in kernels:
        if "traffic_source" in e:
            e["traffic_source"] = "profiles/r3_80_head_tail_traffic.json (rocprofv3 --pmc passes of scripts/bench_head_tail.py at this shape, committed; not re-measured in this run)"
    step_flops = 12.04e6 * Bl * N   # 12.04 MFLOP per point for the full training step
    out = {
        "metric": "DGCNN training-step points/sec (global B=%d,N=%d,k=%d)" % (Bg, N, k),
        "value": Bg * N * steps / dt,
        "unit": "points/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": dt / steps * 1e3,
        "ms_per_step_median": statistics.median(dist_ms),
        "host_enqueue_ms": host["host_enqueue_ms"],
        "region_total_ms": host["region_total_ms"],
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32 (results); kNN pre-filter on f16 MFMA (exact f32 refine); conv2/conv4 forward and backward, the "
                 "64-channel blocks' input gradients and the head's products as split-f16 MFMA products (2^-22 relative)",
        "data": "synthetic",
        "config": {"workload": "train.py step on synthetic blocks (f64 -> f32, DGCNN fwd, softmax + cal_loss, bwd, "
                               "all-reduce of the flat gradient buffer, Adam), BASELINE.json configs[3]",
                   "global_batch": Bg, "batch_per_gpu": Bl, "points": N, "k": k, "emb_dims": 1024, "dropout": 0.5,
                   "parallelism": "dp%d (blocks sharded; per-replica BatchNorm statistics)" % world,
                   "backend": (args.backend if world > 1 else None), "launch": "eager", "exchange": args.exchange},
        "rccl_ranks": ranks,
        "allreduce": {"ms": ar_ms, "bytes": sync.flat.numel() * 4, "samples": len(ar_events),
                      "overlapped": bool(sync.overlap and world > 1 and sync.comm is None),
                      "note": "all-reduce (AVG) of %d floats; with `overlapped` the last 951 552 (conv6 .. conv9) are in flight "
                              "since the head's backward and `ms` is what is left of the exchange behind the backward: the "
                              "front part and the wait for the tail" % sync.flat.numel()},
        "step_tflops": step_flops / (dt / steps) / 1e12,
        "roofline": kernels[0] if kernels else None,
        "kernels": kernels[1:],
    }
    if with_cpu:
        out["cpu_baseline"] = cpu_baseline_train(args.cpu_sample_batch or 2, N, k, steps=2)
        out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    return out


# ----------------------------------------------------------------------------------------------------------------------
# legs "eval_forward" (BASELINE.json configs[2]) and "stress" (configs[4]): the full DGCNN on one GPU
def model_work(B, N, k, training):
    """algorithmic work per launch of the profiled kernel groups inside one full-model pass.  A group's launches differ
    in shape (the kNN of block 1 searches 4 channels, blocks 2 and 3 search 64): the entry is the MEAN over the group's
    launches of a pass, so that work / mean time = (sum of work) / (sum of time)."""
    P, E = B * N, B * N * k
    chans = (4, 64, 64)
    knn_flops = sum(2.0 * B * N * N * c for c in chans) / 3
    refine = sum(E * (4.0 * c + 12) for c in chans) / 3
    conv2 = 2.0 * E * 64 * 64
    full_reduce = E * 260.0 + P * 832.0
    sums_only = E * 260.0 + P * 512.0
    F_, C6, C7, C8 = 192, 1024, 512, 256
    return {
        "knn_bound16s_kernel": ("mfma", knn_flops, PEAK_F16_MFMA_TFLOPS,
                                "threshold pre-pass of the three kNN builds (4, 64, 64 channels): mean 2 B N^2 C flops per launch"),
        "knn_filter16_kernel": ("mfma", knn_flops, PEAK_F16_MFMA_TFLOPS,
                                "exact fp16 filter of the three kNN builds: mean 2 B N^2 C flops per launch"),
        "knn_refine_kernel": ("l2-gather", refine, PEAK_L2_GATHER_GBS,
                              "exact fp32 distances of at least the k winners of every row, mean over the three builds"),
        "edge_reduce_kernel": ("l2-gather", (2 * sums_only + full_reduce) / 3 if training else full_reduce, PEAK_L2_GATHER_GBS,
                               "per edge one fp32 [64] row of TP and its index" + (
                                   "; two of the three launches only take the BatchNorm sums (two-stage blocks)" if training else
                                   " (eval: the one-stage block only)")),
        "edge_rgather_kernel": ("l2-gather", (2 * (E * 516.0) + E * 292.0) / 3 + P * 520.0, PEAK_L2_GATHER_GBS,
                                "reverse gather of the three blocks' backward (two dense: TQ row + dv1 row per in-edge)"),
        "edge_mlp_fwd_kernel": ("mfma", conv2, PEAK_F16_MFMA_TFLOPS, "conv2 / conv4 per edge, 2 E 64 64 flops per launch (3 split-fp16 products executed)"),
        "edge_mlp_bwd_kernel": ("mfma", 2 * conv2, PEAK_F16_MFMA_TFLOPS, "dgrad + wgrad of conv2 / conv4, 2 x 2 E 64 64 flops per launch"),
        "head_conv6_kernel": ("mfma", 2.0 * P * F_ * C6, PEAK_F16_MFMA_TFLOPS, "conv6 + statistics + max over the points, 2 P 192 1024 flops; nothing stored"),
        "head_conv7_kernel": ("hbm", 4.0 * P * (F_ + C7), PEAK_HBM_GBS, "reads cat [P, 192], writes u7 [P, 512]"),
        "head_conv8_kernel": ("hbm", 4.0 * P * (C7 + C8), PEAK_HBM_GBS, "reads u7, writes u8 [P, 256]"),
        "head_dz7_kernel": ("hbm", 4.0 * P * (2 * C8 + 2 * C7), PEAK_HBM_GBS, "reads dv8, u8, u7, writes dv7"),
        "head_dcat_kernel": ("hbm", 4.0 * P * (2 * C7 + F_), PEAK_HBM_GBS, "reads dv7, u7, writes dL/dcat"),
        "head_dw8_kernel": ("hbm", 4.0 * P * (2 * C8 + C7), PEAK_HBM_GBS, "reads dv8, u8, u7 once"),
        "head_dw7_kernel": ("hbm", 4.0 * P * (2 * C7 + F_), PEAK_HBM_GBS, "reads dv7, u7, cat once"),
    }


def sampled_pass(lib, fn, n):
    """mean wall time of fn over n calls between two fences (no events inside), then ONE more call with the per-kernel
    events recorded"""
    import torch
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    lib.dgcnn_profile_enable(1)
    fn()
    lib.dgcnn_profile_enable(0)
    torch.cuda.synchronize()
    return dt, region_times(lib)


def run_model_leg(args, dev, B, N, k, which, steps, with_train, cpu_batch, with_cpu):
    """full DGCNN on ONE GPU: eval forward (test.py:122 form: model.eval(), no_grad) and, with_train, the training step of
    run_train's shape at batch B.  Inputs resident in HBM in the strides train.py:176 / test.py:119 hand over."""
    import torch
    from ahn3_dgcnn_pytorch_amd import DGCNN, _lib, blocks, ops
    from ahn3_dgcnn_pytorch_amd.optim import FlatAdam
    lib = _lib.load()
    torch.manual_seed(0)
    model = DGCNN(5, 4, k, dropout=0.5, emb_dims=1024, cuda=True)
    pts, lab = synthetic_blocks(B, N, 0)
    pts, lab = pts.to(dev), lab.to(dev)
    x = blocks.to_model_input(pts)
    model.eval()

    def fwd():
        with torch.no_grad():
            model(x)

    dt, regions = sampled_pass(lib, fwd, steps)
    kernels = kernel_entries(regions, model_work(B, N, k, False), {})
    out = {
        "metric": "DGCNN eval-forward points/sec (B=%d,N=%d,k=%d)" % (B, N, k),
        "value": B * N / dt, "unit": "points/s", "n_gpus": 1, "steps": steps, "ms_per_step": dt * 1e3,
        "higher_is_better": True, "data": "synthetic",
        "dtype": "f32 (results); kNN pre-filter on f16 MFMA (exact f32 refine); conv2/conv4 and the head's products as split-f16 "
                 "MFMA products (hi.hi + hi.lo + lo.hi, 2^-22 relative) -- the fp16-MFMA edge-MLP of BASELINE.json configs[4]; a "
                 "one-product fp16 mode (2^-11 operand error) is not offered: logits would leave the 1e-4 parity bound",
        "config": {"workload": "full DGCNN segmentation forward, eval mode (4 EdgeConv stages in 3 blocks + emb_dims=1024 head), %s"
                               % which, "batch": B, "points": N, "k": k, "emb_dims": 1024, "launch": "eager"},
        "roofline": kernels[0] if kernels else None,
        "kernels": kernels[1:],
    }
    if with_cpu:
        out["cpu_baseline"] = cpu_baseline_eval(cpu_batch, N, k, B)
        out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    if with_train:
        model.train()
        opt = FlatAdam(model.parameters(), lr=1e-3, weight_decay=1e-4)

        def step():
            xx = blocks.to_model_input(pts)
            opt.zero_grad()
            loss, _pred = ops.seg_loss(model(xx), lab)
            loss.backward()
            opt.step()

        dts, regions = sampled_pass(lib, step, max(3, steps // 2))
        kt = kernel_entries(regions, model_work(B, N, k, True), {})
        out["train_step"] = {"metric": "DGCNN training-step points/sec (B=%d,N=%d,k=%d, one GPU)" % (B, N, k),
                             "value": B * N / dts, "unit": "points/s", "ms_per_step": dts * 1e3,
                             "roofline": kt[0] if kt else None, "kernels": kt[1:]}
    del model
    torch.cuda.empty_cache()
    return out


# ----------------------------------------------------------------------------------------------------------------------
# graph-replay leg: the block step captured once in a FRESH process and replayed (no launch latency, no host in the loop)
def graph_child(args):
    """runs in a child process that has done nothing else on the GPU (PyTorch-ROCm 2.10's stream capture crashes when
    results of eager steps are alive at capture time: DESIGN.md section 10).  Captures kNN + EdgeConv fwd + bwd of the block
    workload once, replays it --steps times between two events and prints one JSON line."""
    import torch
    from ahn3_dgcnn_pytorch_amd import ops
    from param_fill import cotangent
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    B, N, k, C = args.batch or 8, args.points, args.k, args.channels
    stages = make_stages(C, args.stages, dev)
    params = [p for st in stages for p in st.parameters()]
    x_pm = torch.randn(B, N, C, generator=torch.Generator().manual_seed(1000)).to(dev).requires_grad_(True)
    cot = cotangent((B, 64, N), tag="bench").to(dev).transpose(1, 2).contiguous()
    w2 = stages[1][0].weight if args.stages == 2 else None
    bn2 = stages[1][1] if args.stages == 2 else None

    def step():
        x_pm.grad = None
        for p in params:
            p.grad = None
        idx = ops.knn_pm(x_pm, k)
        y = ops.edgeconv_pm(x_pm, idx, stages[0][0].weight, stages[0][1], w2, bn2, training=True, slope=0.2)
        y.backward(cot)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    for _ in range(max(3, args.warmup)):
        graph.replay()
    torch.cuda.synchronize()
    n = args.steps
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print(json.dumps({"ms_per_step_graph": e0.elapsed_time(e1) / n, "ms_per_step_graph_wall": wall / n * 1e3, "replays": n,
                      "note": "one hipGraph of the block step (kNN build + EdgeConv forward + backward) captured in a fresh "
                              "process and replayed %d times between two HIP events: what the step costs without launch latency "
                              "or a host in the loop; `value` stays the eager number" % n}), flush=True)


def run_graph_child(args):
    """parent side: spawn the child BEFORE this process makes its first GPU call; relay its line (None on any failure:
    a crash of the capture must not take the bench line with it)"""
    cmd = [sys.executable, os.path.abspath(__file__), "--graph-child", "--steps", str(max(20, args.steps)), "--warmup", str(args.warmup),
           "--stages", str(args.stages), "--points", str(args.points), "--k", str(args.k), "--channels", str(args.channels)]
    if args.batch:
        cmd += ["--batch", str(args.batch)]
    try:
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    except Exception as e:  # noqa: BLE001
        return {"error": "graph child: %r" % (e,)}
    for line in reversed(p.stdout.splitlines()):
        if line.startswith("{"):
            try:
                return json.loads(line)
            except ValueError:
                break
    return {"error": "graph child exited with %d" % p.returncode, "stderr_tail": p.stderr[-400:]}


# ----------------------------------------------------------------------------------------------------------------------
def main():
    args = parse_args()
    if args.graph_child:
        return graph_child(args)
    if "RANK" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))

    # keep rank 0's stdout to the one JSON line: RCCL writes its version banner and its warnings (missing iommu=pt,
    # unreadable topology nodes in a container) to stdout unless told otherwise
    if os.environ.get("NCCL_DEBUG", "VERSION").upper() in ("VERSION", "INFO"):
        os.environ["NCCL_DEBUG"] = "WARN"
    os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")
    single = int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.gpus == 1
    graph_leg = None
    if single and args.workload == "block" and not args.no_graph_leg:
        graph_leg = run_graph_child(args)      # before this process touches the GPU
    import torch.distributed as dist

    rank, world, dev = init_ranks(args)
    cpu = world == 1 and not args.no_cpu_baseline
    try:
        if args.workload == "train":
            out = run_train(args, rank, world, dev, args.steps, args.warmup, cpu)
        else:
            out = run_block(args, rank, world, dev)
            if out is not None and graph_leg is not None:
                out["graph_replay"] = graph_leg
                if "ms_per_step_graph" in graph_leg:
                    out["ms_per_step_graph"] = graph_leg["ms_per_step_graph"]
            if not args.no_train_leg:
                leg = run_train(args, rank, world, dev, args.train_steps, 3, cpu)
                if world == 1 and leg is not None:
                    # the per-GPU share of the 8-GPU run of the same global batch, measured on this one GPU: what strong
                    # scaling to 8 ranks can reach before the all-reduce (t(B=32) / t(B=4))
                    b4 = run_train(args, rank, world, dev, args.train_steps, 3, False, global_batch=4)
                    leg["per_gpu_share_of_8"] = {k_: b4[k_] for k_ in ("ms_per_step", "ms_per_step_median", "value", "host_enqueue_ms",
                                                                         "region_total_ms")}
                    leg["per_gpu_share_of_8"]["global_batch"] = 4
                    leg["projected_strong_scaling_8"] = leg["ms_per_step"] / b4["ms_per_step"]
                if out is not None:
                    out["train_step"] = leg
            if world == 1 and not args.no_model_legs:
                out["eval_forward"] = run_model_leg(args, dev, 8, args.points, args.k, "BASELINE.json configs[2]", 20, False, 1, cpu)
                out["stress"] = run_model_leg(args, dev, 8, 32768, 64, "BASELINE.json configs[4] (large-tile stress)", 6, True, 1,
                                              cpu and not args.no_stress_cpu)
    finally:
        if _COMM is not None:
            _COMM.close()      # ncclCommDestroy now, not from __del__ at interpreter shutdown
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
