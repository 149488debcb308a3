"""The PRODUCT data-parallel step on the GPU: two fresh processes (one per rank, as deployed; both on
cuda:0 of the one-GPU box, exchange over gloo -- on the 8-GPU node the same code runs over RCCL), each running the HIP
DGCNN on its shard -> device loss -> backward -> FlatGradAllReduce.sync -> FlatAdam.step.  Expectation = the same model
run shard by shard in ONE process, gradients averaged, one FlatAdam step on the average (per-replica BatchNorm
statistics, running buffers averaged).  Also: bench.py starts its own ranks and refuses rank-count mismatches."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
WORLD = 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _env():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["OMP_NUM_THREADS"] = "2"
    return env


@pytest.mark.timeout(600)
@pytest.mark.parametrize("overlap", [False, True])   # True: the all-reduce of conv6 .. conv9 starts inside the backward
def test_two_rank_product_step_matches_shardwise_run(tmp_path, device, overlap):
    sys.path.insert(0, HERE)
    import dp_child
    from ahn3_dgcnn_pytorch_amd.optim import FlatAdam
    from ahn3_dgcnn_pytorch_amd.parallel import FlatGradAllReduce
    g = torch.Generator().manual_seed(0)
    B, N = 4, 384
    cfg = {"classes": 5, "k": 8,
           "x": torch.rand(B, 4, N, generator=g) * torch.tensor([30.0, 30.0, 20.0, 20.0]).view(1, 4, 1),
           "labels": torch.randint(0, 5, (B, N), generator=g)}
    fin, fout = str(tmp_path / "in.pt"), str(tmp_path / "out.pt")
    torch.save(cfg, fin)
    port = str(_free_port())
    env = _env()
    if overlap:
        env["DGCNN_DP_OVERLAP"] = "1"
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dp_child.py"), str(r), str(WORLD), port, "gloo", fin, fout],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(WORLD)]
    outs = [p.communicate(timeout=540)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-4000:]
    got = torch.load(fout, weights_only=False)
    assert got["ranks"] == WORLD

    # shard-wise single-process run of the same product model
    want, rms, losses = None, [], []
    for r in range(WORLD):
        m = dp_child.build_model(cfg)
        sync = FlatGradAllReduce(m.parameters())
        opt = FlatAdam(m.parameters(), lr=1e-3, weight_decay=1e-4, grad_buffer=sync.flat)
        per = B // WORLD
        loss, red = dp_child.local_step(m, cfg["x"][r * per:(r + 1) * per].to(device),
                                        cfg["labels"][r * per:(r + 1) * per].to(device), sync, opt)
        want = red.cpu() if want is None else want + red.cpu()
        rms.append(m.bn1.running_mean.cpu())
        losses.append(float(loss))
    want = want / WORLD
    scale = float(want.abs().max())
    assert float((got["reduced"] - want).abs().max()) <= 1e-6 * scale     # (deterministic kernels; the exchange adds two ranks)
    assert abs(got["loss"] - losses[0]) <= 1e-6 * max(1.0, abs(losses[0]))
    torch.testing.assert_close(got["rm1"], (rms[0] + rms[1]) / 2, rtol=1e-5, atol=1e-6)
    # one FlatAdam step on the averaged gradient from the same initial weights
    m = dp_child.build_model(cfg)
    sync = FlatGradAllReduce(m.parameters())
    opt = FlatAdam(m.parameters(), lr=1e-3, weight_decay=1e-4, grad_buffer=sync.flat)
    p0 = opt.flat.cpu().clone()
    sync.flat.copy_(got["reduced"].to(device))
    for p, v in zip(sync.params, sync.views):
        p.grad = v
    opt.step()
    assert torch.equal(opt.flat.cpu(), got["params"])      # same kernel, same inputs: the update is bit-identical
    assert float((got["params"] - p0).abs().max()) > 0


def _bench(args, timeout=540):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=_env(), capture_output=True,
                          text=True, timeout=timeout)


@pytest.mark.timeout(600)
def test_bench_starts_its_own_ranks_for_the_train_workload():
    """`python bench.py --gpus 2` without a launcher environment starts two ranks itself (rehearsed on one GPU over
    gloo) and reports the configs[3] training step with the all-reduce in it."""
    env_had_rank = "RANK" in os.environ
    assert not env_had_rank
    p = _bench(["--gpus", "2", "--workload", "train", "--backend", "gloo", "--same-device", "--batch", "4",
                "--points", "512", "--k", "8", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"])
    assert p.returncode == 0, (p.stdout + p.stderr)[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["scaling"] == "strong"
    assert out["config"]["global_batch"] == 4 and out["config"]["batch_per_gpu"] == 2
    assert out["allreduce"]["bytes"] == 980864 * 4 and out["allreduce"]["ms"] > 0
    assert out["value"] > 0 and out["unit"] == "points/s"


@pytest.mark.timeout(600)
def test_bench_default_workload_with_two_ranks():
    """the driver's N > 1 form of the headline line: every rank runs the block on its own shard (weak scaling), one
    all-reduce of the flat gradient buffer per step, and the line carries the configs[3] leg sharded over the ranks."""
    p = _bench(["--gpus", "2", "--backend", "gloo", "--same-device", "--batch", "2", "--points", "600", "--k", "10",
                "--steps", "4", "--warmup", "1", "--train-steps", "2"])
    assert p.returncode == 0, (p.stdout + p.stderr)[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["ranks"] == 2
    assert out["value"] > 0 and "cpu_baseline" not in out          # the CPU leg runs at N = 1 only
    assert out["roofline"]["frac"] <= 1.0 and all(k["frac"] <= 1.0 for k in out["kernels"])
    leg = out["train_step"]
    assert leg["rccl_ranks"] == 2 and leg["config"]["global_batch"] == 32 and leg["config"]["batch_per_gpu"] == 16
    assert leg["scaling"] == "strong" and leg["value"] > 0


@pytest.mark.timeout(300)
def test_bench_refuses_a_rank_count_mismatch():
    env = _env()
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode != 0 and "refusing" in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    # more ranks than GPUs under the product backend: every rank needs its own device -> non-zero, no JSON line
    p = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-train-leg"], timeout=280)
    if torch.cuda.device_count() < 2:
        assert p.returncode != 0
        assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_exchange_behind_the_c_abi_single_rank(device):
    """dgcnn_comm_*: communicator of one rank on the box's GPU (the N-rank form needs N GPUs: RCCL refuses two ranks on
    one device) -- init, all-reduce (sum and mean), broadcast, FlatGradAllReduce over it, destroy"""
    from ahn3_dgcnn_pytorch_amd.parallel import FlatGradAllReduce, HipComm
    comm = HipComm(0, 1)
    x = torch.arange(1000, dtype=torch.float32, device=device) * 0.25 - 3.0
    want = x.clone()
    comm.all_reduce(x)
    comm.all_reduce(x, average=True)
    comm.broadcast(x, 0)
    torch.cuda.synchronize()
    assert torch.equal(x, want)
    p = torch.nn.Parameter(torch.ones(7, 3, device=device))
    sync = FlatGradAllReduce([p], comm=comm)
    sync.zero()
    (p * 2.0).sum().backward()
    flat = sync.sync()
    assert sync.world_size == 1 and torch.equal(flat, torch.full((21,), 2.0, device=device))
    with pytest.raises(TypeError):
        comm.all_reduce(torch.zeros(4, dtype=torch.float64, device=device))
    comm.close()


def test_edgeconv_gradients_land_in_the_flat_buffer(device):
    """FlatGradAllReduce names, per parameter, the slice of its exchange buffer (`_dgcnn_grad_sink`); the EdgeConv backward
    writes the parameter gradients there, `pack` finds nothing to copy, and an accumulating backward (zero(set_to_none=False))
    still adds up."""
    from ahn3_dgcnn_pytorch_amd import ops
    from ahn3_dgcnn_pytorch_amd.parallel import FlatGradAllReduce
    torch.manual_seed(5)
    B, N, C, k = 2, 300, 64, 12
    conv = torch.nn.Conv2d(2 * C, 64, 1, bias=False).to(device)
    bn = torch.nn.BatchNorm2d(64).to(device)
    conv2 = torch.nn.Conv2d(64, 64, 1, bias=False).to(device)
    bn2 = torch.nn.BatchNorm2d(64).to(device)
    params = [conv.weight, bn.weight, bn.bias, conv2.weight, bn2.weight, bn2.bias]
    x = torch.randn(B, N, C, device=device)
    idx = ops.knn_pm(x, k)
    cot = torch.randn(B, N, 64, device=device)

    def run():
        y = ops.edgeconv_pm(x, idx, conv.weight, bn, conv2.weight, bn2, training=True, slope=0.2)
        y.backward(cot)

    run()                                             # no sinks yet: plain gradients
    want = [p.grad.clone() for p in params]
    for p in params:
        p.grad = None
    sync = FlatGradAllReduce(params)
    sync.zero()
    run()
    for p, v, w in zip(params, sync.views, want):
        assert p.grad.data_ptr() == v.data_ptr(), "the gradient was not written into its slice"
        assert torch.equal(p.grad, w)
    flat = sync.sync().clone()
    assert torch.equal(flat, torch.cat([w.reshape(-1) for w in want]))
    sync.zero(set_to_none=False)                      # accumulation: .grad stays (zeroed) and two backward passes add up
    run()
    run()
    for p, w in zip(params, want):
        assert torch.allclose(p.grad, 2.0 * w, rtol=1e-6, atol=0.0)


def test_shared_weight_claims_its_sink_once(device):
    """One convolution / BatchNorm used by TWO EdgeConv nodes of the same backward: when the second node runs the
    parameter has no .grad yet (AccumulateGrad has not fired), so a sink decided by `.grad is None` alone would be handed
    out twice and autograd would add a tensor to itself.  The sink is claimed once per zero() / pack() cycle; the sum of
    the two contributions must come out."""
    from ahn3_dgcnn_pytorch_amd import ops
    from ahn3_dgcnn_pytorch_amd.parallel import FlatGradAllReduce
    torch.manual_seed(6)
    B, N, C, k = 2, 300, 64, 12
    conv = torch.nn.Conv2d(2 * C, 64, 1, bias=False).to(device)
    bn = torch.nn.BatchNorm2d(64).to(device)
    params = [conv.weight, bn.weight, bn.bias]
    x = torch.randn(B, N, C, device=device)
    idx = ops.knn_pm(x, k)
    cot = torch.randn(B, N, 64, device=device)

    def run():
        y1 = ops.edgeconv_pm(x, idx, conv.weight, bn, None, None, training=True, slope=0.2)
        y2 = ops.edgeconv_pm(y1, idx, conv.weight, bn, None, None, training=True, slope=0.2)   # the same weights again
        y2.backward(cot)

    run()
    want = [p.grad.clone() for p in params]
    for p in params:
        p.grad = None
    sync = FlatGradAllReduce(params)
    for _ in range(2):                                # two cycles: the claim is released by zero() / pack()
        sync.zero()
        run()
        flat = sync.sync()
        for p, w in zip(params, want):
            assert torch.allclose(p.grad, w, rtol=1e-5, atol=1e-6 * float(w.abs().max()))
        assert torch.allclose(flat, torch.cat([w.reshape(-1) for w in want]), rtol=1e-5, atol=1e-6 * float(flat.abs().max()))
