"""CPU tests (gloo, world_size 2) of the data-parallel exchange step: batch elements sharded over
ranks, one all-reduce of the flat gradient buffer.  The module under DP here is the CPU oracle
(tests may use it); on the GPU box the same wrapper runs over RCCL with the HIP model."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import torch_ref
from param_fill import cotangent, fill_module

WORLD = 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build():
    m = torch_ref.RefDGCNN(5, 4, 6, dropout=0.0)
    fill_module(m)
    m.train()
    return m


def _worker(rank, port, x, R, out):
    from ahn3_dgcnn_pytorch_amd.parallel import FlatGradAllReduce, shard_batch, sync_buffers
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        m = _build()
        sync = FlatGradAllReduce(m.parameters())
        assert sync.flat.numel() == 980864 and len(sync.params) == 25
        sync.zero()
        xs, Rs = shard_batch(x, rank, WORLD), shard_batch(R, rank, WORLD)
        # loss = mean over the GLOBAL batch of per-sample sums -> local sum / local batch, then averaged
        y = m(xs)
        ((y * Rs).sum() / xs.shape[0]).backward()
        flat = sync.sync().clone()
        # the .grad of every parameter is a live view of the flat buffer
        off = 0
        for p in sync.params:
            assert p.grad.data_ptr() == sync.flat.data_ptr() + 4 * off
            off += p.numel()
        sync_buffers(m)
        if rank == 0:
            out["flat"] = flat
            out["rm1"] = m.bn1.running_mean.clone()
        # every rank must hold the same reduced gradients
        g = [torch.zeros_like(flat) for _ in range(WORLD)]
        dist.all_gather(g, flat)
        assert torch.equal(g[0], g[1])
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_flat_grad_allreduce_matches_shardwise_oracle():
    torch.manual_seed(0)
    B, N = 4, 96
    x = torch.rand(B, 4, N) * torch.tensor([30.0, 30.0, 20.0, 20.0]).view(1, 4, 1)
    R = cotangent((B, 5, N))
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(_free_port(), x, R, out), nprocs=WORLD, join=True)

    # expectation: gradients of the oracle run shard by shard and averaged (per-replica BN statistics)
    want, rms = None, []
    for r in range(WORLD):
        m = _build()
        xs, Rs = x[r * 2:(r + 1) * 2], R[r * 2:(r + 1) * 2]
        ((m(xs) * Rs).sum() / xs.shape[0]).backward()
        seen, flat = set(), []
        for p in m.parameters():
            if id(p) not in seen:
                seen.add(id(p))
                flat.append(p.grad.reshape(-1))
        flat = torch.cat(flat)
        want = flat if want is None else want + flat
        rms.append(m.bn1.running_mean.clone())
    want = want / WORLD
    got = out["flat"]
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) <= 2e-5 * scale
    torch.testing.assert_close(out["rm1"], (rms[0] + rms[1]) / 2, rtol=1e-5, atol=1e-6)


def test_shard_batch_contract():
    from ahn3_dgcnn_pytorch_amd.parallel import shard_batch
    x = torch.arange(8 * 3).view(8, 3)
    parts = [shard_batch(x, r, 4) for r in range(4)]
    assert torch.equal(torch.cat(parts), x) and all(p.shape[0] == 2 for p in parts)
    with pytest.raises(ValueError):
        shard_batch(x, 0, 3)


def _overlap_worker(rank, port):
    from ahn3_dgcnn_pytorch_amd.parallel import FlatGradAllReduce
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        sizes = (3, 5, 7, 11)
        ps = [torch.nn.Parameter(torch.zeros(n)) for n in sizes]
        sync = FlatGradAllReduce(ps, overlap=True)
        for _ in range(2):                                   # two steps: the tail state is cleared by sync()
            sync.zero()
            for i, (p, v) in enumerate(zip(ps, sync.views)):
                v.fill_(float(rank + 1) * (i + 1))           # what a backward would have written into the slices
                p.grad = v
            assert sync.wants_overlap()
            assert not sync.reduce_tail_async(ps[2], [ps[2]])        # ps[3] lies behind it and is not final: refused
            assert sync.reduce_tail_async(ps[2], [ps[2], ps[3]])     # the tail is in flight ...
            assert not sync.wants_overlap()
            flat = sync.sync()                                       # ... the front is reduced here, then the join
            want = torch.cat([torch.full((n,), 1.5 * (i + 1)) for i, n in enumerate(sizes)])
            assert torch.allclose(flat, want), (flat, want)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_tail_of_the_buffer_reduced_ahead_of_sync():
    """FlatGradAllReduce(overlap=True): the all-reduce of the buffer's tail is started early (by the backward of
    ops.dgcnn_full, once the head's gradients are final) and joined by sync(), which reduces the front part; the result
    is the plain one-bucket average."""
    mp.spawn(_overlap_worker, args=(_free_port(),), nprocs=WORLD, join=True)
