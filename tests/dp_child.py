"""One rank of tests/test_parallel_gpu.py: the PRODUCT data-parallel training step (HIP DGCNN -> device loss ->
backward -> FlatGradAllReduce.sync -> FlatAdam.step) on this rank's shard of the batch.

    python dp_child.py <rank> <world> <port> <backend> <in.pt> <out.pt>

A fresh process per rank (one process per GPU is the deployment model; on the one-GPU test box both ranks share cuda:0
and the exchange runs over gloo, on the 8-GPU node the same code runs over RCCL)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def build_model(cfg):
    from ahn3_dgcnn_pytorch_amd import DGCNN
    from param_fill import fill_module
    m = DGCNN(cfg["classes"], 4, cfg["k"], dropout=0.0, emb_dims=1024, cuda=True)
    fill_module(m)
    m.train()
    return m


def local_step(model, x, lab, sync, opt):
    """the training step of bench.py --workload train on one shard; returns the loss"""
    from ahn3_dgcnn_pytorch_amd import ops
    opt.zero_grad()
    loss, _ = ops.seg_loss(model(x), lab)
    loss.backward()
    sync.sync()
    reduced = sync.flat.clone()
    opt.step()
    return loss.detach(), reduced


def main():
    rank, world, port, backend, fin, fout = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5], sys.argv[6]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ahn3_dgcnn_pytorch_amd.optim import FlatAdam
        from ahn3_dgcnn_pytorch_amd.parallel import FlatGradAllReduce, broadcast, shard_batch, sync_buffers
        cfg = torch.load(fin, weights_only=False)
        model = build_model(cfg)
        # DGCNN_DP_OVERLAP=1: the tail of the buffer (conv6 .. conv9) is reduced under the blocks' backward
        sync = FlatGradAllReduce(model.parameters(), overlap=bool(os.environ.get("DGCNN_DP_OVERLAP")))
        opt = FlatAdam(model.parameters(), lr=1e-3, weight_decay=1e-4, grad_buffer=sync.flat)
        assert sync.flat.numel() == 980864 + (cfg["classes"] - 5) * 256 and len(sync.params) == 25
        broadcast(opt.flat, 0)
        x = shard_batch(cfg["x"], rank, world).to(dev)
        lab = shard_batch(cfg["labels"], rank, world).to(dev)
        loss, reduced = local_step(model, x, lab, sync, opt)
        off = 0
        for p in sync.params:   # parameters live in FlatAdam's buffer, their .grad in the shared gradient buffer
            assert p.data_ptr() == opt.flat.data_ptr() + 4 * off and p.grad.data_ptr() == sync.flat.data_ptr() + 4 * off
            off += p.numel()
        sync_buffers(model)
        got = [torch.zeros_like(reduced.cpu()) for _ in range(world)]
        dist.all_gather(got, reduced.cpu())
        assert all(torch.equal(got[0], g) for g in got), "ranks hold different reduced gradients"
        if rank == 0:
            torch.save({"reduced": reduced.cpu(), "params": opt.flat.cpu(), "loss": float(loss),
                        "rm1": model.bn1.running_mean.cpu(), "ranks": dist.get_world_size()}, fout)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
