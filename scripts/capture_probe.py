"""dev probe for the hipGraph capture crash recorded in round 1 (gpurun_out/cap.log: segfault in torch/cuda/graphs.py
capture_end): runs the capture of one kNN + EdgeConv fwd + bwd step in fresh child processes, one state variant each,
and reports which of them die.   usage: capture_probe.py            (parent)   |   capture_probe.py <variant>   (child)"""
import faulthandler
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
VARIANTS = ["clean", "keep_graph", "keep_detached", "keep_idx", "profile_on", "no_side_stream", "keep_graph_no_backward",
            "grads_alive_default_stream", "torch_only_keep", "keep_detached_no_knn", "keep_detached_fwd_only", "keep_two_unrelated", "keep_detached:thread_local",
            "keep_detached:relaxed", "torch_only_keep:thread_local", "torch_only_keep:relaxed"]


def child(variant):
    faulthandler.enable()
    mode = "global"
    if ":" in variant:
        variant, mode = variant.split(":")
    import torch
    from ahn3_dgcnn_pytorch_amd import _lib, ops
    from oracle import torch_ref
    from param_fill import fill_tensor
    device = torch.device("cuda:0")
    B, C, N, k = 2, 64, 512, 20
    stage = torch_ref.make_edge_stage(2 * C)
    stage.load_state_dict({n: fill_tensor("stage0.%s" % n, v) for n, v in stage.state_dict().items()})
    stage.to(device).train()
    x = torch.randn(B, N, C, generator=torch.Generator().manual_seed(5)).to(device).requires_grad_(True)
    cot = torch.randn(B, N, 64, generator=torch.Generator().manual_seed(6)).to(device)

    idx_fixed = ops.knn_pm(x, k)
    wt = torch.randn(64, 64, device=device)

    def step(backward=True):
        x.grad = None
        for p in stage.parameters():
            p.grad = None
        if variant == "torch_only_keep":    # no kernel of this library inside the capture
            y = torch.relu(torch.matmul(x, wt)).sum(dim=1)
            if backward:
                y.sum().backward()
            return idx_fixed, y
        if variant == "keep_detached_fwd_only":
            backward = False
        idx = idx_fixed if variant == "keep_detached_no_knn" else ops.knn_pm(x, k)
        y = ops.edgeconv_pm(x, idx, stage[0].weight, stage[1], None, None, training=True, slope=0.2)
        if backward:
            y.backward(cot)
        return idx, y

    keep = None
    if variant == "keep_graph":
        keep = step()                       # outputs with their autograd graph (already backwarded) stay alive
    elif variant == "keep_graph_no_backward":
        keep = step(backward=False)         # ... with the saved tensors still held
    elif variant in ("keep_detached", "torch_only_keep", "keep_detached_no_knn", "keep_detached_fwd_only"):
        i, y = step()
        keep = (i.clone(), y.detach().clone())
    elif variant == "keep_two_unrelated":
        step()
        keep = (torch.zeros(2, 512, 20, dtype=torch.int32, device=device), torch.zeros(2, 512, 64, device=device))
    elif variant == "keep_idx":
        keep = step()[0]
    elif variant == "grads_alive_default_stream":
        step()                              # .grad tensors allocated on the DEFAULT stream are alive at capture start
    else:
        step()
    if variant == "profile_on":
        _lib.load().dgcnn_profile_enable(1)
    torch.cuda.synchronize()
    if variant not in ("no_side_stream", "grads_alive_default_stream"):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode=mode):
        step()
    graph.replay()
    torch.cuda.synchronize()
    print("variant %s (capture_error_mode=%s): capture + replay ok (kept: %s)" % (variant, mode, type(keep).__name__))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        for v in VARIANTS:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), v], capture_output=True, text=True, timeout=300)
            tail = (r.stdout.strip().splitlines() or [""])[-1]
            err = [ln for ln in r.stderr.splitlines() if "Error" in ln or "Fatal" in ln or "File" in ln][:6]
            print("%-28s rc=%4d  %s  %s" % (v, r.returncode, tail, " | ".join(err)), flush=True)
