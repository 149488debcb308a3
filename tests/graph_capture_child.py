"""Child process of test_edgeconv_gpu.py::test_step_is_graph_capturable."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

from ahn3_dgcnn_pytorch_amd import ops  # noqa: E402
from oracle import torch_ref  # noqa: E402
from param_fill import fill_tensor  # noqa: E402


def main():
    device = torch.device("cuda:0")
    B, C, N, k = 2, 64, 512, 20
    stage = torch_ref.make_edge_stage(2 * C)
    stage.load_state_dict({n: fill_tensor("stage0.%s" % n, v) for n, v in stage.state_dict().items()})
    stage.to(device).train()
    x = torch.randn(B, N, C, generator=torch.Generator().manual_seed(5)).to(device).requires_grad_(True)
    cot = torch.randn(B, N, 64, generator=torch.Generator().manual_seed(6)).to(device)

    def step():
        x.grad = None
        for p in stage.parameters():
            p.grad = None
        idx = ops.knn_pm(x, k)
        y = ops.edgeconv_pm(x, idx, stage[0].weight, stage[1], None, None, training=True, slope=0.2)
        y.backward(cot)
        return idx, y

    idx_e, y_e = step()  # eager reference; detached copies: an autograd graph kept alive across the capture makes
    idx_e, y_e = idx_e.clone(), y_e.detach().clone()  # hipStreamEndCapture crash (torch 2.10 / ROCm 7.0)
    gx_e, gw_e = x.grad.clone(), stage[0].weight.grad.clone()
    torch.cuda.synchronize()
    # PyTorch's capture protocol: the last eager executions before the capture run on a side stream
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        idx_g, y_g = step()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(idx_g, idx_e)
    assert torch.equal(y_g, y_e)  # the forward is deterministic
    assert float((x.grad - gx_e).abs().max()) <= 1e-5 * float(gx_e.abs().max())
    assert float((stage[0].weight.grad - gw_e).abs().max()) <= 1e-5 * float(gw_e.abs().max())
    print("replay matches eager")


if __name__ == "__main__":
    main()
