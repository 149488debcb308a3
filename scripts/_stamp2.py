import torch, sys, copy
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from ahn3_dgcnn_pytorch_amd import ops
from oracle import torch_ref
dev = torch.device('cuda:0')
B, C, N, k = 8, 64, 7000, 30
x = torch.randn(B, N, C, device=dev, requires_grad=True)
st = [torch_ref.make_edge_stage(2 * C).to(dev), torch_ref.make_edge_stage(64).to(dev)]
idx = ops.knn_pm(x.detach(), k)
for i in range(2):
    y = ops.edgeconv_pm(x, idx, st[0][0].weight, st[0][1], st[1][0].weight, st[1][1], training=True, slope=0.2)
    y.sum().backward()
    torch.cuda.synchronize()
    print("---- run", i)
