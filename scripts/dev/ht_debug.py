import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import torch, copy
import test_head_tail_gpu as T
from ahn3_dgcnn_pytorch_amd import ops
dev = torch.device("cuda:0")
def case(B, N, training=True, p_drop=0.0, seed=1):
    P = B * N
    w7, w8, w9, bn7, bn8 = T.make_layers(seed)
    g = torch.Generator().manual_seed(100 + seed)
    cat = torch.randn(P, 192, generator=g); add7 = torch.randn(B, 512, generator=g); cot = torch.randn(P, 5, generator=g)
    bn7.train(training), bn8.train(training)
    dbn7, dbn8 = copy.deepcopy(bn7).to(dev), copy.deepcopy(bn8).to(dev)
    leaves = [t.to(dev).requires_grad_(True) for t in (cat, add7, w7, w8, w9)]
    dcat, dadd, dw7, dw8, dw9 = leaves
    y = ops.head_tail_rows(dcat, dadd, dw7[:, 1024:], dbn7, dw8, dbn8, dw9, B, N, training, 0.2, p_drop)
    (y * cot.to(dev)).sum().backward()
    rl = [t.double().requires_grad_(True) for t in (cat, add7, w7, w8, w9)]
    rcat, radd, rw7, rw8, rw9 = rl
    rb7, rb8 = copy.deepcopy(bn7).double(), copy.deepcopy(bn8).double()
    want, _ = T.reference(rcat, radd, rw7[:, 1024:], rw8, rw9, rb7, rb8, B, N, training, None, 1.0)
    (want * cot.double()).sum().backward()
    def e(got, ref): return ((got.detach().cpu().double() - ref.detach()).abs().max() / ref.detach().abs().max()).item()
    print("B=%d N=%d train=%d: y %.1e dcat %.1e dadd %.1e dW7 %.1e dW8 %.1e dW9 %.1e dg7 %.1e db7 %.1e dg8 %.1e db8 %.1e" % (
        B, N, training, e(y, want), e(dcat.grad, rcat.grad), e(dadd.grad, radd.grad), e(dw7.grad[:, 1024:], rw7.grad[:, 1024:]),
        e(dw8.grad, rw8.grad), e(dw9.grad, rw9.grad), e(dbn7.weight.grad, rb7.weight.grad), e(dbn7.bias.grad, rb7.bias.grad),
        e(dbn8.weight.grad, rb8.weight.grad), e(dbn8.bias.grad, rb8.bias.grad)))
    if e(dcat.grad, rcat.grad) > 1e-4:
        d = (dcat.grad.cpu().double() - rcat.grad).abs().amax(1)
        bad = (d > 1e-4 * rcat.grad.abs().max()).nonzero().flatten()
        print("   bad rows:", bad[:20].tolist(), "... n=", len(bad), "first/last", bad.min().item(), bad.max().item())
for B, N in [(2, 256), (3, 256), (1, 999), (3, 333), (1, 300), (2, 384)]:
    case(B, N)
case(3, 333, training=False)
