"""oracle/torch_ref.py -- TEST INFRASTRUCTURE ONLY (checker and CPU baseline, never the product).

CPU restatement, in plain PyTorch ops, of the reference's EdgeConv path and of the DGCNN
segmentation network around it.  It deliberately issues the SAME ATen op sequence as the
reference so that (a) under one torch build its outputs are bit-equal to the imported
reference (asserted by tests/golden/gen_golden.py when the fixtures are made) and (b) it has
the reference's cost profile -- including the materialised B x N x N distance tensor -- when
bench.py times it as the `cpu_baseline` ("kind": "port").

Reference lines followed (all /root/reference/model.py):
    pairwise_knn      <- knn, :25-40
    edge_features     <- get_graph_feature, :43-83  (difference first, centre second, :81)
    RefDGCNN.__init__ <- DGCNN.__init__, :88-141    (layer hyper-parameters, module names)
    RefDGCNN.forward  <- DGCNN.forward, :144-193    (block wiring and `depth` exits)

Differences from the reference, on purpose:
  * the stray `dim9=` keyword of model.py:147 is accepted and ignored;
  * the device follows the input tensor instead of a constructor flag (D8);
  * `forward` can take externally supplied neighbour indices and can return the indices it
    used, so that tests can separate kNN parity (bit-exact gate) from edge-MLP parity (1e-4).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Parity status: pinned -- see tests/golden/ and tests/test_oracle_golden.py.
"""
import torch
import torch.nn as nn

# (name, in_channels, out_channels) of the five edge convolutions and their block wiring
EDGE_BLOCKS = (("conv1", "conv2"), ("conv3", "conv4"), ("conv5",))
NEG_SLOPE = 0.2


def pairwise_knn(x, k):
    """x [B, C, N] -> int64 [B, N, k]: the k largest of -||xi - xl||^2 per row, as torch.topk
    orders them (model.py:35-39)."""
    inner = -2 * torch.matmul(x.transpose(2, 1), x)
    xx = torch.sum(x ** 2, dim=1, keepdim=True)
    pd = -xx - inner - xx.transpose(2, 1)
    return pd.topk(k=k, dim=-1)[1]


def edge_features(x, k=20, idx=None, **_ignored):
    """x [B, C, N] -> [B, 2C, N, k] edge tuples (x_j - x_i, x_i), memory [B, N, k, 2C]
    (model.py:54-83).  `idx` ([B, N, k], local indices) overrides the kNN search."""
    B, N = x.size(0), x.size(2)
    x = x.view(B, -1, N)
    if idx is None:
        idx = pairwise_knn(x, k=k)
    base = torch.arange(0, B, device=x.device).view(-1, 1, 1) * N
    flat = (idx.to(x.device).long() + base).view(-1)
    C = x.size(1)
    pts = x.transpose(2, 1).contiguous()
    nbr = pts.view(B * N, -1)[flat, :].view(B, N, k, C)
    ctr = pts.view(B, N, 1, C).repeat(1, 1, k, 1)
    return torch.cat((nbr - ctr, ctr), dim=3).permute(0, 3, 1, 2)


class RefDGCNN(nn.Module):
    """Same parameters, buffers and module names as the reference DGCNN (95 state_dict keys:
    every BatchNorm is registered both as `bnN` and inside the `convN` Sequential)."""

    def __init__(self, num_classes, num_features, k, dropout=0.5, emb_dims=1024):
        super().__init__()
        self.k, self.num_classes, self.num_features = k, num_classes, num_features
        widths = [64, 64, 64, 64, 64, emb_dims, 512, 256]
        for i, w in enumerate(widths, start=1):
            setattr(self, "bn%d" % i, nn.BatchNorm2d(w) if i <= 5 else nn.BatchNorm1d(w))
        ins = [2 * num_features, 64, 128, 64, 128, 192, 1216, 512]
        for i, (cin, w) in enumerate(zip(ins, widths), start=1):
            conv = nn.Conv2d(cin, w, kernel_size=1, bias=False) if i <= 5 else nn.Conv1d(cin, w, kernel_size=1, bias=False)
            setattr(self, "conv%d" % i, nn.Sequential(conv, getattr(self, "bn%d" % i),
                                                      nn.LeakyReLU(negative_slope=NEG_SLOPE)))
        self.dp1 = nn.Dropout(p=dropout)
        self.conv9 = nn.Conv1d(256, num_classes, kernel_size=1, bias=False)

    def forward(self, x, depth=10, idx=None, return_idx=False):
        """idx: optional list of three [B, N, k] index tensors (one per EdgeConv block)."""
        n_pts = x.size(2)
        used, feats, cur = [], [], x
        for bi, names in enumerate(EDGE_BLOCKS):
            nbr = idx[bi] if idx is not None else pairwise_knn(cur.view(cur.size(0), -1, n_pts), self.k)
            used.append(nbr)
            e = edge_features(cur, k=self.k, idx=nbr)
            for nm in names:
                e = getattr(self, nm)(e)
            cur = e.max(dim=-1, keepdim=False)[0]
            feats.append(cur)
            if depth == bi + 1:
                return (cur, used) if return_idx else cur

        def done(t):
            return (t, used) if return_idx else t

        y = torch.cat(feats, dim=1)
        if depth == 4:
            return done(y)
        y = self.conv6(y)
        if depth == 5:
            return done(y)
        y = y.max(dim=-1, keepdim=True)[0]
        if depth == 6:
            return done(y)
        y = torch.cat([y.repeat(1, 1, n_pts)] + feats, dim=1)
        if depth == 7:
            return done(y)
        y = self.conv7(y)
        if depth == 8:
            return done(y)
        y = self.conv8(y)
        if depth == 9:
            return done(y)
        return done(self.conv9(self.dp1(y)))


def edgeconv_block(x, k, convs, idx=None):
    """One EdgeConv block on x [B, C, N] with `convs` = list of Sequential(Conv2d, BN2d, LReLU):
    graph feature -> conv stages -> max over k (model.py:147-150 / 154-157 / 161-163)."""
    e = edge_features(x, k=k, idx=idx)
    for c in convs:
        e = c(e)
    return e.max(dim=-1, keepdim=False)[0]


def make_edge_stage(cin, cout=64):
    """Conv2d 1x1 (no bias) + BatchNorm2d + LeakyReLU(0.2): model.py:116-130."""
    return nn.Sequential(nn.Conv2d(cin, cout, kernel_size=1, bias=False), nn.BatchNorm2d(cout),
                         nn.LeakyReLU(negative_slope=NEG_SLOPE))
