"""Data formats either side of DGCNN.forward: the reference's block files and batch
layout on the way in, per-block classification on the way out.  Host-side mirror of the few lines of data.py /
train.py / test.py that touch the model; file discovery, LAS I/O, DTM / AGL features and the dataset classes are the
reference's own business and stay out ("out of scope").

    read_block      <- data.py:65-66     one tile: float64 .npy [n_points, F + 1], last column = label
    collate_pcs     <- data.py:169-182   batch of tiles, each subsampled (without replacement) to the smallest one
    to_model_input  <- train.py:175-176, test.py:118-119   .to(device), permute(0, 2, 1), .float()
    classify_block  <- test.py:296-311   block-local x / y, forward, softmax, max, confidence threshold
    classify_blocks <- test.py:281-330   the loop over a tile's blocks: points and labels stacked in block order
"""
import numpy as np
import torch

from . import ops
from .model import _compute_device

UNCLASSIFIED = 31  # test.py: the label of points below the confidence threshold


def read_block(path, mmap=True):
    """One tile file -> (points float64 [n, F], labels float64 [n]).  Memory-mapped by default."""
    tile = np.load(path, mmap_mode="r" if mmap else None)
    if tile.ndim != 2 or tile.shape[1] < 2:
        raise ValueError("%s: expected a [n_points, F + 1] array, got shape %s" % (path, tile.shape))
    return tile[:, :-1], tile[:, -1]


def collate_pcs(data):
    """[(points [n_i, F], labels [n_i], idx), ...] -> (float64 [B, n_min, F], float64 [B, n_min], idxs), every tile
    subsampled without replacement to the smallest tile of the batch (same numpy RNG calls as the reference, so a
    seeded run draws the same points)."""
    n_min = min(item[0].shape[0] for item in data)
    F = data[0][0].shape[1]
    batch_points = np.zeros((len(data), n_min, F))
    batch_labels = np.zeros((len(data), n_min))
    for i, (points, labels, _) in enumerate(data):
        keep = np.random.choice(range(points.shape[0]), n_min, replace=False)
        batch_points[i] = np.asarray(points)[keep]
        batch_labels[i] = np.asarray(labels)[keep]
    return torch.tensor(batch_points), torch.tensor(batch_labels), torch.tensor([item[2] for item in data])


def to_model_input(points, device=None):
    """float64 [B, N, F] (host or device) -> float32 [B, F, N] on the HIP device: the layout the reference feeds to
    forward, as a permuted VIEW of point-major memory (which is what the EdgeConv kernels read)."""
    dev = device if device is not None else _compute_device()
    p = points.to(dev, non_blocking=True)
    p = ops.points_to_float(p) if p.dtype == torch.float64 else p.float()
    return p.permute(0, 2, 1)


def classify_block(model, X, min_class_confidence=0.0, unclassified=UNCLASSIFIED):
    """Labels of one block (test.py:296-311).  X: ndarray [n, F] in tile coordinates (not modified); the first two
    columns are shifted to the block's lower-left corner before the forward pass.  Returns int64 ndarray [n, 1]."""
    X = np.array(X, dtype=np.float64, copy=True)
    X[:, 0] -= np.amin(X[:, 0])
    X[:, 1] -= np.amin(X[:, 1])
    x = to_model_input(torch.from_numpy(X[np.newaxis]))
    was_training = model.training
    model.eval()
    try:
        with torch.no_grad():
            pred, _ = ops.seg_predict(model(x), min_class_confidence, unclassified)
    finally:
        model.train(was_training)
    return pred.cpu().numpy().reshape(-1, 1)


def classify_blocks(model, block_data, min_class_confidence=0.0, unclassified=UNCLASSIFIED):
    """Labels of every block of a tile (the loop of test.py:281-330 without its file output): block_data is an
    iterable of ndarrays [n_i, F]; returns (points [sum n_i, F] stacked in block order, labels int64 [sum n_i, 1])."""
    pts, preds = [], []
    for X in block_data:
        pts.append(np.asarray(X, dtype=np.float64))
        preds.append(classify_block(model, X, min_class_confidence, unclassified))
    if not pts:
        return np.zeros((0, 0)), np.zeros((0, 1), dtype=np.int64)
    return np.vstack(pts), np.vstack(preds)
