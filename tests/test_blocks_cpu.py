"""Host logic of the block formats either side of the path; no GPU needed."""
import numpy as np
import torch


def test_read_block_splits_the_label_column(tmp_path):
    from ahn3_dgcnn_pytorch_amd import blocks
    tile = np.random.RandomState(0).rand(50, 5)
    tile[:, -1] = np.arange(50) % 5
    f = tmp_path / "Area_1_block.npy"
    np.save(f, tile)
    pts, lab = blocks.read_block(str(f))
    assert pts.shape == (50, 4) and lab.shape == (50,)
    assert np.array_equal(np.asarray(pts), tile[:, :4]) and np.array_equal(np.asarray(lab), tile[:, 4])


def test_collate_subsamples_to_the_smallest_tile_with_the_reference_rng_calls():
    from ahn3_dgcnn_pytorch_amd import blocks
    rs = np.random.RandomState(1)
    tiles = [(rs.rand(n, 4), rs.randint(0, 5, n).astype(np.float64), i) for i, n in enumerate((40, 25, 33))]
    np.random.seed(7)
    pts, lab, idxs = blocks.collate_pcs(tiles)
    assert pts.shape == (3, 25, 4) and pts.dtype == torch.float64
    assert lab.shape == (3, 25) and lab.dtype == torch.float64
    assert idxs.tolist() == [0, 1, 2]
    # the draws data.py:175-177 makes under the same seed
    np.random.seed(7)
    for i, (p, l, _) in enumerate(tiles):
        keep = np.random.choice(range(p.shape[0]), 25, replace=False)
        assert np.array_equal(pts[i].numpy(), p[keep]) and np.array_equal(lab[i].numpy(), l[keep])
