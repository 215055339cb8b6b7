"""Synthetic training input of the detector for bench.py / the tests: Waymo-shaped point clouds (mssvt_amd/synthetic.py)
with ground-truth boxes `(B, M, 8)` = [x, y, z, dx, dy, dz, heading, class] as the reference's dataloader collates them
(pcdet/datasets/dataset.py:collate_batch: zero rows pad the samples with fewer boxes)."""
import numpy as np
import torch

from . import synthetic
from .dist import scene_seeds


def make_boxes(batch, num_boxes=12, seed=0, extent=60.0):
    rng = np.random.default_rng(seed)
    gt = np.zeros((batch, num_boxes, 8), np.float32)
    for b in range(batch):
        n = num_boxes - (b % 3)  # ragged: the padding rows stay zero
        gt[b, :n, 0:2] = rng.uniform(-extent, extent, (n, 2))
        gt[b, :n, 2] = rng.uniform(-1.0, 1.0, n)
        gt[b, :n, 3:6] = rng.uniform([1.5, 0.8, 1.0], [5.0, 2.5, 2.0], (n, 3))
        gt[b, :n, 6] = rng.uniform(-3.1, 3.1, n)
        gt[b, :n, 7] = rng.integers(1, 4, n)
    return gt


def make_scene(points, batch, rank, device, frame=0, num_boxes=12):
    """(points (P, 6) [b, x, y, z, intensity, elongation], gt_boxes (B, M, 8)) of this rank's frame `frame`, on `device`."""
    seed0 = scene_seeds(rank, batch)[0] + 7919 * frame
    pts = synthetic.make_batch_points(points, batch, seed0=seed0)
    gt = make_boxes(batch, num_boxes, seed=1000 + seed0)
    return torch.from_numpy(pts).to(device), torch.from_numpy(gt).to(device)
