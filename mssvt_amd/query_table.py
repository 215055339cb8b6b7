"""Integer offset tables of the chessboard / mixed-scale window sampling (host, ctor time).

Interface of ``MixedScaleSparseTransformerBlock.get_vox_query_table``
(ref: pcdet/models/backbones_3d/mssvt_backbone.py:73-122): offsets of every cell of
the large (key) window relative to its centre cell, sorted by Chebyshev distance,
split into

  odd   -- win1 cells whose x AND y offsets are odd  (python modulo: -1 is odd)
  even  -- win1 cells whose x AND y offsets are even
  win1  -- the remaining win1 cells (mixed parity)
  win2  -- cells of the large window outside win1

win1 membership is ``-(w//2) <= o <= w//2 + (1 - w%2)`` per axis, which for an EVEN
``w`` spans ``w+1`` cells (windows then overlap) -- a reference quirk kept as is.

The reference orders equal-distance cells with an unstable device sort, so its tie
order is not defined; here it is a STABLE sort over the x-major enumeration, which
makes list truncation and the FPS seed deterministic.  Tables can also be supplied
explicitly (``MixedScaleSparseTransformerBlock.set_vox_query_table``), e.g. to
reproduce tables exported from a reference run.
"""
import numpy as np


def chebyshev_sorted_offsets(size):
    ax = [np.arange(s) - s // 2 for s in size]
    xyz = np.stack(np.meshgrid(*ax, indexing="ij"), axis=-1).reshape(-1, 3)
    order = np.argsort(np.abs(xyz).max(axis=1), kind="stable")
    return xyz[order]


def vox_query_table(win1_size, win2_size=None, cbs_mode="odd_even"):
    """-> (dict of (n,3) int32 arrays, max_num_odd, max_num_even)."""
    if win2_size is None:
        return {"win1": chebyshev_sorted_offsets(win1_size).astype(np.int32)}, None, None
    if any((win2_size[i] - win1_size[i]) % 2 for i in range(3)):
        raise AssertionError("win2 - win1 must be even on every axis")
    if cbs_mode != "odd_even":
        raise NotImplementedError(cbs_mode)
    xyz = chebyshev_sorted_offsets(win2_size)
    lo = np.array([-(w // 2) for w in win1_size])
    hi = np.array([w // 2 + (1 - w % 2) for w in win1_size])
    in1 = ((xyz >= lo) & (xyz <= hi)).all(axis=1)
    w1 = xyz[in1]
    px, py = w1[:, 0] % 2, w1[:, 1] % 2
    odd, even = (px == 1) & (py == 1), (px == 0) & (py == 0)
    tab = {"odd": w1[odd], "even": w1[even], "win1": w1[~(odd | even)], "win2": xyz[~in1]}
    return {k: np.ascontiguousarray(v, dtype=np.int32) for k, v in tab.items()}, int(odd.sum()), int(even.sum())
