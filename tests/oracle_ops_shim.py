"""CPU stand-ins for the HIP operator modules, backed by the oracle (TESTS ONLY).

``-m "not gpu"`` tests use this to exercise the HOST logic of the drop-in module
(``forward_ops`` orchestration, state-dict layout, config handling) in the CPU-only
container: the functions of ``mssvt_amd.mssvt_ops`` / ``mssvt_amd.pointnet2_utils`` are
monkeypatched with oracle-backed equivalents on CPU tensors.  The product never does
this; on a GPU box the real HIP library is used.
"""
import numpy as np
import torch

from oracle import cref


def _n(t):
    return t.detach().cpu().numpy() if torch.is_tensor(t) else t


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def install(monkeypatch):
    from mssvt_amd import mssvt_ops, pointnet2_utils

    def build_hash_table(batch_size, hash_size, spatial_shape, voxel_indices, v_bs_cnt, workspace=None):
        return _t(cref.build_hash_table(batch_size, hash_size, spatial_shape, _n(voxel_indices), _n(v_bs_cnt)))

    def get_non_empty_window_center(win_size, max_num_wins, batch_size, hash_size, spatial_shape, voxel_indices):
        a, b = cref.get_non_empty_window_center(win_size, max_num_wins, batch_size, hash_size, spatial_shape,
                                                _n(voxel_indices))
        return _t(a), _t(b)

    def gather_two_window_voxels(*a):
        return tuple(_t(x) for x in cref.gather_two_window_voxels(*[_n(x) for x in a]))

    def gather_one_window_voxels(*a):
        return tuple(_t(x) for x in cref.gather_one_window_voxels(*[_n(x) for x in a]))

    def grouping_operation(features, fcnt, idx, icnt):
        return _t(cref.grouping_operation(_n(features), _n(fcnt), _n(idx), _n(icnt)))

    for f in (build_hash_table, get_non_empty_window_center, gather_two_window_voxels,
              gather_one_window_voxels, grouping_operation):
        monkeypatch.setattr(mssvt_ops, f.__name__, f)

    def farthest_point_sample(xyz, npoint):
        return _t(cref.farthest_point_sample(_n(xyz), npoint))

    def three_nn(unknown, known):
        d, i = cref.three_nn(_n(unknown), _n(known))
        return _t(d), _t(i)

    def pn2_grouping_operation(features, idx):
        return _t(cref.group_points(_n(features), _n(idx)))

    monkeypatch.setattr(pointnet2_utils, "farthest_point_sample", farthest_point_sample)
    monkeypatch.setattr(pointnet2_utils, "three_nn", three_nn)
    monkeypatch.setattr(pointnet2_utils, "grouping_operation", pn2_grouping_operation)
