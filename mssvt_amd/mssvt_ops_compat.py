"""Stand-in for the reference's pybind module ``pcdet.ops.mssvt.mssvt_ops_cuda`` (ref: pcdet/ops/mssvt/src/ms_api.cpp:7-14).

Same function names, argument order and in-place output convention as the six pybind functions (signatures:
ms_sparse_attention_gpu.h:10-67, group_features_gpu.h:17-26), implemented by ``libmssvt_hip.so`` through its C ABI
(include/mssvt_hip.h part 1a).  With

    # pcdet/ops/mssvt/mssvt_ops.py:4
    from mssvt_amd import mssvt_ops_compat as mssvt_ops_cuda

the reference's own ``mssvt_ops.py`` runs unchanged on MI355X.  Differences the C ABI needs are derived here:
``batch_size`` from the table / count tensors, a scratch workspace, torch's current stream.  Errors raise
(``MssvtHipError``) instead of ``exit(-1)``; like the reference, every function returns 1.
"""
import ctypes

import torch

from . import _lib

_i = ctypes.c_int


def _ws(num_voxels, batch_size, device):
    n = int(_lib.lib().mssvt_hash_workspace_ints(_i(int(num_voxels)), _i(int(batch_size))))
    return torch.empty(n, dtype=torch.int32, device=device)


def build_mapping_with_hash_wrapper(x_max, y_max, z_max, num_voxels, hash_size, v_indices, v_bs_cnt, xyz_to_vidx):
    B = int(xyz_to_vidx.shape[0])
    _lib.call("mssvt_build_mapping_with_hash", _i(x_max), _i(y_max), _i(z_max), _i(num_voxels), _i(hash_size), _i(B),
              _lib.ptr(v_indices), _lib.ptr(v_bs_cnt), _lib.ptr(xyz_to_vidx), _lib.ptr(_ws(num_voxels, B, v_indices.device)),
              _lib.stream())
    return 1


def window_with_hash_wrapper(x_wgs, y_wgs, z_wgs, x_ws, y_ws, z_ws, num_voxels, num_windows, hash_size, v_indices,
                             w_indices, xyz_to_vidx, vcount):
    B = int(xyz_to_vidx.shape[0])
    _lib.call("mssvt_window_with_hash", _i(x_wgs), _i(y_wgs), _i(z_wgs), _i(x_ws), _i(y_ws), _i(z_ws), _i(num_voxels),
              _i(num_windows), _i(hash_size), _i(B), _lib.ptr(v_indices), _lib.ptr(w_indices), _lib.ptr(xyz_to_vidx),
              _lib.ptr(vcount), _lib.ptr(_ws(num_voxels, B, v_indices.device)), _lib.stream())
    return 1


def gather_two_window_voxels_with_hash_wrapper(x_max, y_max, z_max, x_ws, y_ws, z_ws, max_num_odd, max_num_even,
                                               max_num_win1, max_num_win2, num_wins, hash_size, num_odd, num_even,
                                               num_win1, num_win2, vox_ind_odd, vox_ind_even, vox_ind_win1,
                                               vox_ind_win2, vox_coord_odd, vox_coord_even, vox_coord_win1,
                                               vox_coord_win2, vox_query_odd, vox_query_even, vox_query_win1,
                                               vox_query_win2, v_indices, xyz_to_vidx):
    _lib.call("mssvt_gather_two_window_voxels_with_hash", _i(x_max), _i(y_max), _i(z_max), _i(x_ws), _i(y_ws), _i(z_ws),
              _i(max_num_odd), _i(max_num_even), _i(max_num_win1), _i(max_num_win2), _i(num_wins), _i(hash_size),
              _i(num_odd), _i(num_even), _i(num_win1), _i(num_win2), _lib.ptr(vox_ind_odd), _lib.ptr(vox_ind_even),
              _lib.ptr(vox_ind_win1), _lib.ptr(vox_ind_win2), _lib.ptr(vox_coord_odd), _lib.ptr(vox_coord_even),
              _lib.ptr(vox_coord_win1), _lib.ptr(vox_coord_win2), _lib.ptr(vox_query_odd), _lib.ptr(vox_query_even),
              _lib.ptr(vox_query_win1), _lib.ptr(vox_query_win2), _lib.ptr(v_indices), _lib.ptr(xyz_to_vidx),
              _lib.stream())
    return 1


def gather_one_window_voxels_with_hash_wrapper(x_max, y_max, z_max, x_ws, y_ws, z_ws, max_num_win1, num_wins,
                                               hash_size, num_win1, vox_ind_win1, vox_coord_win1, vox_query_win1,
                                               v_indices, xyz_to_vidx):
    _lib.call("mssvt_gather_one_window_voxels_with_hash", _i(x_max), _i(y_max), _i(z_max), _i(x_ws), _i(y_ws), _i(z_ws),
              _i(max_num_win1), _i(num_wins), _i(hash_size), _i(num_win1), _lib.ptr(vox_ind_win1),
              _lib.ptr(vox_coord_win1), _lib.ptr(vox_query_win1), _lib.ptr(v_indices), _lib.ptr(xyz_to_vidx),
              _lib.stream())
    return 1


def group_features_wrapper(B, M, C, nsample, features, features_batch_cnt, idx, idx_batch_cnt, out):
    _lib.call("mssvt_group_features", _i(B), _i(M), _i(C), _i(nsample), _lib.ptr(features), _lib.ptr(features_batch_cnt),
              _lib.ptr(idx), _lib.ptr(idx_batch_cnt), _lib.ptr(out), _lib.stream())
    return 1


def group_features_grad_wrapper(B, M, C, N, nsample, grad_out, idx, idx_batch_cnt, features_batch_cnt, grad_features):
    _lib.call("mssvt_group_features_grad", _i(B), _i(M), _i(C), _i(N), _i(nsample), _lib.ptr(grad_out), _lib.ptr(idx),
              _lib.ptr(idx_batch_cnt), _lib.ptr(features_batch_cnt), _lib.ptr(grad_features), _lib.stream())
    return 1
