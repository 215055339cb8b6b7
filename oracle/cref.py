"""ctypes front-end of the CPU oracle (oracle/mssvt_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of mssvt_oracle.c.  Nothing under
``mssvt_amd/`` may import this module; only tests/, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg do.

Two layers:

* ``orc_*`` thin wrappers over the C symbols taking C-contiguous numpy arrays
  that the caller pre-allocates / pre-fills exactly as the reference's Python
  callers do (``pcdet/ops/mssvt/mssvt_ops.py``,
  ``pcdet/ops/pointnet2/pointnet2_batch/pointnet2_utils.py``);
* op-level helpers (``build_hash_table`` ... ``three_nn``) that restate those
  Python callers on numpy arrays (allocation, -1/0 pre-fill, per-sample
  compaction of the window list), each citing the lines it follows.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libmssvt_oracle.so")


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "mssvt_oracle.c"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def set_num_threads(n):
    """Threads of the OpenMP loops of the C oracle (0: the OpenMP default = all cores)."""
    lib().orc_set_num_threads(int(n))


def max_threads():
    return int(lib().orc_get_max_threads())


def _ip(a):
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"], (a.dtype, a.flags)
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))


def _fp(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"], (a.dtype, a.flags)
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# --------------------------------------------------------------------------
# op-level helpers (numpy in / numpy out)
# --------------------------------------------------------------------------

def bs_cnt(indices, batch_size):
    """``with_bs_cnt`` / ``build_map_table`` counts: mssvt_backbone.py:124-130,
    mssvt_utils.py:35-37."""
    return np.array([(indices[:, 0] == i).sum() for i in range(batch_size)], dtype=np.int32)


def build_hash_table(batch_size, hash_size, spatial_shape, voxel_indices, v_bs_cnt):
    """mssvt_ops.py:10-20 (BuildHashTable.forward) -> K1."""
    x_max, y_max, z_max = (int(v) for v in spatial_shape)
    voxel_indices = _i32(voxel_indices)
    table = np.full((batch_size, hash_size, 2), -1, dtype=np.int32)
    lib().orc_build_mapping_with_hash(
        x_max, y_max, z_max, voxel_indices.shape[0], hash_size,
        _ip(voxel_indices), _ip(_i32(v_bs_cnt)), _ip(table))
    return table


def get_non_empty_window_center(win_size, max_num_wins, batch_size, hash_size,
                                spatial_shape, voxel_indices):
    """mssvt_ops.py:31-54 (WindowPartition.forward) -> K2 + per-sample compaction."""
    x_ws, y_ws, z_ws = (int(v) for v in win_size)
    x_max, y_max, z_max = (int(v) for v in spatial_shape)
    voxel_indices = _i32(voxel_indices)
    table = np.full((batch_size, hash_size, 2), -1, dtype=np.int32)
    win_indices = np.full((batch_size, max_num_wins, 3), -1, dtype=np.int32)
    vcount = np.zeros(batch_size, dtype=np.int32)
    rc = lib().orc_window_with_hash(
        x_max, y_max, z_max, x_ws, y_ws, z_ws, voxel_indices.shape[0], max_num_wins,
        hash_size, _ip(voxel_indices), _ip(win_indices), _ip(table), _ip(vcount))
    if rc != 0:
        raise RuntimeError("more than max_num_wins windows in a sample (reference writes OOB here)")
    wins = []
    for i in range(batch_size):
        w = win_indices[i]
        w = w[w[:, 0] >= 0]
        wins.append(np.concatenate([np.full((w.shape[0], 1), i, dtype=np.int32), w], axis=1))
    return np.ascontiguousarray(np.concatenate(wins, axis=0)), table


def gather_two_window_voxels(spatial_shape, win_size, max_num_odd, max_num_even,
                             max_num_win1, max_num_win2, q_odd, q_even, q_win1, q_win2,
                             win_indices, table):
    """mssvt_ops.py:66-96 (GatherTwoWindowVoxels.forward) -> K3."""
    x_max, y_max, z_max = (int(v) for v in spatial_shape)
    x_ws, y_ws, z_ws = (int(v) for v in win_size)
    hash_size = table.shape[1]
    q_odd, q_even, q_win1, q_win2 = _i32(q_odd), _i32(q_even), _i32(q_win1), _i32(q_win2)
    win_indices = _i32(win_indices)
    nw = win_indices.shape[0]
    maxes = (max_num_odd, max_num_even, max_num_win1, max_num_win2)
    inds = [np.full((nw, m), -1, dtype=np.int32) for m in maxes]
    coords = [np.zeros((nw, m, 3), dtype=np.int32) for m in maxes]
    lib().orc_gather_two_window_voxels(
        x_max, y_max, z_max, x_ws, y_ws, z_ws, *maxes, nw, hash_size,
        q_odd.shape[0], q_even.shape[0], q_win1.shape[0], q_win2.shape[0],
        *[_ip(a) for a in inds], *[_ip(a) for a in coords],
        _ip(q_odd), _ip(q_even), _ip(q_win1), _ip(q_win2), _ip(win_indices), _ip(_i32(table)))
    return (*inds, *coords)


def gather_one_window_voxels(spatial_shape, win_size, max_num_win1, q_win1, win_indices, table):
    """mssvt_ops.py:108-127 (GatherOneWindowVoxels.forward) -> K4."""
    x_max, y_max, z_max = (int(v) for v in spatial_shape)
    x_ws, y_ws, z_ws = (int(v) for v in win_size)
    hash_size = table.shape[1]
    q_win1 = _i32(q_win1)
    win_indices = _i32(win_indices)
    nw = win_indices.shape[0]
    ind = np.full((nw, max_num_win1), -1, dtype=np.int32)
    coord = np.zeros((nw, max_num_win1, 3), dtype=np.int32)
    lib().orc_gather_one_window_voxels(
        x_max, y_max, z_max, x_ws, y_ws, z_ws, max_num_win1, nw, hash_size, q_win1.shape[0],
        _ip(ind), _ip(coord), _ip(q_win1), _ip(win_indices), _ip(_i32(table)))
    return ind, coord


def grouping_operation(features, features_batch_cnt, idx, idx_batch_cnt):
    """mssvt_ops.py:139-170 (GroupingOperation.forward) -> K5. (M,C,nsample)."""
    features, idx = _f32(features), _i32(idx)
    features_batch_cnt, idx_batch_cnt = _i32(features_batch_cnt), _i32(idx_batch_cnt)
    assert features.shape[0] == features_batch_cnt.sum()
    assert idx.shape[0] == idx_batch_cnt.sum()
    M, nsample = idx.shape
    _, C = features.shape
    out = np.zeros((M, C, nsample), dtype=np.float32)
    lib().orc_group_features(idx_batch_cnt.shape[0], M, C, nsample, _fp(features),
                             _ip(features_batch_cnt), _ip(idx), _ip(idx_batch_cnt), _fp(out))
    return out


def grouping_operation_grad(grad_out, N, idx, idx_batch_cnt, features_batch_cnt):
    """mssvt_ops.py:173-190 (GroupingOperation.backward) -> K6."""
    grad_out, idx = _f32(grad_out), _i32(idx)
    M, C, nsample = grad_out.shape
    grad = np.zeros((N, C), dtype=np.float32)
    lib().orc_group_features_grad(len(idx_batch_cnt), M, C, N, nsample, _fp(grad_out), _ip(idx),
                                  _ip(_i32(idx_batch_cnt)), _ip(_i32(features_batch_cnt)), _fp(grad))
    return grad


def farthest_point_sample(xyz, npoint):
    """pointnet2_utils.py:12-29 (FarthestPointSampling.forward) -> K7."""
    xyz = _f32(xyz)
    B, N, _ = xyz.shape
    out = np.zeros((B, npoint), dtype=np.int32)
    temp = np.full((B, N), 1e10, dtype=np.float32)
    lib().orc_farthest_point_sampling(B, N, npoint, _fp(xyz), _fp(temp), _ip(out))
    return out


def gather_operation(features, idx):
    """pointnet2_utils.py:42-60 (GatherOperation.forward) -> K8. (B,C,npoint)."""
    features, idx = _f32(features), _i32(idx)
    B, npoint = idx.shape
    _, C, N = features.shape
    out = np.zeros((B, C, npoint), dtype=np.float32)
    lib().orc_gather_points(B, C, N, npoint, _fp(features), _ip(idx), _fp(out))
    return out


def three_nn(unknown, known):
    """pointnet2_utils.py:79-99 (ThreeNN.forward) -> K9; returns (sqrt(dist2), idx)."""
    unknown, known = _f32(unknown), _f32(known)
    B, N, _ = unknown.shape
    m = known.shape[1]
    dist2 = np.zeros((B, N, 3), dtype=np.float32)
    idx = np.zeros((B, N, 3), dtype=np.int32)
    lib().orc_three_nn(B, N, m, _fp(unknown), _fp(known), _fp(dist2), _ip(idx))
    return np.sqrt(dist2), idx


def group_points(features, idx):
    """pointnet2_utils.py:159-177 (GroupingOperation.forward) -> K10. (B,C,np,ns)."""
    features, idx = _f32(features), _i32(idx)
    B, npts, ns = idx.shape
    _, C, N = features.shape
    out = np.zeros((B, C, npts, ns), dtype=np.float32)
    lib().orc_group_points(B, C, N, npts, ns, _fp(features), _ip(idx), _fp(out))
    return out


def opt_n_threads(n):
    return int(lib().orc_opt_n_threads(int(n)))
