"""Operator surface of the MsSVT window ops on MI355X.

Same entry-point names, argument order, shapes, dtypes and padding conventions as
the reference's ``pcdet/ops/mssvt/mssvt_ops.py`` (``build_hash_table`` :26,
``get_non_empty_window_center`` :60, ``gather_two_window_voxels`` :102,
``gather_one_window_voxels`` :133, ``grouping_operation`` :192), so the
reference's call sites work unchanged -- but every buffer is allocated directly
in HBM (the reference fills its tables on the CPU and copies them over each
call), the window list is compacted on the device, and the only host
synchronisation left is the one the data-dependent output shape forces
(``get_non_empty_window_center`` must know ``nw`` to size its result).

All work happens in libmssvt_hip.so (hand-written HIP, include/mssvt_hip.h);
there is no CPU fallback.
"""
import ctypes
import os

import torch

from . import _lib

ST_DUPLICATE_KEY, ST_TABLE_OVERFLOW, ST_WINDOW_OVERFLOW, ST_UNSORTED = 1, 2, 4, 8

# The reference asserts ``features.shape[0] == features_batch_cnt.sum()`` (mssvt_ops.py:157-160),
# which costs two host syncs per call; here those checks are opt-in.
CHECK_COUNTS = os.environ.get("MSSVT_CHECK_COUNTS", "0") == "1"

_i = ctypes.c_int


def _ints(xs):
    return [int(v) for v in xs]



class FillArena(object):
    """One -1-filled int32 buffer per forward, carved into the tables / owner arrays / list prefills the
    path needs (each would otherwise be its own fill launch, ~4 us of GPU time and a framework call), with a
    ZEROED tail for the level's status words / occupancy bitmap: one launch (mssvt_fill_two) for both.
    Sized from the previous forward's demand; a request that does not fit falls back to its own fill."""
    current = None  # set by MixedScaleSparseTransformer.forward around the fused path

    def __init__(self, numel, device, zero_numel=0):
        numel = (max(int(numel), 1) + 63) // 64 * 64
        zero_numel = (max(int(zero_numel), 0) + 63) // 64 * 64
        dev = torch.device(device)
        if dev.type == "cuda":
            whole = torch.empty(numel + zero_numel, dtype=torch.int32, device=dev)
            _lib.call("mssvt_fill_two", _lib.ptr_fast(whole), ctypes.c_longlong(numel), _i(-1),
                      ctypes.c_void_p(whole.data_ptr() + 4 * numel), ctypes.c_longlong(zero_numel), _i(0), _lib.stream())
            self.buf, self.zeros = whole[:numel], whole[numel:]
        else:
            self.buf = torch.full((numel,), -1, dtype=torch.int32, device=dev)
            self.zeros = torch.zeros((zero_numel,), dtype=torch.int32, device=dev)
        self.used = self.zero_used = 0
        self.demand = self.zero_demand = 0

    def take(self, shape):
        n = 1
        for d in shape:
            n *= int(d)
        n_al = (n + 63) // 64 * 64  # keep every piece 256-byte aligned
        self.demand += n_al
        if self.used + n_al > self.buf.numel():
            return None
        out = self.buf[self.used:self.used + n].view(*shape)
        self.used += n_al
        return out

    def take_zero(self, n):
        """n zeroed ints (256-byte aligned) or None when the zero tail is exhausted."""
        n_al = (int(n) + 63) // 64 * 64
        self.zero_demand += n_al
        if self.zero_used + n_al > self.zeros.numel():
            return None
        out = self.zeros[self.zero_used:self.zero_used + int(n)]
        self.zero_used += n_al
        return out


def full_neg1(shape, device):
    """int32 tensor of -1s: a slice of the current forward's FillArena when there is one."""
    a = FillArena.current
    if a is not None and a.buf.device == torch.device(device):
        t = a.take(shape)
        if t is not None:
            return t
    return torch.full(tuple(shape), -1, dtype=torch.int32, device=device)


def hash_workspace(num_voxels, batch_size, device):
    n = int(_lib.lib().mssvt_hash_workspace_ints(_i(int(num_voxels)), _i(int(batch_size))))
    return torch.empty(n, dtype=torch.int32, device=device)


def _check_int32(t, name):
    assert t.dtype == torch.int32, "%s must be int32" % name
    assert t.is_contiguous(), "%s must be contiguous" % name


def build_hash_table(batch_size, hash_size, spatial_shape, voxel_indices, v_bs_cnt, workspace=None):
    """(B, H, 2) int32 table key -> voxel index within its sample.

    ref: BuildHashTable.forward, mssvt_ops.py:10-20.  Layout = sequential insertion
    in voxel-index order (deterministic; see include/mssvt_hip.h)."""
    x_max, y_max, z_max = _ints(spatial_shape)
    _check_int32(voxel_indices, "voxel_indices")
    v_bs_cnt = v_bs_cnt.to(device=voxel_indices.device, dtype=torch.int32).contiguous()
    n = voxel_indices.shape[0]
    table = full_neg1((batch_size, hash_size, 2), voxel_indices.device)
    ws = workspace if workspace is not None else hash_workspace(n, batch_size, voxel_indices.device)
    _lib.call("mssvt_build_mapping_with_hash", _i(x_max), _i(y_max), _i(z_max), _i(n),
              _i(int(hash_size)), _i(int(batch_size)), _lib.ptr(voxel_indices), _lib.ptr(v_bs_cnt),
              _lib.ptr(table), _lib.ptr(ws), _lib.stream())
    build_hash_table.last_status = ws[0:1]  # device word: ST_* bits (read lazily by the fused path)
    return table


build_hash_table.last_status = None


def window_partition_device(win_size, max_num_wins, batch_size, hash_size, spatial_shape,
                            voxel_indices, workspace=None):
    """Device-resident window discovery: returns (win_ind_padded (N,4), table (B,H,2),
    k_bs_cnt (B), workspace) without synchronising; ``workspace[1]`` holds nw and
    ``workspace[0]`` the status bits."""
    x_ws, y_ws, z_ws = _ints(win_size)
    x_wgs, y_wgs, z_wgs = _ints(spatial_shape)
    _check_int32(voxel_indices, "voxel_indices")
    dev = voxel_indices.device
    n = voxel_indices.shape[0]
    table = full_neg1((batch_size, hash_size, 2), dev)
    win = torch.empty((max(n, 1), 4), dtype=torch.int32, device=dev)
    vcount = torch.empty(batch_size, dtype=torch.int32, device=dev)  # every entry is written by the scan kernel
    ws = workspace if workspace is not None else hash_workspace(n, batch_size, dev)
    _lib.call("mssvt_window_partition_compact", _i(x_wgs), _i(y_wgs), _i(z_wgs), _i(x_ws), _i(y_ws),
              _i(z_ws), _i(n), _i(int(max_num_wins)), _i(int(hash_size)), _i(int(batch_size)),
              _lib.ptr(voxel_indices), _lib.ptr(win), _lib.ptr(table), _lib.ptr(vcount), _lib.ptr(ws),
              _lib.stream())
    return win, table, vcount, ws


def window_partitions_device(win_sizes, max_num_wins, batch_size, hash_size, spatial_shapes, voxel_indices):
    """Several partitions of one voxel list in the same launches (``mssvt_window_partition_multi``):
    a list of (win_ind_padded, table, k_bs_cnt, workspace) as ``window_partition_device`` returns them."""
    k = len(win_sizes)
    _check_int32(voxel_indices, "voxel_indices")
    dev = voxel_indices.device
    n = voxel_indices.shape[0]
    tables = [full_neg1((batch_size, hash_size, 2), dev) for _ in range(k)]
    scratch = [full_neg1((batch_size, hash_size, 2), dev) for _ in range(k)]
    wins = [torch.empty((max(n, 1), 4), dtype=torch.int32, device=dev) for _ in range(k)]
    vcounts = torch.empty((k, batch_size), dtype=torch.int32, device=dev)
    stride = int(_lib.lib().mssvt_hash_workspace_ints(_i(int(n)), _i(int(batch_size))))
    ws = torch.empty((k, stride), dtype=torch.int32, device=dev)
    ints = lambda rows: (ctypes.c_int * (3 * k))(*[int(v) for r in rows for v in r])  # noqa: E731
    ptrs = lambda ts: (ctypes.c_void_p * k)(*[t.data_ptr() for t in ts])  # noqa: E731
    _lib.call("mssvt_window_partition_multi", _i(k), ints(spatial_shapes), ints(win_sizes),
              (ctypes.c_int * k)(*[int(m) for m in max_num_wins]), _i(n), _i(int(hash_size)), _i(int(batch_size)),
              _lib.ptr(voxel_indices), ptrs(wins), ptrs(tables), ptrs(scratch), ptrs([vcounts[i] for i in range(k)]),
              _lib.ptr(ws), ctypes.c_longlong(stride), _lib.stream())
    return [(wins[i], tables[i], vcounts[i], ws[i]) for i in range(k)]


def get_non_empty_window_center(win_size, max_num_wins, batch_size, hash_size, spatial_shape,
                                voxel_indices):
    """-> (win_ind (nw,4) int32 [b,wz,wy,wx], window table (B,H,2) int32).

    ref: WindowPartition.forward, mssvt_ops.py:31-54.  ``spatial_shape`` is the window
    grid (``spatial_shape // win_size``).  Windows are numbered by first occurrence in
    voxel-index order per sample (the reference's order depends on thread timing)."""
    win, table, _, ws = window_partition_device(win_size, max_num_wins, batch_size, hash_size,
                                                spatial_shape, voxel_indices)
    status, nw = ws[:2].tolist()  # the one sync the output shape needs
    if status & ST_WINDOW_OVERFLOW:
        raise _lib.MssvtHipError("a sample has more than max_num_wins=%d non-empty windows "
                                 "(the reference writes out of bounds here)" % max_num_wins)
    if status & ST_TABLE_OVERFLOW:
        raise _lib.MssvtHipError("window hash table overflow: more windows than hash_size=%d" % hash_size)
    return win[:nw].contiguous(), table


def gather_two_window_voxels(spatial_shape, win_size, max_num_odd, max_num_even, max_num_win1,
                             max_num_win2, vox_query_odd, vox_query_even, vox_query_win1,
                             vox_query_win2, win_indices, dense_map):
    """-> 4 index tensors (nw, max_num_*) int32 (-1 padded) + 4 offset tensors
    (nw, max_num_*, 3) int32 (0 padded), for the odd / even / win1 / win2 lists.

    ref: GatherTwoWindowVoxels.forward, mssvt_ops.py:66-96."""
    x_max, y_max, z_max = _ints(spatial_shape)
    x_ws, y_ws, z_ws = _ints(win_size)
    _, hash_size, _ = dense_map.shape
    _check_int32(win_indices, "win_indices")
    dev = win_indices.device
    nw = win_indices.shape[0]
    tabs = [t.to(device=dev, dtype=torch.int32).contiguous()
            for t in (vox_query_odd, vox_query_even, vox_query_win1, vox_query_win2)]
    maxes = _ints((max_num_odd, max_num_even, max_num_win1, max_num_win2))
    inds = [torch.full((nw, m), -1, dtype=torch.int32, device=dev) for m in maxes]
    coords = [torch.zeros((nw, m, 3), dtype=torch.int32, device=dev) for m in maxes]
    _lib.call("mssvt_gather_two_window_voxels_with_hash", _i(x_max), _i(y_max), _i(z_max), _i(x_ws),
              _i(y_ws), _i(z_ws), *[_i(m) for m in maxes], _i(nw), _i(int(hash_size)),
              *[_i(t.shape[0]) for t in tabs], *[_lib.ptr(t) for t in inds],
              *[_lib.ptr(t) for t in coords], *[_lib.ptr(t) for t in tabs], _lib.ptr(win_indices),
              _lib.ptr(dense_map), _lib.stream())
    return (*inds, *coords)


def gather_one_window_voxels(spatial_shape, win_size, max_num_win1, vox_query_win1, win_indices,
                             dense_map):
    """-> (vox_ind_win1 (nw,max_num_win1), vox_coord_win1 (nw,max_num_win1,3)).

    ref: GatherOneWindowVoxels.forward, mssvt_ops.py:108-127."""
    x_max, y_max, z_max = _ints(spatial_shape)
    x_ws, y_ws, z_ws = _ints(win_size)
    _, hash_size, _ = dense_map.shape
    _check_int32(win_indices, "win_indices")
    dev = win_indices.device
    nw = win_indices.shape[0]
    tab = vox_query_win1.to(device=dev, dtype=torch.int32).contiguous()
    ind = torch.full((nw, int(max_num_win1)), -1, dtype=torch.int32, device=dev)
    coord = torch.zeros((nw, int(max_num_win1), 3), dtype=torch.int32, device=dev)
    _lib.call("mssvt_gather_one_window_voxels_with_hash", _i(x_max), _i(y_max), _i(z_max), _i(x_ws),
              _i(y_ws), _i(z_ws), _i(int(max_num_win1)), _i(nw), _i(int(hash_size)), _i(tab.shape[0]),
              _lib.ptr(ind), _lib.ptr(coord), _lib.ptr(tab), _lib.ptr(win_indices), _lib.ptr(dense_map),
              _lib.stream())
    return ind, coord


class _GroupFeatures(torch.autograd.Function):
    """Stacked-batch row gather, differentiable w.r.t. ``features``.

    ref: GroupingOperation, mssvt_ops.py:136-190 (forward K5, backward K6)."""

    @staticmethod
    def forward(ctx, features, features_batch_cnt, idx, idx_batch_cnt):
        assert features.is_contiguous() and idx.is_contiguous()
        assert features.dtype == torch.float32
        _check_int32(idx, "idx")
        features_batch_cnt = features_batch_cnt.to(torch.int32).contiguous()
        idx_batch_cnt = idx_batch_cnt.to(torch.int32).contiguous()
        M, nsample = idx.shape
        N, C = features.shape
        B = idx_batch_cnt.shape[0]
        if CHECK_COUNTS:
            assert N == int(features_batch_cnt.sum()), (features.shape, features_batch_cnt)
            assert M == int(idx_batch_cnt.sum()), (idx.shape, idx_batch_cnt)
        out = torch.zeros((M, C, nsample), dtype=torch.float32, device=features.device)
        _lib.call("mssvt_group_features", _i(B), _i(M), _i(C), _i(nsample), _lib.ptr(features),
                  _lib.ptr(features_batch_cnt), _lib.ptr(idx), _lib.ptr(idx_batch_cnt), _lib.ptr(out),
                  _lib.stream())
        ctx.for_backwards = (B, N, idx, features_batch_cnt, idx_batch_cnt)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        B, N, idx, features_batch_cnt, idx_batch_cnt = ctx.for_backwards
        M, C, nsample = grad_out.shape
        grad_out = grad_out.contiguous()
        grad = torch.zeros((N, C), dtype=torch.float32, device=grad_out.device)
        _lib.call("mssvt_group_features_grad", _i(B), _i(M), _i(C), _i(N), _i(nsample),
                  _lib.ptr(grad_out), _lib.ptr(idx), _lib.ptr(idx_batch_cnt),
                  _lib.ptr(features_batch_cnt), _lib.ptr(grad), _lib.stream())
        return grad, None, None, None


def grouping_operation(features, features_batch_cnt, idx, idx_batch_cnt):
    """features (N1+N2.., C), idx (M1+M2.., nsample) with per-sample indices (<0 = empty
    slot) -> (M, C, nsample); empty slots stay 0.  ref: mssvt_ops.py:139-170."""
    return _GroupFeatures.apply(features, features_batch_cnt, idx, idx_batch_cnt)
