"""Device voxelizer: points -> sorted unique voxel coordinates (+ point -> voxel map).

Produces exactly the ``voxel_coords`` / ``unq_inv`` of the reference's ``DynamicVFE.forward``
index path (pcdet/models/backbones_3d/vfe/dynamic_vfe.py:83-93,114-118) -- the input contract
of ``MixedScaleSparseTransformer`` -- with an occupancy bitmap + popcount rank instead of the
radix sort inside ``torch.unique`` (csrc/voxelize.hip).  No CPU fallback.
"""
import ctypes

import torch

from . import _lib

_i = ctypes.c_int


def _f3(xs):
    return (ctypes.c_float * 3)(*[float(v) for v in xs])


@torch.no_grad()
def voxelize(points, point_cloud_range, voxel_size, grid_size, batch_size, capacity=None):
    """points (P, >=4) f32 cuda rows [b, x, y, z, ...].

    Returns (voxel_coords (N,4) int32 [b,z,y,x] sorted by (b,x,y,z), point_voxel (P,) int32 with -1
    for points outside the grid).  One host sync (N sizes the result); pass ``capacity`` and use
    ``voxelize_device`` to stay asynchronous."""
    coords, pv, n_dev = voxelize_device(points, point_cloud_range, voxel_size, grid_size, batch_size, capacity)
    n = int(n_dev.item())
    if n > coords.shape[0]:
        raise _lib.MssvtHipError("voxel capacity %d too small for %d voxels" % (coords.shape[0], n))
    return coords[:n].contiguous(), pv


@torch.no_grad()
def voxelize_device(points, point_cloud_range, voxel_size, grid_size, batch_size, capacity=None):
    assert points.is_cuda and points.dtype == torch.float32 and points.is_contiguous()
    P, stride = points.shape
    X, Y, Z = (int(v) for v in grid_size)
    cap = int(capacity) if capacity is not None else max(P, 1)  # at most one voxel per point
    lib = _lib.lib()
    lib.mssvt_voxelize_workspace_ints.restype = ctypes.c_longlong
    ws = torch.empty(int(lib.mssvt_voxelize_workspace_ints(_i(batch_size), _i(X), _i(Y), _i(Z))),
                     dtype=torch.int32, device=points.device)
    coords = torch.empty((cap, 4), dtype=torch.int32, device=points.device)
    pv = torch.empty(max(P, 1), dtype=torch.int32, device=points.device)
    n_dev = torch.zeros(1, dtype=torch.int32, device=points.device)
    _lib.call("mssvt_voxelize", _lib.ptr(points), _i(stride), ctypes.c_longlong(P), _i(batch_size),
              _f3(point_cloud_range[0:3]), _f3(voxel_size), _i(X), _i(Y), _i(Z), _i(cap), _lib.ptr(coords),
              _lib.ptr(pv), _lib.ptr(n_dev), _lib.ptr(ws), _lib.stream())
    return coords, pv[:P], n_dev
