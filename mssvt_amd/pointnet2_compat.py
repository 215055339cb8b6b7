"""Stand-in for the reference's pybind module ``pointnet2_batch_cuda`` -- the six functions the MsSVT path uses
(ref: pcdet/ops/pointnet2/pointnet2_batch/src/pointnet2_api.cpp:10-24; signatures sampling_gpu.h:9-24,
interpolate_gpu.h:10-11, group_points_gpu.h:10-17), on ``libmssvt_hip.so`` (include/mssvt_hip.h part 1b):

    # pcdet/ops/pointnet2/pointnet2_batch/pointnet2_utils.py:7
    from mssvt_amd import pointnet2_compat as pointnet2

Same names, argument order and in-place outputs.  ``ball_query`` / ``three_interpolate`` are not on the path
 and raise ``NotImplementedError``.
"""
import ctypes

from . import _lib

_i = ctypes.c_int


def farthest_point_sampling_wrapper(b, n, m, points, temp, idx):
    _lib.call("mssvt_farthest_point_sampling", _i(b), _i(n), _i(m), _lib.ptr(points), _lib.ptr(temp), _lib.ptr(idx),
              _lib.stream())
    return 1


def gather_points_wrapper(b, c, n, npoints, points, idx, out):
    _lib.call("mssvt_gather_points", _i(b), _i(c), _i(n), _i(npoints), _lib.ptr(points), _lib.ptr(idx), _lib.ptr(out),
              _lib.stream())
    return 1


def gather_points_grad_wrapper(b, c, n, npoints, grad_out, idx, grad_points):
    _lib.call("mssvt_gather_points_grad", _i(b), _i(c), _i(n), _i(npoints), _lib.ptr(grad_out), _lib.ptr(idx),
              _lib.ptr(grad_points), _lib.stream())
    return 1


def three_nn_wrapper(b, n, m, unknown, known, dist2, idx):
    _lib.call("mssvt_three_nn", _i(b), _i(n), _i(m), _lib.ptr(unknown), _lib.ptr(known), _lib.ptr(dist2), _lib.ptr(idx),
              _lib.stream())


def group_points_wrapper(b, c, n, npoints, nsample, points, idx, out):
    _lib.call("mssvt_group_points", _i(b), _i(c), _i(n), _i(npoints), _i(nsample), _lib.ptr(points), _lib.ptr(idx),
              _lib.ptr(out), _lib.stream())
    return 1


def group_points_grad_wrapper(b, c, n, npoints, nsample, grad_out, idx, grad_points):
    _lib.call("mssvt_group_points_grad", _i(b), _i(c), _i(n), _i(npoints), _i(nsample), _lib.ptr(grad_out), _lib.ptr(idx),
              _lib.ptr(grad_points), _lib.stream())
    return 1


def _not_on_the_path(*_args, **_kwargs):
    raise NotImplementedError("not used by the MsSVT backbone path (SURVEY.md section 8a)")


ball_query_wrapper = three_interpolate_wrapper = three_interpolate_grad_wrapper = _not_on_the_path
