"""The four PointNet++ batch ops the MsSVT block uses, on MI355X.

Same names and signatures as the reference's
``pcdet/ops/pointnet2/pointnet2_batch/pointnet2_utils.py``
(``farthest_point_sample`` :36, ``gather_operation`` :73, ``three_nn`` :105,
``grouping_operation`` :197).  Backed by libmssvt_hip.so; no CPU fallback.
"""
import ctypes

import torch

from . import _lib

_i = ctypes.c_int


def farthest_point_sample(xyz, npoint):
    """xyz (B, N, 3) f32 -> (B, npoint) int32, starting from index 0.

    ref: FarthestPointSampling.forward, pointnet2_utils.py:12-29.  Arg-max ties are
    resolved exactly as the reference's CUDA block resolves them."""
    assert xyz.is_contiguous() and xyz.dtype == torch.float32
    B, N, _ = xyz.shape
    out = torch.empty((B, npoint), dtype=torch.int32, device=xyz.device)
    temp = torch.full((B, N), 1e10, dtype=torch.float32, device=xyz.device)
    _lib.call("mssvt_farthest_point_sampling", _i(B), _i(N), _i(int(npoint)), _lib.ptr(xyz),
              _lib.ptr(temp), _lib.ptr(out), _lib.stream())
    return out


furthest_point_sample = farthest_point_sample


class _GatherPoints(torch.autograd.Function):
    """ref: GatherOperation, pointnet2_utils.py:39-70."""

    @staticmethod
    def forward(ctx, features, idx):
        assert features.is_contiguous() and idx.is_contiguous()
        B, npoint = idx.shape
        _, C, N = features.shape
        out = torch.empty((B, C, npoint), dtype=torch.float32, device=features.device)
        _lib.call("mssvt_gather_points", _i(B), _i(C), _i(N), _i(npoint), _lib.ptr(features),
                  _lib.ptr(idx), _lib.ptr(out), _lib.stream())
        ctx.for_backwards = (idx, C, N)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, C, N = ctx.for_backwards
        B, npoint = idx.shape
        grad = torch.zeros((B, C, N), dtype=torch.float32, device=grad_out.device)
        grad_out = grad_out.contiguous()
        _lib.call("mssvt_gather_points_grad", _i(B), _i(C), _i(N), _i(npoint), _lib.ptr(grad_out),
                  _lib.ptr(idx), _lib.ptr(grad), _lib.stream())
        return grad, None


def gather_operation(features, idx):
    """features (B, C, N) f32, idx (B, npoint) int32 -> (B, C, npoint)."""
    return _GatherPoints.apply(features, idx)


def three_nn(unknown, known):
    """unknown (B, N, 3), known (B, M, 3) -> (dist (B,N,3) = sqrt of squared distance,
    idx (B,N,3) int32).  ref: ThreeNN.forward, pointnet2_utils.py:79-99."""
    assert unknown.is_contiguous() and known.is_contiguous()
    B, N, _ = unknown.shape
    m = known.shape[1]
    dist2 = torch.empty((B, N, 3), dtype=torch.float32, device=unknown.device)
    idx = torch.empty((B, N, 3), dtype=torch.int32, device=unknown.device)
    _lib.call("mssvt_three_nn", _i(B), _i(N), _i(m), _lib.ptr(unknown), _lib.ptr(known),
              _lib.ptr(dist2), _lib.ptr(idx), _lib.stream())
    return torch.sqrt(dist2), idx


class _GroupPoints(torch.autograd.Function):
    """ref: GroupingOperation, pointnet2_utils.py:156-194."""

    @staticmethod
    def forward(ctx, features, idx):
        assert features.is_contiguous() and idx.is_contiguous()
        B, npts, nsample = idx.shape
        _, C, N = features.shape
        out = torch.empty((B, C, npts, nsample), dtype=torch.float32, device=features.device)
        _lib.call("mssvt_group_points", _i(B), _i(C), _i(N), _i(npts), _i(nsample),
                  _lib.ptr(features), _lib.ptr(idx), _lib.ptr(out), _lib.stream())
        ctx.for_backwards = (idx, N)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, N = ctx.for_backwards
        B, C, npts, nsample = grad_out.shape
        grad = torch.zeros((B, C, N), dtype=torch.float32, device=grad_out.device)
        grad_out = grad_out.contiguous()
        _lib.call("mssvt_group_points_grad", _i(B), _i(C), _i(N), _i(npts), _i(nsample),
                  _lib.ptr(grad_out), _lib.ptr(idx), _lib.ptr(grad), _lib.stream())
        return grad, None


def grouping_operation(features, idx):
    """features (B, C, N) f32, idx (B, npoint, nsample) int32 -> (B, C, npoint, nsample)."""
    return _GroupPoints.apply(features, idx)
