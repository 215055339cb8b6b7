"""``DynamicVFE`` for the MI355X path: points -> (voxel_features, voxel_coords) without torch_scatter.

Interface-compatible with the reference's ``pcdet/models/backbones_3d/vfe/dynamic_vfe.py:12-131`` (same
constructor, ``forward(batch_dict)`` keys, ``get_output_feature_dim`` and state-dict keys ``pfn.{i}.{0,1}.*``).
The index part (``torch.unique`` of the merged coordinate) runs in the bitmap voxelizer
(``csrc/voxelize.hip``), ``scatter_mean`` / ``scatter_max`` of the un-vendored torch_scatter package in
``csrc/vfe.hip``; the tiny per-point PFN layers (Linear + BatchNorm1d + ReLU) stay torch modules.
Inference (eval mode / no grad): the HIP reductions; points outside the grid are kept in the arrays (their voxel id
is -1) instead of being filtered out, so nothing is compacted or synchronised.  Training (``.train()`` with autograd
on): the reference's semantics in differentiable torch operations -- out-of-grid points are DROPPED before the PFN, so
BatchNorm1d sees exactly the reference's rows (dynamic_vfe.py:85-91), the per-voxel mean is an ``index_add``, the
per-voxel max a ``scatter_reduce('amax')`` (gradient to the maximal rows; pinned to a training step of the reference's
module by tests/golden/dynamic_vfe_train_*.npz: output, every parameter gradient, the BatchNorm buffers).
"""
import ctypes
import os

import torch
from torch import nn

from . import _lib, voxelize

_i = ctypes.c_int
FUSED_PFN = os.environ.get("MSSVT_FUSED_PFN", "1") != "0"  # csrc/pfn_fused.hip for the default DynamicVFE configuration
# ... over points grouped by voxel (csrc/pfn_sorted.hip: no atomics on feature rows, no fills, x2 never stored); "0": the
# atomic reductions of round 5 (kept: the comparator of tests/test_vfe_gpu.py)
SORTED_PFN = os.environ.get("MSSVT_PFN_SORTED", "1") != "0"


def voxel_mean_xyz(points, point_voxel, num_voxels):
    """scatter_mean(points[:, 1:4], point_voxel) -> (N,3) f32, points per voxel (N,) int32."""
    P, stride = points.shape
    dev = points.device
    mean = torch.empty((max(num_voxels, 1), 3), dtype=torch.float32, device=dev)
    cnt = torch.empty(max(num_voxels, 1), dtype=torch.int32, device=dev)
    scratch = torch.empty(max(num_voxels, 1) * 3, dtype=torch.int64, device=dev)
    _lib.call("mssvt_voxel_mean_xyz", _lib.ptr(points), _i(stride), ctypes.c_longlong(P), _lib.ptr(point_voxel),
              _i(num_voxels), _lib.ptr(mean), _lib.ptr(cnt), _lib.ptr(scratch), _lib.stream())
    return mean[:num_voxels], cnt[:num_voxels]


def voxel_max(features, point_voxel, num_voxels):
    """scatter_max(features, point_voxel)[0] -> (N,F) f32."""
    features = features.contiguous()
    P, F = features.shape
    out = torch.empty((max(num_voxels, 1), F), dtype=torch.float32, device=features.device)
    _lib.call("mssvt_voxel_max", _lib.ptr(features), _i(F), ctypes.c_longlong(P), _lib.ptr(point_voxel),
              _i(num_voxels), _lib.ptr(out), _lib.stream())
    return out[:num_voxels]


class DynamicVFE(nn.Module):
    def __init__(self, model_cfg, num_point_features, voxel_size, grid_size, point_cloud_range, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        get = model_cfg.get if hasattr(model_cfg, 'get') else (lambda k, d=None: getattr(model_cfg, k, d))
        self.num_point_features_in = num_point_features
        self.grid_size_l = [int(v) for v in grid_size]
        self.voxel_size_l = [float(v) for v in voxel_size]
        self.point_cloud_range_l = [float(v) for v in point_cloud_range]
        # ref :29-35: voxel centre = coord * voxel_size + (voxel_size / 2 + range_min)
        self.register_buffer('voxel_size_t', torch.tensor(self.voxel_size_l, dtype=torch.float32).view(1, 3),
                             persistent=False)
        self.register_buffer('xyz_offset', torch.tensor(
            [self.voxel_size_l[k] / 2 + self.point_cloud_range_l[k] for k in range(3)], dtype=torch.float32).view(1, 3),
            persistent=False)
        self.with_cluster_center = get('WITH_CLUSTER_CENTER', True)
        self.with_voxel_center = get('WITH_VOXEL_CENTER', True)
        self.with_distance = get('WITH_DISTANCE', False)
        in_channels = num_point_features + (3 if self.with_cluster_center else 0) + \
            (3 if self.with_voxel_center else 0) + (1 if self.with_distance else 0)
        self.in_channels = in_channels
        filters = list(get('NUM_FILTERS', [64, 128]))
        self.num_point_features = filters[-1]
        self.pfn = nn.ModuleList([])
        in_c = in_channels
        for out_c in filters:
            self.pfn.append(nn.Sequential(nn.Linear(in_c, out_c), nn.BatchNorm1d(out_c), nn.ReLU(inplace=True)))
            in_c = out_c * 2

    def get_output_feature_dim(self):
        return self.num_point_features

    def forward(self, batch_dict, **kwargs):
        if self.training:
            # BatchNorm1d on batch statistics must see exactly the rows the reference feeds it, and the reductions must
            # carry gradients: the differentiable branch (also under no_grad in train mode: the running statistics move)
            return self._forward_train(batch_dict)
        with torch.no_grad():
            return self._forward_eval(batch_dict)

    def _forward_train(self, batch_dict):
        """ref dynamic_vfe.py:71-131 with torch_scatter's two reductions as differentiable torch operations."""
        pts = batch_dict['points']
        dev = pts.device
        lo = torch.tensor(self.point_cloud_range_l[:3], dtype=pts.dtype, device=dev)
        grid = torch.tensor(self.grid_size_l, dtype=torch.int32, device=dev)
        cell = torch.floor((pts[:, 1:4] - lo) / self.voxel_size_t.to(pts.dtype)).int()  # ref :85
        keep = ((cell >= 0) & (cell < grid)).all(dim=1)
        pts, cell = pts[keep], cell[keep]
        gx, gy, gz = self.grid_size_l
        key = ((pts[:, 0].long() * gx + cell[:, 0].long()) * gy + cell[:, 1].long()) * gz + cell[:, 2].long()  # ref :89-92
        keys, inv = torch.unique(key, return_inverse=True)  # sorted: (b, x, y, z) order, as the voxelizer's
        N = keys.shape[0]

        def per_voxel_mean(v):
            tot = torch.zeros((N, v.shape[1]), dtype=v.dtype, device=dev).index_add(0, inv, v)
            cnt = torch.zeros(N, dtype=v.dtype, device=dev).index_add(0, inv, torch.ones_like(inv, dtype=v.dtype))
            return tot / cnt.clamp(min=1).unsqueeze(1)

        def per_voxel_max(v):
            idx = inv.unsqueeze(1).expand_as(v)
            return torch.zeros((N, v.shape[1]), dtype=v.dtype, device=dev).scatter_reduce(0, idx, v, "amax", include_self=False)

        xyz = pts[:, 1:4]
        cols = [pts[:, 1:self.num_point_features_in + 1]]
        if self.with_cluster_center:
            cols.append(xyz - per_voxel_mean(xyz)[inv])
        if self.with_voxel_center:
            cols.append(xyz - (cell.to(pts.dtype) * self.voxel_size_t + self.xyz_offset))
        if self.with_distance:
            cols.append(torch.norm(xyz, p=2, dim=1, keepdim=True))
        x = torch.cat(cols, dim=-1)
        for i, blk in enumerate(self.pfn):
            x = blk(x)
            if i < len(self.pfn) - 1:
                x = torch.cat((x, per_voxel_max(x)[inv]), dim=-1)
        b = keys // (gx * gy * gz)
        rem = keys % (gx * gy * gz)
        coords = torch.stack((b, rem % gz, (rem // gz) % gy, rem // (gy * gz)), dim=1).int()  # [b, z, y, x] (ref :122-126)
        batch_dict['voxel_features'] = per_voxel_max(x).contiguous()
        batch_dict['voxel_coords'] = coords.contiguous()
        return batch_dict

    def _fused_pfn_ok(self, points):
        """The two PFN layers as two HIP launches (csrc/pfn_fused.hip): the default configuration only."""
        if not FUSED_PFN or self.num_point_features_in != 5 or not self.with_cluster_center or not self.with_voxel_center or \
                self.with_distance or len(self.pfn) != 2 or points.dtype != torch.float32 or points.shape[1] < 6:
            return False
        shapes = [(blk[0].in_features, blk[0].out_features) for blk in self.pfn]
        return shapes == [(11, 64), (128, 128)] and all(
            blk[0].bias is not None and blk[1].affine and blk[1].track_running_stats and blk[0].weight.dtype == torch.float32
            for blk in self.pfn)

    def _forward_eval(self, batch_dict):
        points = batch_dict['points'].contiguous()  # (P, 1 + F) rows [b, x, y, z, intensity, ...]
        batch_size = batch_dict['batch_size']
        voxel_coords, pv = voxelize.voxelize(points, self.point_cloud_range_l, self.voxel_size_l, self.grid_size_l,
                                             batch_size)
        N = voxel_coords.shape[0]
        if N and self._fused_pfn_ok(points) and SORTED_PFN:
            P, dev = points.shape[0], points.device
            lib = _lib.lib()
            lib.mssvt_pfn_sorted_workspace_ints.restype = ctypes.c_longlong
            ws = torch.empty(int(lib.mssvt_pfn_sorted_workspace_ints(ctypes.c_longlong(P), _i(N))), dtype=torch.int32, device=dev)
            x1 = torch.empty((P, 64), dtype=torch.float32, device=dev)
            m1 = torch.empty((N, 64), dtype=torch.float32, device=dev)
            out = torch.empty((N, 128), dtype=torch.float32, device=dev)
            (l1, n1), (l2, n2) = (self.pfn[0][0], self.pfn[0][1]), (self.pfn[1][0], self.pfn[1][1])
            f3 = lambda xs: (ctypes.c_float * 3)(*[float(v) for v in xs])  # noqa: E731
            vc = voxel_coords.contiguous()
            _lib.call("mssvt_pfn_sorted_64_128", _lib.ptr(points), _i(points.shape[1]), ctypes.c_longlong(P), _lib.ptr(pv), _i(N),
                      _lib.ptr(vc), f3(self.voxel_size_l),
                      f3([self.voxel_size_l[k] / 2 + self.point_cloud_range_l[k] for k in range(3)]),
                      _lib.ptr(l1.weight), _lib.ptr(l1.bias), _lib.ptr(n1.weight), _lib.ptr(n1.bias), _lib.ptr(n1.running_mean),
                      _lib.ptr(n1.running_var), ctypes.c_float(n1.eps), _lib.ptr(l2.weight), _lib.ptr(l2.bias), _lib.ptr(n2.weight),
                      _lib.ptr(n2.bias), _lib.ptr(n2.running_mean), _lib.ptr(n2.running_var), ctypes.c_float(n2.eps),
                      _lib.ptr(ws), _lib.ptr(x1), _lib.ptr(m1), _lib.ptr(out), _lib.stream())
            batch_dict['voxel_features'] = out
            batch_dict['voxel_coords'] = vc
            return batch_dict
        if N and self._fused_pfn_ok(points):
            xyz_mean, _ = voxel_mean_xyz(points, pv, N)
            P, dev = points.shape[0], points.device
            x1 = torch.empty((P, 64), dtype=torch.float32, device=dev)
            m1 = torch.empty((N, 64), dtype=torch.float32, device=dev)
            x2 = torch.empty((P, 128), dtype=torch.float32, device=dev)
            out = torch.empty((N, 128), dtype=torch.float32, device=dev)
            (l1, n1), (l2, n2) = (self.pfn[0][0], self.pfn[0][1]), (self.pfn[1][0], self.pfn[1][1])
            f3 = lambda xs: (ctypes.c_float * 3)(*[float(v) for v in xs])  # noqa: E731
            vc = voxel_coords.contiguous()
            _lib.call("mssvt_pfn_fused_64_128", _lib.ptr(points), _i(points.shape[1]), ctypes.c_longlong(P), _lib.ptr(pv), _i(N),
                      _lib.ptr(xyz_mean), _lib.ptr(vc), f3(self.voxel_size_l),
                      f3([self.voxel_size_l[k] / 2 + self.point_cloud_range_l[k] for k in range(3)]),
                      _lib.ptr(l1.weight), _lib.ptr(l1.bias), _lib.ptr(n1.weight), _lib.ptr(n1.bias), _lib.ptr(n1.running_mean),
                      _lib.ptr(n1.running_var), ctypes.c_float(n1.eps), _lib.ptr(l2.weight), _lib.ptr(l2.bias), _lib.ptr(n2.weight),
                      _lib.ptr(n2.bias), _lib.ptr(n2.running_mean), _lib.ptr(n2.running_var), ctypes.c_float(n2.eps),
                      _lib.ptr(x1), _lib.ptr(m1), _lib.ptr(x2), _lib.ptr(out), _lib.stream())
            batch_dict['voxel_features'] = out
            batch_dict['voxel_coords'] = vc
            return batch_dict
        gather = pv.clamp(min=0).long()  # points outside the grid read voxel 0; their rows are never reduced
        xyz = points[:, 1:4]
        feats = [points[:, 1:self.num_point_features_in + 1]]
        if self.with_cluster_center:
            xyz_mean, _ = voxel_mean_xyz(points, pv, N)
            feats.append(xyz - xyz_mean[gather] if N else torch.zeros_like(xyz))
        if self.with_voxel_center:
            pc = voxel_coords[gather][:, [3, 2, 1]].to(torch.float32) if N else torch.zeros_like(xyz)
            feats.append(xyz - (pc * self.voxel_size_t + self.xyz_offset))
        if self.with_distance:
            feats.append(torch.norm(xyz, p=2, dim=1, keepdim=True))  # (the reference's dim=2 cannot run)
        x = torch.cat(feats, dim=-1)
        for i, blk in enumerate(self.pfn):
            x = blk(x)
            if i < len(self.pfn) - 1:
                fea_v = voxel_max(x, pv, N)
                x = torch.cat((x, fea_v[gather] if N else torch.zeros_like(x)), dim=-1)
        batch_dict['voxel_features'] = voxel_max(x, pv, N).contiguous()
        batch_dict['voxel_coords'] = voxel_coords.contiguous()
        return batch_dict
