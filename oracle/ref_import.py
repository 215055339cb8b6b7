"""Import the reference's own Python for the hot path on CPU (BUILD CONTAINER ONLY).

TEST INFRASTRUCTURE.  Used by oracle/gen_golden.py to pin the oracle against the
reference itself: the reference's ``MixedScaleAttention``, ``get_vox_query_table``,
``MixedScaleSparseTransformerBlock/CompressBlock.forward`` and
``MixedScaleSparseTransformer.forward`` (``pcdet/models/backbones_3d/mssvt_backbone.py``,
``pcdet/models/model_utils/mssvt_utils.py``) are executed unmodified, from
``/root/reference``, with

* namespace stubs for the ``pcdet`` packages (their ``__init__`` files pull in
  spconv / SharedArray / a generated ``version.py``),
* the two CUDA extension modules (``mssvt_ops_cuda``, ``pointnet2_batch_cuda``)
  replaced by the C oracle (oracle/mssvt_oracle.c) working in place on CPU tensors,
* ``torch.cuda.FloatTensor/IntTensor``, ``Tensor.cuda`` and ``device='cuda'``
  redirected to the CPU, and a 4-line ``timm`` ``DropPath`` stub.

Nothing here travels to the GPU box: /root/reference does not exist there, and
no reference source is copied -- only the numeric outputs (tests/golden/*.npz).
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

from . import cref

REFERENCE_ROOT = os.environ.get("MSSVT_REFERENCE_ROOT", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "pcdet"))


def _np(t):
    assert t.is_contiguous(), "reference wrappers require contiguous tensors"
    return t.numpy()


def _make_mssvt_ops_cuda():
    """CPU stand-in for pcdet/ops/mssvt/src/ms_api.cpp:7-14 (same names/args)."""
    m = types.ModuleType("pcdet.ops.mssvt.mssvt_ops_cuda")
    L = cref.lib()
    ip, fp = cref._ip, cref._fp

    def build_mapping_with_hash_wrapper(x_max, y_max, z_max, num_voxels, hash_size,
                                        v_indices, v_bs_cnt, table):
        L.orc_build_mapping_with_hash(x_max, y_max, z_max, num_voxels, hash_size,
                                      ip(_np(v_indices)), ip(_np(v_bs_cnt)), ip(_np(table)))
        return 1

    def window_with_hash_wrapper(x_wgs, y_wgs, z_wgs, x_ws, y_ws, z_ws, num_voxels,
                                 num_windows, hash_size, v_indices, w_indices, table, vcount):
        rc = L.orc_window_with_hash(x_wgs, y_wgs, z_wgs, x_ws, y_ws, z_ws, num_voxels,
                                    num_windows, hash_size, ip(_np(v_indices)),
                                    ip(_np(w_indices)), ip(_np(table)), ip(_np(vcount)))
        assert rc == 0
        return 1

    def gather_two_window_voxels_with_hash_wrapper(
            x_max, y_max, z_max, x_ws, y_ws, z_ws, max_num_odd, max_num_even, max_num_win1,
            max_num_win2, num_wins, hash_size, num_odd, num_even, num_win1, num_win2,
            ind_odd, ind_even, ind_win1, ind_win2, c_odd, c_even, c_win1, c_win2,
            q_odd, q_even, q_win1, q_win2, win_indices, table):
        L.orc_gather_two_window_voxels(
            x_max, y_max, z_max, x_ws, y_ws, z_ws, max_num_odd, max_num_even, max_num_win1,
            max_num_win2, num_wins, hash_size, num_odd, num_even, num_win1, num_win2,
            ip(_np(ind_odd)), ip(_np(ind_even)), ip(_np(ind_win1)), ip(_np(ind_win2)),
            ip(_np(c_odd)), ip(_np(c_even)), ip(_np(c_win1)), ip(_np(c_win2)),
            ip(_np(q_odd.contiguous())), ip(_np(q_even.contiguous())),
            ip(_np(q_win1.contiguous())), ip(_np(q_win2.contiguous())),
            ip(_np(win_indices)), ip(_np(table)))
        return 1

    def gather_one_window_voxels_with_hash_wrapper(
            x_max, y_max, z_max, x_ws, y_ws, z_ws, max_num_win1, num_wins, hash_size, num_win1,
            ind_win1, c_win1, q_win1, win_indices, table):
        L.orc_gather_one_window_voxels(
            x_max, y_max, z_max, x_ws, y_ws, z_ws, max_num_win1, num_wins, hash_size, num_win1,
            ip(_np(ind_win1)), ip(_np(c_win1)), ip(_np(q_win1.contiguous())),
            ip(_np(win_indices)), ip(_np(table)))
        return 1

    def group_features_wrapper(B, M, C, nsample, features, features_batch_cnt, idx,
                               idx_batch_cnt, out):
        L.orc_group_features(B, M, C, nsample, fp(_np(features)), ip(_np(features_batch_cnt)),
                             ip(_np(idx)), ip(_np(idx_batch_cnt)), fp(_np(out)))
        return 1

    def group_features_grad_wrapper(B, M, C, N, nsample, grad_out, idx, idx_batch_cnt,
                                    features_batch_cnt, grad_features):
        L.orc_group_features_grad(B, M, C, N, nsample, fp(_np(grad_out)), ip(_np(idx)),
                                  ip(_np(idx_batch_cnt)), ip(_np(features_batch_cnt)),
                                  fp(_np(grad_features)))
        return 1

    for f in (build_mapping_with_hash_wrapper, window_with_hash_wrapper,
              gather_two_window_voxels_with_hash_wrapper,
              gather_one_window_voxels_with_hash_wrapper, group_features_wrapper,
              group_features_grad_wrapper):
        setattr(m, f.__name__, f)
    return m


def _make_pointnet2_batch_cuda():
    """CPU stand-in for pointnet2_batch/src/pointnet2_api.cpp:10-24 (ops on the path)."""
    m = types.ModuleType("pcdet.ops.pointnet2.pointnet2_batch.pointnet2_batch_cuda")
    L = cref.lib()
    ip, fp = cref._ip, cref._fp

    def farthest_point_sampling_wrapper(B, N, npoint, xyz, temp, out):
        L.orc_farthest_point_sampling(B, N, npoint, fp(_np(xyz)), fp(_np(temp)), ip(_np(out)))
        return 1

    def gather_points_wrapper(B, C, N, npoint, features, idx, out):
        L.orc_gather_points(B, C, N, npoint, fp(_np(features)), ip(_np(idx)), fp(_np(out)))
        return 1

    def gather_points_grad_wrapper(B, C, N, npoint, grad_out, idx, grad_points):
        L.orc_gather_points_grad(B, C, N, npoint, fp(_np(grad_out)), ip(_np(idx)),
                                 fp(_np(grad_points)))
        return 1

    def three_nn_wrapper(B, N, m_, unknown, known, dist2, idx):
        L.orc_three_nn(B, N, m_, fp(_np(unknown)), fp(_np(known)), fp(_np(dist2)), ip(_np(idx)))
        return 1

    def group_points_wrapper(B, C, N, npts, ns, features, idx, out):
        L.orc_group_points(B, C, N, npts, ns, fp(_np(features)), ip(_np(idx)), fp(_np(out)))
        return 1

    def group_points_grad_wrapper(B, C, N, npts, ns, grad_out, idx, grad_points):
        L.orc_group_points_grad(B, C, N, npts, ns, fp(_np(grad_out)), ip(_np(idx)),
                                fp(_np(grad_points)))
        return 1

    for f in (farthest_point_sampling_wrapper, gather_points_wrapper, gather_points_grad_wrapper,
              three_nn_wrapper, group_points_wrapper, group_points_grad_wrapper):
        setattr(m, f.__name__, f)
    return m


_loaded = None


def load():
    """Return (mssvt_backbone, mssvt_utils, mssvt_ops, pointnet2_utils) reference modules."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError("reference tree not present (expected in the build container only)")

    # --- CPU redirections of the hard-coded CUDA bits ----------------------
    torch.cuda.FloatTensor = torch.FloatTensor
    torch.cuda.IntTensor = torch.IntTensor
    torch.Tensor.cuda = lambda self, *a, **k: self
    _orig_tensor = torch.tensor

    def _tensor(*a, **k):
        if str(k.get("device", "")).startswith("cuda"):
            k.pop("device")
        return _orig_tensor(*a, **k)

    torch.tensor = _tensor

    # --- timm stub ---------------------------------------------------------
    if "timm" not in sys.modules:
        timm = types.ModuleType("timm")
        timm_models = types.ModuleType("timm.models")
        timm_layers = types.ModuleType("timm.models.layers")

        class DropPath(torch.nn.Module):  # identity in eval; never constructed with p>0 in tests
            def __init__(self, p=0.0):
                super().__init__()
                self.p = p

            def forward(self, x):
                assert not self.training or self.p == 0.0
                return x

        timm_layers.DropPath = DropPath
        timm.models = timm_models
        timm_models.layers = timm_layers
        sys.modules.update({"timm": timm, "timm.models": timm_models,
                            "timm.models.layers": timm_layers})

    # --- namespace stubs (bypass every package __init__) --------------------
    def ns(name, rel):
        mod = types.ModuleType(name)
        mod.__path__ = [os.path.join(REFERENCE_ROOT, rel)]
        sys.modules[name] = mod
        return mod

    ns("pcdet", "pcdet")
    ns("pcdet.models", "pcdet/models")
    ns("pcdet.models.model_utils", "pcdet/models/model_utils")
    ns("pcdet.models.backbones_3d", "pcdet/models/backbones_3d")
    ns("pcdet.ops", "pcdet/ops")
    ns("pcdet.ops.mssvt", "pcdet/ops/mssvt")
    ns("pcdet.ops.pointnet2", "pcdet/ops/pointnet2")
    ns("pcdet.ops.pointnet2.pointnet2_batch", "pcdet/ops/pointnet2/pointnet2_batch")

    ext1 = _make_mssvt_ops_cuda()
    ext2 = _make_pointnet2_batch_cuda()
    sys.modules[ext1.__name__] = ext1
    sys.modules[ext2.__name__] = ext2
    sys.modules["pcdet.ops.mssvt"].mssvt_ops_cuda = ext1
    sys.modules["pcdet.ops.pointnet2.pointnet2_batch"].pointnet2_batch_cuda = ext2

    mssvt_ops = importlib.import_module("pcdet.ops.mssvt.mssvt_ops")
    pointnet2_utils = importlib.import_module("pcdet.ops.pointnet2.pointnet2_batch.pointnet2_utils")
    mssvt_utils = importlib.import_module("pcdet.models.model_utils.mssvt_utils")
    mssvt_backbone = importlib.import_module("pcdet.models.backbones_3d.mssvt_backbone")
    _loaded = (mssvt_backbone, mssvt_utils, mssvt_ops, pointnet2_utils)
    return _loaded


class AttrDict(dict):
    """Minimal EasyDict stand-in (``easydict`` is absent; the reference only uses
    attribute access and ``.get`` on its config nodes)."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        return v

    @staticmethod
    def wrap(o):
        if isinstance(o, dict):
            return AttrDict({k: AttrDict.wrap(v) for k, v in o.items()})
        if isinstance(o, (list, tuple)):
            return [AttrDict.wrap(v) for v in o]
        return o
