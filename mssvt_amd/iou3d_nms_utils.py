"""Rotated BEV NMS on MI355X -- the entry point the reference's ``model_nms_utils.class_agnostic_nms`` resolves by
name (``getattr(iou3d_nms_utils, nms_config.NMS_TYPE)``, ref: pcdet/models/model_utils/model_nms_utils.py:27-30;
pcdet/ops/iou3d_nms/iou3d_nms_utils.py:83-98).  Same signature and return value; the suppression matrix AND the greedy
walk run on the device (csrc/nms_bev.hip), one host sync for the number of boxes kept."""
import ctypes

import torch

from . import _lib

_i = ctypes.c_int


def nms_gpu(boxes, scores, thresh, pre_maxsize=None, **kwargs):
    """boxes (N, 7) [x, y, z, dx, dy, dz, heading], scores (N) -> (indices of the kept boxes, best first; None)."""
    assert boxes.shape[1] == 7
    order = scores.sort(0, descending=True)[1]
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    n = int(order.shape[0])
    if n == 0:
        return order, None
    b = boxes[order].float().contiguous()
    ws = torch.empty(int(_lib.lib().mssvt_nms_workspace_bytes(_i(n))) // 8 + 1, dtype=torch.int64, device=b.device)
    keep = torch.empty(n, dtype=torch.int32, device=b.device)
    cnt = torch.empty(1, dtype=torch.int32, device=b.device)
    _lib.call("mssvt_nms_bev", _i(n), _lib.ptr(b), ctypes.c_float(float(thresh)), _lib.ptr(ws), _lib.ptr(keep),
              _lib.ptr(cnt), _lib.stream())
    return order[keep[:int(cnt.item())].long()].contiguous(), None
