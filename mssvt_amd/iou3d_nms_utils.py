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
    if n > NMS_MAX_BOXES:
        # beyond one launch's capacity (the reference takes any N): score-ordered chunks, the boxes kept so far put in
        # front of the next chunk -- they outrank it and do not suppress each other, so greedy NMS keeps them and
        # applies them to the chunk: the same result as one pass
        kept = order[:0]
        for lo in range(0, n, NMS_MAX_BOXES // 2):
            chunk = order[lo:lo + NMS_MAX_BOXES // 2]
            if kept.numel() + chunk.numel() > NMS_MAX_BOXES:
                raise _lib.MssvtHipError("nms_gpu: more than %d boxes survive" % (NMS_MAX_BOXES // 2))
            cand = torch.cat([kept, chunk])
            kept = cand[_nms_sorted(boxes[cand].float().contiguous(), thresh)]
        return kept.contiguous(), None
    b = boxes[order].float().contiguous()
    return order[_nms_sorted(b, thresh)].contiguous(), None


NMS_MAX_BOXES = 16384  # csrc/nms_bev.hip: one launch


def _nms_sorted(b, thresh):
    """Indices (into `b`, best first) that greedy rotated NMS keeps among boxes already sorted by descending score."""
    n = int(b.shape[0])
    ws = torch.empty(int(_lib.lib().mssvt_nms_workspace_bytes(_i(n))) // 8 + 1, dtype=torch.int64, device=b.device)
    keep = torch.empty(n, dtype=torch.int32, device=b.device)
    cnt = torch.empty(1, dtype=torch.int32, device=b.device)
    _lib.call("mssvt_nms_bev", _i(n), _lib.ptr(b), ctypes.c_float(float(thresh)), _lib.ptr(ws), _lib.ptr(keep),
              _lib.ptr(cnt), _lib.stream())
    return keep[:int(cnt.item())].long()
