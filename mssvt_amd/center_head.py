"""``CenterHead`` (inference) -- the dense head of the CenterPoint detector that mssvt.yaml configures.

Drop-in for the reference's pcdet/models/dense_heads/center_head.py: ``SeparateHead`` (:11-46) and ``CenterHead``
(:49-103, forward :350-381, ``generate_predicted_boxes`` :252-331) with the same constructor arguments, config keys
and **state-dict keys** (``shared_conv.{0,1}.*``, ``heads_list.{i}.{hm,center,center_z,dim,rot[,vel]}.{k}...``), so a
reference checkpoint loads by key.  Decoding follows centernet_utils.decode_bbox_from_heatmap (:154-216): top-K peaks of
the sigmoid heat map, sub-cell centre offsets, exp() sizes, atan2 heading, centre-range and score filters; then the
class-agnostic rotated NMS of model_nms_utils.class_agnostic_nms (:6-38) on the device (mssvt_amd/iou3d_nms_utils.py).
Training targets / losses (assign_targets, get_loss: :105-250) are outside this build's scope: the head raises in
training mode instead of silently returning nothing."""
import copy

import numpy as np
import torch
from torch import nn

from . import iou3d_nms_utils


def _get(cfg, key, default=None):
    return cfg.get(key, default) if hasattr(cfg, "get") else getattr(cfg, key, default)


class SeparateHead(nn.Module):
    def __init__(self, input_channels, sep_head_dict, init_bias=-2.19, use_bias=False):
        super().__init__()
        self.sep_head_dict = sep_head_dict
        for name, spec in sep_head_dict.items():
            layers = []
            for _ in range(int(spec["num_conv"]) - 1):
                layers.append(nn.Sequential(nn.Conv2d(input_channels, input_channels, 3, 1, 1, bias=use_bias),
                                            nn.BatchNorm2d(input_channels), nn.ReLU()))
            layers.append(nn.Conv2d(input_channels, int(spec["out_channels"]), 3, 1, 1, bias=True))
            fc = nn.Sequential(*layers)
            if "hm" in name:
                fc[-1].bias.data.fill_(init_bias)
            else:
                for m in fc.modules():
                    if isinstance(m, nn.Conv2d):
                        nn.init.kaiming_normal_(m.weight.data)
                        if m.bias is not None:
                            nn.init.constant_(m.bias, 0)
            setattr(self, name, fc)

    def forward(self, x):
        return {name: getattr(self, name)(x) for name in self.sep_head_dict}


def class_agnostic_nms(box_scores, box_preds, nms_config, score_thresh=None, idx=None, nms_fn=None):
    """ref model_nms_utils.class_agnostic_nms (:6-38): top NMS_PRE_MAXSIZE by score -> NMS -> first NMS_POST_MAXSIZE."""
    src_scores = box_scores
    scores_mask = None
    if score_thresh is not None:
        scores_mask = box_scores >= score_thresh
        box_scores, box_preds = box_scores[scores_mask], box_preds[scores_mask]
    thr = _get(nms_config, "NMS_THRESH")
    if isinstance(thr, (list, tuple)):
        thr = thr[idx] if idx is not None else thr[0]
    selected = torch.zeros(0, dtype=torch.long, device=box_scores.device)
    if box_scores.shape[0] > 0:
        top_scores, indices = torch.topk(box_scores, k=min(int(_get(nms_config, "NMS_PRE_MAXSIZE")), box_scores.shape[0]))
        fn = nms_fn or getattr(iou3d_nms_utils, _get(nms_config, "NMS_TYPE"))
        keep, _ = fn(box_preds[indices][:, 0:7], top_scores, thr)
        selected = indices[keep[:int(_get(nms_config, "NMS_POST_MAXSIZE"))]]
    if scores_mask is not None:
        selected = scores_mask.nonzero().view(-1)[selected]
    return selected, src_scores[selected]


def decode_bbox_from_heatmap(heatmap, rot_cos, rot_sin, center, center_z, dim, point_cloud_range, voxel_size,
                             feature_map_stride, vel=None, K=100, score_thresh=None, post_center_limit_range=None):
    """ref centernet_utils.decode_bbox_from_heatmap (:154-216); the two-stage top-K of ``_topk`` (:136-151) equals a
    top-K over all (class, cell) pairs."""
    B, ncls, H, W = heatmap.shape
    scores, flat = torch.topk(heatmap.reshape(B, -1), K)
    cls = (flat // (H * W)).int()
    cell = flat % (H * W)
    ys, xs = (cell // W).float(), (cell % W).float()

    def pick(t):  # (B, c, H, W) -> (B, K, c) at the peak cells
        c = t.shape[1]
        return t.permute(0, 2, 3, 1).reshape(B, H * W, c).gather(1, cell.unsqueeze(2).expand(B, K, c))

    ctr, cz, dm = pick(center), pick(center_z), pick(dim)
    angle = torch.atan2(pick(rot_sin), pick(rot_cos))
    x = (xs.unsqueeze(2) + ctr[:, :, 0:1]) * feature_map_stride * voxel_size[0] + point_cloud_range[0]
    y = (ys.unsqueeze(2) + ctr[:, :, 1:2]) * feature_map_stride * voxel_size[1] + point_cloud_range[1]
    parts = [x, y, cz, dm, angle]
    if vel is not None:
        parts.append(pick(vel))
    boxes = torch.cat(parts, dim=-1)
    mask = (boxes[..., :3] >= post_center_limit_range[:3]).all(2) & (boxes[..., :3] <= post_center_limit_range[3:]).all(2)
    if score_thresh is not None:
        mask &= scores > score_thresh
    return [dict(pred_boxes=boxes[b, mask[b]], pred_scores=scores[b, mask[b]], pred_labels=cls[b, mask[b]])
            for b in range(B)]


class CenterHead(nn.Module):
    def __init__(self, model_cfg, input_channels, num_class, class_names, grid_size, point_cloud_range, voxel_size,
                 predict_boxes_when_training=True):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_class = num_class
        self.grid_size, self.point_cloud_range, self.voxel_size = grid_size, point_cloud_range, voxel_size
        self.feature_map_stride = _get(_get(model_cfg, "TARGET_ASSIGNER_CONFIG"), "FEATURE_MAP_STRIDE", None)
        self.class_names = list(class_names)
        self.class_names_each_head, maps = [], []
        for names in _get(model_cfg, "CLASS_NAMES_EACH_HEAD"):
            cur = [x for x in names if x in self.class_names]
            self.class_names_each_head.append(cur)
            maps.append(torch.from_numpy(np.array([self.class_names.index(x) for x in cur], dtype=np.int64)))
        self._class_maps = maps  # moved to the features' device on use (the reference calls .cuda() in the constructor)
        assert sum(len(x) for x in self.class_names_each_head) == len(self.class_names)
        shared = int(_get(model_cfg, "SHARED_CONV_CHANNEL"))
        use_bias = bool(_get(model_cfg, "USE_BIAS_BEFORE_NORM", False))
        self.shared_conv = nn.Sequential(nn.Conv2d(input_channels, shared, 3, stride=1, padding=1, bias=use_bias),
                                         nn.BatchNorm2d(shared), nn.ReLU())
        self.separate_head_cfg = _get(model_cfg, "SEPARATE_HEAD_CFG")
        self.heads_list = nn.ModuleList()
        for cur in self.class_names_each_head:
            head_dict = copy.deepcopy({k: dict(v) for k, v in dict(_get(self.separate_head_cfg, "HEAD_DICT")).items()})
            head_dict["hm"] = dict(out_channels=len(cur), num_conv=int(_get(model_cfg, "NUM_HM_CONV")))
            self.heads_list.append(SeparateHead(shared, head_dict, init_bias=-2.19, use_bias=use_bias))
        self.predict_boxes_when_training = predict_boxes_when_training
        self.forward_ret_dict = {}
        self.nms_fn = None  # test hook: an alternative NMS with the signature of iou3d_nms_utils.nms_gpu

    def generate_predicted_boxes(self, batch_size, pred_dicts):
        post = _get(self.model_cfg, "POST_PROCESSING")
        nms_cfg = _get(post, "NMS_CONFIG")
        dev = pred_dicts[0]["hm"].device
        limit = torch.tensor(list(_get(post, "POST_CENTER_LIMIT_RANGE")), dtype=torch.float32, device=dev)
        ret = [dict(pred_boxes=[], pred_scores=[], pred_labels=[]) for _ in range(batch_size)]
        has_vel = "vel" in list(_get(self.separate_head_cfg, "HEAD_ORDER"))
        for idx, pd in enumerate(pred_dicts):
            finals = decode_bbox_from_heatmap(
                heatmap=pd["hm"].sigmoid(), rot_cos=pd["rot"][:, 0:1], rot_sin=pd["rot"][:, 1:2], center=pd["center"],
                center_z=pd["center_z"], dim=pd["dim"].exp(), vel=pd["vel"] if has_vel else None,
                point_cloud_range=self.point_cloud_range, voxel_size=self.voxel_size,
                feature_map_stride=self.feature_map_stride, K=int(_get(post, "MAX_OBJ_PER_SAMPLE")),
                score_thresh=_get(post, "SCORE_THRESH"), post_center_limit_range=limit)
            cmap = self._class_maps[idx].to(dev)
            thr = _get(nms_cfg, "NMS_THRESH")
            per_class = isinstance(thr, (list, tuple)) and len(thr) > 1
            for k, fd in enumerate(finals):
                if per_class:  # one threshold per class of this head (ref :283-305)
                    boxes, scores, labels = [], [], []
                    for i in range(len(thr)):
                        sel_c = fd["pred_labels"] == i
                        cb, cs = fd["pred_boxes"][sel_c], fd["pred_scores"][sel_c]
                        cl = cmap[fd["pred_labels"][sel_c].long()]
                        sel, ssc = class_agnostic_nms(cs, cb, nms_cfg, None, idx=i, nms_fn=self.nms_fn)
                        boxes.append(cb[sel]); scores.append(ssc); labels.append(cl[sel])
                    fd = dict(pred_boxes=torch.cat(boxes, 0), pred_scores=torch.cat(scores, 0),
                              pred_labels=torch.cat(labels, 0))
                else:
                    labels = cmap[fd["pred_labels"].long()]
                    sel, ssc = class_agnostic_nms(fd["pred_scores"], fd["pred_boxes"], nms_cfg, None, nms_fn=self.nms_fn)
                    fd = dict(pred_boxes=fd["pred_boxes"][sel], pred_scores=ssc, pred_labels=labels[sel])
                for key in ret[k]:
                    ret[k][key].append(fd[key])
        for k in range(batch_size):
            ret[k]["pred_boxes"] = torch.cat(ret[k]["pred_boxes"], dim=0)
            ret[k]["pred_scores"] = torch.cat(ret[k]["pred_scores"], dim=0)
            ret[k]["pred_labels"] = torch.cat(ret[k]["pred_labels"], dim=0) + 1
        return ret

    def forward(self, data_dict):
        if self.training:
            raise NotImplementedError("mssvt_amd CenterHead is inference-only: target assignment and the CenterNet losses "
                                      "(ref center_head.py:105-250) are outside this build's scope; call .eval()")
        x = self.shared_conv(data_dict["spatial_features_2d"])
        pred_dicts = [head(x) for head in self.heads_list]
        self.forward_ret_dict["pred_dicts"] = pred_dicts
        data_dict["final_box_dicts"] = self.generate_predicted_boxes(data_dict["batch_size"], pred_dicts)
        return data_dict
