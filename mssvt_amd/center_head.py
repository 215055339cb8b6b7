"""``CenterHead`` -- the dense head of the CenterPoint detector that mssvt.yaml configures.

Drop-in for the reference's pcdet/models/dense_heads/center_head.py: ``SeparateHead`` (:11-46) and ``CenterHead``
(:49-103, forward :350-381, ``generate_predicted_boxes`` :252-331) with the same constructor arguments, config keys
and **state-dict keys** (``shared_conv.{0,1}.*``, ``heads_list.{i}.{hm,center,center_z,dim,rot[,vel]}.{k}...``), so a
reference checkpoint loads by key.  Decoding follows centernet_utils.decode_bbox_from_heatmap (:154-216): top-K peaks of
the sigmoid heat map, sub-cell centre offsets, exp() sizes, atan2 heading, centre-range and score filters; then the
class-agnostic rotated NMS of model_nms_utils.class_agnostic_nms (:6-38) on the device (mssvt_amd/iou3d_nms_utils.py).
Training (``.train()``): ``assign_targets`` (:103-214: Gaussian heat maps with CornerNet's radius, sub-cell offsets, log sizes,
cos / sin headings, flat cell indices, masks) and ``get_loss`` (:220-250: CenterNet focal loss on the clamped sigmoid + masked
L1 on the gathered regression maps, ``code_weights`` / ``loc_weight``) -- pinned to a training step of the reference's own
module with its own loss classes (tests/golden/det_head_train.npz: targets, loss terms, every parameter gradient)."""
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


def gaussian_radius(height, width, min_overlap):
    """CornerNet's radius: the smallest of the three roots that keep a shifted box above `min_overlap` IoU
    (ref centernet_utils.py:9-35); vectorised over the objects."""
    s, p = height + width, height * width
    r1 = (s + (s * s - 4.0 * p * (1.0 - min_overlap) / (1.0 + min_overlap)).sqrt()) / 2.0
    r2 = (2.0 * s + (4.0 * s * s - 16.0 * (1.0 - min_overlap) * p).sqrt()) / 2.0
    b3 = -2.0 * min_overlap * s
    r3 = (b3 + (b3 * b3 - 16.0 * min_overlap * (min_overlap - 1.0) * p).sqrt()) / 2.0
    return torch.min(torch.min(r1, r2), r3)


def splat_gaussian(heatmap, cx, cy, radius):
    """max-merge a (2 r + 1)^2 Gaussian (sigma = (2 r + 1) / 6, evaluated in float64, entries below eps x peak dropped)
    centred on cell (cx, cy) into `heatmap` (H, W), clipped at the borders (ref centernet_utils.py:38-69)."""
    H, W = heatmap.shape
    ax = np.arange(-radius, radius + 1, dtype=np.float64)
    sigma = (2 * radius + 1) / 6.0
    g = np.exp(-(ax[None, :] ** 2 + ax[:, None] ** 2) / (2.0 * sigma * sigma))
    g[g < np.finfo(g.dtype).eps * g.max()] = 0
    left, right = min(cx, radius), min(W - cx, radius + 1)
    top, bottom = min(cy, radius), min(H - cy, radius + 1)
    if right + left <= 0 or bottom + top <= 0:
        return
    patch = torch.from_numpy(g[radius - top:radius + bottom, radius - left:radius + right]).to(heatmap.device).float()
    region = heatmap[cy - top:cy + bottom, cx - left:cx + right]
    if min(patch.shape) > 0 and min(region.shape) > 0:
        torch.max(region, patch, out=region)


def centernet_focal_loss(pred, gt):
    """ref loss_utils.py:264-299 (CornerNet's penalty-reduced focal loss): pred = clamped sigmoid, gt = Gaussian heat map."""
    pos = gt.eq(1).float()
    neg = gt.lt(1).float()
    pos_term = (torch.log(pred) * torch.pow(1 - pred, 2) * pos).sum()
    neg_term = (torch.log(1 - pred) * torch.pow(pred, 2) * torch.pow(1 - gt, 4) * neg).sum()
    num_pos = pos.sum()
    return -neg_term if num_pos == 0 else -(pos_term + neg_term) / num_pos


def centernet_reg_loss(maps, mask, ind, target):
    """ref loss_utils.py:314-386: L1 between the regression maps gathered at the objects' cells and the targets, per code
    dimension, over the masked objects of the whole batch / their number.  maps (B, D, H, W), ind / mask (B, M), target (B, M, D)."""
    B, D = maps.shape[0], maps.shape[1]
    flat = maps.permute(0, 2, 3, 1).reshape(B, -1, D)
    pred = flat.gather(1, ind.unsqueeze(2).expand(-1, -1, D))
    num = mask.float().sum()
    m = mask.unsqueeze(2).expand_as(target).float() * (~torch.isnan(target)).float()
    per_dim = torch.abs(pred * m - target * m).sum(dim=(0, 1))
    return per_dim / torch.clamp_min(num, 1.0)


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

    def assign_target_of_single_head(self, num_classes, gt_boxes, feature_map_size, feature_map_stride, num_max_objs=500,
                                     gaussian_overlap=0.1, min_radius=2):
        """gt_boxes (n, 8+) [x, y, z, dx, dy, dz, heading, ..., class in 1..num_classes] of ONE sample and head ->
        heat map (classes, H, W), target boxes (M, 8+), flat cell indices (M), mask (M)  (ref :103-158)."""
        W, H = int(feature_map_size[0]), int(feature_map_size[1])
        heatmap = gt_boxes.new_zeros(num_classes, H, W)
        ret_boxes = gt_boxes.new_zeros((num_max_objs, gt_boxes.shape[-1]))
        inds = gt_boxes.new_zeros(num_max_objs).long()
        mask = gt_boxes.new_zeros(num_max_objs).long()
        n = min(num_max_objs, gt_boxes.shape[0])
        if n == 0:
            return heatmap, ret_boxes, inds, mask
        g = gt_boxes[:n]
        cx = torch.clamp((g[:, 0] - self.point_cloud_range[0]) / self.voxel_size[0] / feature_map_stride, min=0, max=W - 0.5)
        cy = torch.clamp((g[:, 1] - self.point_cloud_range[1]) / self.voxel_size[1] / feature_map_stride, min=0, max=H - 0.5)
        centre = torch.stack((cx, cy), dim=-1)
        cell = centre.int()
        dx = g[:, 3] / self.voxel_size[0] / feature_map_stride
        dy = g[:, 4] / self.voxel_size[1] / feature_map_stride
        radius = torch.clamp_min(gaussian_radius(dx, dy, gaussian_overlap).int(), min=min_radius)
        ok = (dx > 0) & (dy > 0) & (cell[:, 0] >= 0) & (cell[:, 0] <= W) & (cell[:, 1] >= 0) & (cell[:, 1] <= H)
        cls = (g[:, -1] - 1).long()
        for k in torch.nonzero(ok).flatten().tolist():  # (the Gaussians overlap: max-merged in object order)
            splat_gaussian(heatmap[int(cls[k])], int(cell[k, 0]), int(cell[k, 1]), int(radius[k]))
        inds[:n][ok] = (cell[:, 1].long() * W + cell[:, 0].long())[ok]
        mask[:n] = ok.long()
        cols = [centre - cell.to(g.dtype), g[:, 2:3], g[:, 3:6].log(), torch.cos(g[:, 6:7]), torch.sin(g[:, 6:7])]
        if g.shape[1] > 8:
            cols.append(g[:, 7:-1])
        ret_boxes[:n][ok] = torch.cat(cols, dim=1)[ok]  # (skipped objects keep zero rows)
        return heatmap, ret_boxes, inds, mask

    def assign_targets(self, gt_boxes, feature_map_size=None, **kwargs):
        """gt_boxes (B, M, 8+) with the class id (1-based over class_names, 0 = padding) last; feature_map_size (H, W)
        -> per head: heatmaps (B, c, H, W), target_boxes (B, M', 8+), inds / masks (B, M')  (ref :160-214).  As in the
        reference a box reaches a head with its class re-numbered inside that head, and the re-numbering is written into the
        working copy of the boxes that the FOLLOWING heads read their class names from (ref :190-193)."""
        size_xy = list(feature_map_size)[::-1]
        tcfg = _get(self.model_cfg, "TARGET_ASSIGNER_CONFIG")
        work = gt_boxes.clone()
        names = ["bg"] + self.class_names
        out = dict(heatmaps=[], target_boxes=[], inds=[], masks=[], heatmap_masks=[])
        for head_names in self.class_names_each_head:
            per_sample = []
            for b in range(work.shape[0]):
                boxes = work[b]
                labels = boxes[:, -1].long().tolist()
                rows = [i for i, c in enumerate(labels) if names[c] in head_names]
                for i in rows:
                    boxes[i, -1] = head_names.index(names[labels[i]]) + 1
                sel = boxes[rows] if rows else boxes[:0]
                per_sample.append(self.assign_target_of_single_head(
                    num_classes=len(head_names), gt_boxes=sel.cpu(), feature_map_size=size_xy,
                    feature_map_stride=_get(tcfg, "FEATURE_MAP_STRIDE"), num_max_objs=_get(tcfg, "NUM_MAX_OBJS"),
                    gaussian_overlap=_get(tcfg, "GAUSSIAN_OVERLAP"), min_radius=_get(tcfg, "MIN_RADIUS")))
            for key, j in (("heatmaps", 0), ("target_boxes", 1), ("inds", 2), ("masks", 3)):
                out[key].append(torch.stack([t[j] for t in per_sample], dim=0).to(gt_boxes.device))
        return out

    def get_loss(self):
        """ref :220-250: per head the focal loss of the clamped sigmoid heat map + loc_weight x sum(code_weights x L1)."""
        weights = _get(_get(self.model_cfg, "LOSS_CONFIG"), "LOSS_WEIGHTS")
        order = list(_get(self.separate_head_cfg, "HEAD_ORDER"))
        loss, tb = 0, {}
        for idx, pd in enumerate(self.forward_ret_dict["pred_dicts"]):
            td = self.forward_ret_dict["target_dicts"]
            hm = torch.clamp(pd["hm"].sigmoid(), min=1e-4, max=1 - 1e-4)
            hm_loss = centernet_focal_loss(hm, td["heatmaps"][idx])
            reg = centernet_reg_loss(torch.cat([pd[name] for name in order], dim=1), td["masks"][idx], td["inds"][idx],
                                     td["target_boxes"][idx])
            loc_loss = (reg * reg.new_tensor(list(weights["code_weights"]))).sum() * weights["loc_weight"]
            loss = loss + hm_loss + loc_loss
            tb["hm_loss_head_%d" % idx] = hm_loss.item()
            tb["loc_loss_head_%d" % idx] = loc_loss.item()
        tb["rpn_loss"] = loss.item()
        return loss, tb

    def forward(self, data_dict):
        x = self.shared_conv(data_dict["spatial_features_2d"])
        pred_dicts = [head(x) for head in self.heads_list]
        if self.training:
            self.forward_ret_dict["target_dicts"] = self.assign_targets(
                data_dict["gt_boxes"], feature_map_size=data_dict["spatial_features_2d"].shape[2:])
        self.forward_ret_dict["pred_dicts"] = pred_dicts
        if not self.training or self.predict_boxes_when_training:
            with torch.no_grad():
                boxes = self.generate_predicted_boxes(data_dict["batch_size"], pred_dicts)
            if self.training:
                data_dict["pred_box_dicts"] = boxes  # (the reference re-orders them into `rois` for a second stage: not built)
            else:
                data_dict["final_box_dicts"] = boxes
        return data_dict
