"""Golden vectors of the detector shell behind the backbone (BUILD CONTAINER ONLY).

Runs the REFERENCE's own ``BaseBEVBackbone`` (pcdet/models/backbones_2d/base_bev_backbone.py) and ``CenterHead``
(pcdet/models/dense_heads/center_head.py, incl. its centernet_utils decode and model_nms_utils.class_agnostic_nms)
on the CPU, imported unmodified from /root/reference with

* namespace stubs for the packages, a stub ``numba`` (centernet_utils imports it for target assignment only), empty
  loss classes for ``pcdet.utils.loss_utils`` (training only), ``Tensor.cuda`` -> CPU (oracle/ref_import.py);
* the CUDA extension ``iou3d_nms_cuda`` replaced by the numpy oracle (oracle/nms_ref.py), like the mssvt ops;

and commits inputs, weights, key lists and outputs as tests/golden/det_*.npz (data only).

    python -m oracle.gen_golden_det
"""
import importlib
import json
import os
import sys
import types

import numpy as np
import torch

from . import nms_ref, ref_import

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

BEV2D = dict(LAYER_NUMS=[2, 2], LAYER_STRIDES=[1, 2], NUM_FILTERS=[32, 64], UPSAMPLE_STRIDES=[1, 2],
             NUM_UPSAMPLE_FILTERS=[32, 32])
HEAD = dict(CLASS_AGNOSTIC=False, CLASS_NAMES_EACH_HEAD=[["Vehicle", "Pedestrian", "Cyclist"]], SHARED_CONV_CHANNEL=32,
            USE_BIAS_BEFORE_NORM=True, NUM_HM_CONV=2,
            SEPARATE_HEAD_CFG=dict(HEAD_ORDER=["center", "center_z", "dim", "rot"],
                                   HEAD_DICT=dict(center=dict(out_channels=2, num_conv=2),
                                                  center_z=dict(out_channels=1, num_conv=2),
                                                  dim=dict(out_channels=3, num_conv=2),
                                                  rot=dict(out_channels=2, num_conv=2))),
            TARGET_ASSIGNER_CONFIG=dict(FEATURE_MAP_STRIDE=1, NUM_MAX_OBJS=500, GAUSSIAN_OVERLAP=0.1, MIN_RADIUS=2),
            POST_PROCESSING=dict(SCORE_THRESH=0.1, POST_CENTER_LIMIT_RANGE=[-20.0, -20.0, -2.0, 20.0, 20.0, 4.0],
                                 MAX_OBJ_PER_SAMPLE=200,
                                 NMS_CONFIG=dict(NMS_TYPE="nms_gpu", NMS_THRESH=0.5, NMS_PRE_MAXSIZE=1000,
                                                 NMS_POST_MAXSIZE=80)))
CLASSES = ["Vehicle", "Pedestrian", "Cyclist"]
PCR = [-19.2, -19.2, -2.0, 19.2, 19.2, 4.0]
VOXEL = [0.6, 0.6, 6.0]
GRID = [64, 64, 1]


def load_reference():
    ref_import.load()  # namespace stubs + CPU redirections

    def ns(name, rel):
        mod = types.ModuleType(name)
        mod.__path__ = [os.path.join(ref_import.REFERENCE_ROOT, rel)]
        sys.modules[name] = mod
        return mod

    ns("pcdet.models.backbones_2d", "pcdet/models/backbones_2d")
    ns("pcdet.models.dense_heads", "pcdet/models/dense_heads")
    ns("pcdet.utils", "pcdet/utils")
    ns("pcdet.ops.iou3d_nms", "pcdet/ops/iou3d_nms")
    sys.modules.setdefault("numba", types.ModuleType("numba"))
    sys.modules["numba"].jit = lambda *a, **k: (lambda f: f)
    loss = types.ModuleType("pcdet.utils.loss_utils")

    class _NoLoss(torch.nn.Module):
        pass

    loss.FocalLossCenterNet = loss.RegLossCenterNet = _NoLoss
    sys.modules["pcdet.utils.loss_utils"] = loss
    sys.modules["pcdet.utils"].loss_utils = loss
    cu = types.ModuleType("pcdet.utils.common_utils")
    cu.check_numpy_to_torch = lambda x: (torch.from_numpy(x).float(), True) if isinstance(x, np.ndarray) else (x, False)
    sys.modules["pcdet.utils.common_utils"] = cu
    sys.modules["pcdet.utils"].common_utils = cu
    ext = types.ModuleType("pcdet.ops.iou3d_nms.iou3d_nms_cuda")

    def nms_gpu(boxes, keep, thresh):  # iou3d_nms.cpp:90-135 on the numpy oracle; boxes arrive sorted
        n = boxes.shape[0]
        kept = nms_ref.nms(boxes.numpy(), -np.arange(n, dtype=np.float64), thresh)
        keep[:len(kept)] = torch.from_numpy(np.asarray(kept, dtype=np.int64))
        return len(kept)

    ext.nms_gpu = nms_gpu
    sys.modules["pcdet.ops.iou3d_nms.iou3d_nms_cuda"] = ext
    sys.modules["pcdet.ops.iou3d_nms"].iou3d_nms_cuda = ext
    bev = importlib.import_module("pcdet.models.backbones_2d.base_bev_backbone")
    head = importlib.import_module("pcdet.models.dense_heads.center_head")
    return bev, head


def load_reference_losses():
    """The reference's own CenterNet losses (pcdet/utils/loss_utils.py:264-386) for the training golden: the module is
    imported unmodified; its one import that needs compiled extensions (box_utils, unused by these two losses) is an empty
    stand-in module."""
    box = types.ModuleType("pcdet.utils.box_utils")
    sys.modules["pcdet.utils.box_utils"] = box
    sys.modules["pcdet.utils"].box_utils = box
    sys.modules.pop("pcdet.utils.loss_utils", None)
    real = importlib.import_module("pcdet.utils.loss_utils")
    sys.modules["pcdet.utils"].loss_utils = real
    return real


def gen_head_training_step(head_mod):
    """One TRAINING step of the reference's CenterHead (ref center_head.py:103-250, 350-378): target assignment
    (Gaussian heat maps, regression targets, indices, masks), the CenterNet focal + L1 losses, every parameter gradient."""
    real_losses = load_reference_losses()
    head_mod.loss_utils = real_losses
    A = ref_import.AttrDict.wrap
    cfg = dict(HEAD)
    cfg["LOSS_CONFIG"] = dict(LOSS_WEIGHTS=dict(cls_weight=1.0, loc_weight=2.0, code_weights=[1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.2, 0.2]))
    torch.manual_seed(21)
    head = head_mod.CenterHead(A(cfg), 64, len(CLASSES), CLASSES, np.array(GRID), np.array(PCR), VOXEL,
                               predict_boxes_when_training=False).train()
    B, M = 2, 14
    g = torch.Generator().manual_seed(5)
    gt = torch.zeros(B, M, 8)
    for b in range(B):
        n = M - 3 * b  # the rest stays zero padding (class 0 = background: skipped)
        gt[b, :n, 0:2] = (torch.rand(n, 2, generator=g) - 0.5) * 36.0
        gt[b, :n, 2] = torch.rand(n, generator=g) * 2.0 - 1.0
        gt[b, :n, 3:6] = torch.rand(n, 3, generator=g) * torch.tensor([4.0, 2.0, 1.5]) + torch.tensor([0.5, 0.4, 0.8])
        gt[b, :n, 6] = (torch.rand(n, generator=g) - 0.5) * 6.28
        gt[b, :n, 7] = torch.randint(1, 4, (n,), generator=g).float()
    gt[0, 1, 0:2] = gt[0, 0, 0:2] + 0.3  # two objects whose Gaussians overlap / share a cell neighbourhood
    gt[1, 2, 0] = 19.1                   # an object at the border of the range: the Gaussian is clipped
    x = torch.randn(B, 64, GRID[1], GRID[0], generator=g)
    sd0 = {k: v.clone() for k, v in head.state_dict().items()}
    head(dict(spatial_features_2d=x, batch_size=B, gt_boxes=gt.clone()))
    loss, tb = head.get_loss()
    loss.backward()
    td = head.forward_ret_dict["target_dicts"]
    d = dict(spatial_features_2d=x.numpy(), gt_boxes=gt.numpy(), batch_size=B, loss=float(loss.item()),
             tb_json=json.dumps({k: float(v) for k, v in tb.items()}),
             cfg_json=json.dumps(dict(HEAD=cfg, CLASSES=CLASSES, PCR=PCR, VOXEL=VOXEL, GRID=GRID, input_channels=64)))
    for i in range(len(td["heatmaps"])):
        d["target%d.heatmaps" % i] = td["heatmaps"][i].numpy()
        d["target%d.target_boxes" % i] = td["target_boxes"][i].numpy()
        d["target%d.inds" % i] = td["inds"][i].numpy()
        d["target%d.masks" % i] = td["masks"][i].numpy()
    d.update({"head." + k: v.numpy() for k, v in sd0.items()})
    d.update({"grad." + k: v.grad.numpy() for k, v in head.named_parameters() if v.grad is not None})
    np.savez_compressed(os.path.join(OUT, "det_head_train.npz"), **d)
    print("det_head_train: loss", float(loss.item()), tb, "objects", int(td["masks"][0].sum()))


def main():
    os.makedirs(OUT, exist_ok=True)
    bev_mod, head_mod = load_reference()
    A = ref_import.AttrDict.wrap
    torch.manual_seed(11)
    bev = bev_mod.BaseBEVBackbone(A(BEV2D), 48).eval()
    head = head_mod.CenterHead(A(HEAD), bev.num_bev_features, len(CLASSES), CLASSES, np.array(GRID), np.array(PCR),
                               VOXEL, predict_boxes_when_training=False).eval()
    with torch.no_grad():
        for m in list(bev.modules()) + list(head.modules()):  # non-trivial BatchNorm statistics
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.2)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.1)
        for h in head.heads_list:  # a heat map with peaks above the score threshold
            h.hm[-1].bias.fill_(-1.0)
            h.hm[-1].weight.mul_(3.0)
        B = 2
        x = torch.randn(B, 48, GRID[1], GRID[0])
        d = bev(dict(spatial_features=x))
        f2d = d["spatial_features_2d"]
        out = head(dict(spatial_features_2d=f2d, batch_size=B))
    pd = head.forward_ret_dict["pred_dicts"][0]
    g = dict(spatial_features=x.numpy(), spatial_features_2d=f2d.numpy(), batch_size=B,
             cfg_json=json.dumps(dict(BEV2D=BEV2D, HEAD=HEAD, CLASSES=CLASSES, PCR=PCR, VOXEL=VOXEL, GRID=GRID,
                                      input_channels=48)),
             bev_keys=json.dumps(list(bev.state_dict().keys())), head_keys=json.dumps(list(head.state_dict().keys())))
    for k, v in pd.items():
        g["raw." + k] = v.numpy()
    for b in range(B):
        for k in ("pred_boxes", "pred_scores", "pred_labels"):
            g["final%d.%s" % (b, k)] = out["final_box_dicts"][b][k].numpy()
    g.update({"bev." + k: v.numpy() for k, v in bev.state_dict().items()})
    g.update({"head." + k: v.numpy() for k, v in head.state_dict().items()})
    np.savez_compressed(os.path.join(OUT, "det_bev_head.npz"), **g)
    print("det_bev_head:", f2d.shape, [out["final_box_dicts"][b]["pred_boxes"].shape for b in range(B)])
    # the full detector's state-dict keys as the reference modules name them (checkpoint compatibility by key):
    # vfe / backbone_3d keys are pinned by the round-1 goldens; here backbone_2d.* and dense_head.* of mssvt.yaml
    print("head keys", len(head.state_dict()), "bev keys", len(bev.state_dict()))
    gen_head_training_step(head_mod)


if __name__ == "__main__":
    main()
