"""Detector shell behind the backbone on the CPU: the BaseBEVBackbone / CenterHead mirrors
against runs of the reference's own modules (tests/golden/det_bev_head.npz, oracle/gen_golden_det.py: identical
state-dict keys, outputs within fp32 re-association), checkpoint loading by key, and the NMS oracle's known answers."""
import json
import os

import numpy as np
import pytest
import torch

from mssvt_amd.base_bev_backbone import BaseBEVBackbone
from mssvt_amd.center_head import CenterHead
from mssvt_amd.config import Config
from oracle import nms_ref


def _oracle_nms(boxes, scores, thresh, pre_maxsize=None, **kw):
    keep = nms_ref.nms(boxes.cpu().numpy(), scores.cpu().numpy(), float(thresh), pre_maxsize)
    return torch.from_numpy(np.asarray(keep, dtype=np.int64)).to(boxes.device), None


def build_from_golden(golden_dir):
    d = np.load(os.path.join(golden_dir, "det_bev_head.npz"))
    cfg = json.loads(str(d["cfg_json"]))
    bev = BaseBEVBackbone(Config.wrap(cfg["BEV2D"]), cfg["input_channels"]).eval()
    head = CenterHead(Config.wrap(cfg["HEAD"]), bev.num_bev_features, len(cfg["CLASSES"]), cfg["CLASSES"],
                      np.array(cfg["GRID"]), np.array(cfg["PCR"]), cfg["VOXEL"], predict_boxes_when_training=False).eval()
    assert list(bev.state_dict().keys()) == json.loads(str(d["bev_keys"]))  # a reference checkpoint loads by key
    assert list(head.state_dict().keys()) == json.loads(str(d["head_keys"]))
    bev.load_state_dict({k[4:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("bev.")}, strict=True)
    head.load_state_dict({k[5:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("head.")}, strict=True)
    return d, bev, head


def check_against_golden(d, bev, head, dev="cpu", tol=2e-5):
    with torch.no_grad():
        out = bev(dict(spatial_features=torch.from_numpy(d["spatial_features"]).to(dev)))
        f2d = out["spatial_features_2d"]
        np.testing.assert_allclose(f2d.cpu().numpy(), d["spatial_features_2d"], rtol=1e-4, atol=tol)
        # the head on the golden's own BEV features: box lists depend on score order, keep the inputs identical
        res = head(dict(spatial_features_2d=torch.from_numpy(d["spatial_features_2d"]).to(dev), batch_size=int(d["batch_size"])))
    raw = head.forward_ret_dict["pred_dicts"][0]
    for k in ("hm", "center", "center_z", "dim", "rot"):
        np.testing.assert_allclose(raw[k].cpu().numpy(), d["raw." + k], rtol=1e-4, atol=tol)
    for b in range(int(d["batch_size"])):
        got = res["final_box_dicts"][b]
        want_boxes, want_scores = d["final%d.pred_boxes" % b], d["final%d.pred_scores" % b]
        assert got["pred_boxes"].shape == want_boxes.shape  # the same boxes survive the NMS
        np.testing.assert_allclose(got["pred_scores"].cpu().numpy(), want_scores, rtol=1e-4, atol=tol)
        # peaks with EQUAL scores (a saturated heat map) may come out of the top-K in another order: compare the rows
        # as sets (sorted by position)
        rows = lambda bx, lb: np.concatenate([bx, lb[:, None].astype(np.float32)], 1)  # noqa: E731
        g = rows(got["pred_boxes"].cpu().numpy(), got["pred_labels"].cpu().numpy())
        w = rows(want_boxes, d["final%d.pred_labels" % b])
        g, w = g[np.lexsort((g[:, 1], g[:, 0]))], w[np.lexsort((w[:, 1], w[:, 0]))]
        np.testing.assert_allclose(g, w, rtol=1e-4, atol=1e-4)


def test_bev_backbone_and_center_head_match_the_reference_run(golden_dir):
    d, bev, head = build_from_golden(golden_dir)
    head.nms_fn = _oracle_nms  # the HIP NMS needs the GPU (tests/test_detector_gpu.py); here the CPU oracle stands in
    check_against_golden(d, bev, head)


def test_center_head_training_mode_needs_ground_truth(golden_dir):
    """train mode assigns targets (tests/test_head_train_cpu.py pins them to the reference): without gt_boxes it fails loudly"""
    _, _, head = build_from_golden(golden_dir)
    head.train()
    with pytest.raises(KeyError):
        head(dict(spatial_features_2d=torch.zeros(1, 64, 8, 8), batch_size=1))


def test_detector_topology_keys_and_checkpoint_by_key(tmp_path):
    """mssvt.yaml -> CenterPoint: modules registered under the reference's topology names (state-dict prefixes), a
    checkpoint written in the reference's format loads by key, entries with another shape are skipped and reported."""
    from mssvt_amd import centerpoint
    torch.manual_seed(0)
    det = centerpoint.build_detector()
    sd = det.state_dict()
    prefixes = {k.split(".")[0] for k in sd}
    assert prefixes == {"global_step", "vfe", "backbone_3d", "map_to_bev_module", "backbone_2d", "dense_head"}
    assert "backbone_3d.backbone.0.ms_attn.to_qs.0.weight" in sd and "backbone_3d.backbone.4.pos_proj.2.weight" in sd
    assert "backbone_2d.blocks.0.1.weight" in sd and "backbone_2d.deblocks.1.0.weight" in sd
    assert "dense_head.shared_conv.0.weight" in sd and "dense_head.heads_list.0.hm.1.bias" in sd
    ckpt = {"epoch": 3, "it": 100, "model_state": {k: v.clone() for k, v in sd.items()}, "optimizer_state": None,
            "version": "pcdet+0.5.2"}
    ckpt["model_state"]["dense_head.heads_list.0.hm.1.bias"] = torch.zeros(7)  # another class count: must be skipped
    ckpt["model_state"]["roi_head.some.weight"] = torch.zeros(3)  # a module this config does not have
    path = os.path.join(tmp_path, "checkpoint_epoch_3.pth")
    torch.save(ckpt, path)
    torch.manual_seed(1)
    det2 = centerpoint.build_detector()
    assert not torch.equal(det2.state_dict()["backbone_3d.backbone.0.linear1.weight"], sd["backbone_3d.backbone.0.linear1.weight"])
    loaded, total, missed = det2.load_params_from_file(path, to_cpu=True)
    assert total == len(sd) and loaded == total - 1 and missed == ["dense_head.heads_list.0.hm.1.bias"]
    sd2 = det2.state_dict()
    for k in sd:
        if k not in missed:
            assert torch.equal(sd2[k], sd[k]), k


def test_nms_oracle_known_answers():
    """Hand-derived cases of the rotated-overlap restatement (oracle/nms_ref.py): identical boxes, a half shift,
    disjoint boxes, a square turned by 45 degrees inside its twin, and the greedy keep order."""
    a = [0, 0, 0, 4, 2, 1, 0.0]
    assert abs(float(nms_ref.iou_bev(a, a)) - 1.0) < 1e-6
    assert abs(float(nms_ref.box_overlap(a, [1, 0, 0, 4, 2, 1, 0.0])) - 6.0) < 1e-5  # 3 x 2 overlap
    assert float(nms_ref.iou_bev(a, [10, 10, 0, 1, 1, 1, 0.3])) == 0.0
    sq, tilted = [0, 0, 0, 2, 2, 1, 0.0], [0, 0, 0, 2, 2, 1, np.pi / 4]
    assert abs(float(nms_ref.box_overlap(tilted, sq)) - 8 * (np.sqrt(2) - 1)) < 1e-4  # a regular octagon
    boxes = np.array([a, [0.2, 0, 0, 4, 2, 1, 0.05], [8, 8, 0, 2, 2, 1, 1.0], [0, 3.5, 0, 4, 2, 1, 0.0]], np.float32)
    scores = np.array([0.9, 0.8, 0.95, 0.3])
    assert nms_ref.nms(boxes, scores, 0.5).tolist() == [2, 0, 3]  # box 1 overlaps box 0 and goes
