"""The training branch of the CenterHead mirror (mssvt_amd/center_head.py: assign_targets, get_loss) against ONE TRAINING STEP of
the reference's own head with its own CenterNet losses (pcdet/models/dense_heads/center_head.py:103-250, 350-378;
pcdet/utils/loss_utils.py:264-386), run on the CPU by oracle/gen_golden_det.py: targets, loss terms, parameter gradients."""
import json
import os

import numpy as np
import pytest
import torch


def _step(golden_dir, device):
    from mssvt_amd.center_head import CenterHead
    d = np.load(os.path.join(golden_dir, "det_head_train.npz"))
    cfg = json.loads(str(d["cfg_json"]))
    head = CenterHead(cfg["HEAD"], cfg["input_channels"], len(cfg["CLASSES"]), cfg["CLASSES"], np.array(cfg["GRID"]),
                      np.array(cfg["PCR"]), cfg["VOXEL"], predict_boxes_when_training=False)
    missing, unexpected = head.load_state_dict({k[5:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("head.")}, strict=True)
    assert not missing and not unexpected
    head = head.to(device).train()
    gt = torch.from_numpy(d["gt_boxes"]).to(device)
    head(dict(spatial_features_2d=torch.from_numpy(d["spatial_features_2d"]).to(device), batch_size=int(d["batch_size"]), gt_boxes=gt))
    assert torch.equal(gt.cpu(), torch.from_numpy(d["gt_boxes"]))  # the caller's boxes are not re-labelled in place
    return d, head


def _check(d, head):
    td = head.forward_ret_dict["target_dicts"]
    for k in ("heatmaps", "target_boxes", "inds", "masks"):
        got = td[k][0].cpu().numpy()
        if k in ("inds", "masks"):
            np.testing.assert_array_equal(got, d["target0." + k], err_msg=k)
        else:
            np.testing.assert_allclose(got, d["target0." + k], rtol=0, atol=1e-6, err_msg=k)
    assert int(td["masks"][0].sum()) == 25 and float(td["heatmaps"][0].max()) == 1.0
    loss, tb = head.get_loss()
    want_tb = json.loads(str(d["tb_json"]))
    assert abs(float(loss.item()) - float(d["loss"])) <= 1e-5 * float(d["loss"])
    for k, v in want_tb.items():
        assert abs(tb[k] - v) <= 1e-5 * max(1.0, abs(v)), k
    loss.backward()
    n = 0
    for k, v in head.named_parameters():
        if "grad." + k in d.files:
            want = d["grad." + k]
            np.testing.assert_allclose(v.grad.cpu().numpy(), want, rtol=1e-4, atol=1e-5 * max(1.0, float(np.abs(want).max())), err_msg=k)
            n += 1
    assert n >= 30


def test_training_step_matches_the_reference_run(golden_dir):
    _check(*_step(golden_dir, "cpu"))


def test_objects_without_size_or_padding_rows_get_no_target(golden_dir):
    d, head = _step(golden_dir, "cpu")
    gt = torch.from_numpy(d["gt_boxes"]).clone()
    gt[0, 3, 3] = 0.0  # no extent along x: skipped (ref :137-138), its slot stays empty
    head(dict(spatial_features_2d=torch.from_numpy(d["spatial_features_2d"]), batch_size=2, gt_boxes=gt))
    td = head.forward_ret_dict["target_dicts"]
    assert int(td["masks"][0][0, 3]) == 0 and float(td["target_boxes"][0][0, 3].abs().sum()) == 0.0
    assert int(td["masks"][0].sum()) == 24
    assert bool(torch.isfinite(td["target_boxes"][0]).all())


@pytest.mark.gpu
@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs the MI355X")
def test_training_step_matches_the_reference_run_on_the_gpu(golden_dir):
    _check(*_step(golden_dir, "cuda"))
