"""Detector shell on the MI355X: the HIP rotated NMS (csrc/nms_bev.hip, through the C ABI) against the CPU oracle,
CenterHead / BaseBEVBackbone with it against the reference-run golden, and the whole mssvt.yaml CenterPoint detector on
a synthetic scene (points -> DynamicVFE -> MsSVT backbone -> HeightCompression -> BEV backbone -> CenterHead -> boxes)."""
import numpy as np
import pytest
import torch

from mssvt_amd import synthetic
from oracle import nms_ref

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _random_boxes(n, seed, spread=20.0):
    rng = np.random.default_rng(seed)
    b = np.zeros((n, 7), np.float32)
    b[:, 0:2] = rng.uniform(-spread, spread, (n, 2))
    b[:, 2] = rng.uniform(-1, 1, n)
    b[:, 3] = rng.uniform(1.5, 5.0, n)
    b[:, 4] = rng.uniform(0.8, 2.5, n)
    b[:, 5] = rng.uniform(1.0, 2.0, n)
    b[:, 6] = rng.uniform(-np.pi, np.pi, n)
    # near duplicates (what NMS exists for): jittered copies of the first quarter
    q = n // 4
    b[q:2 * q] = b[:q] + rng.normal(0, 0.15, (q, 7)).astype(np.float32)
    return b, rng.uniform(0.1, 1.0, n).astype(np.float32)


@pytest.mark.parametrize("n,thresh,seed", [(1, 0.5, 0), (70, 0.1, 1), (130, 0.5, 2), (130, 0.7, 3), (193, 0.25, 4)])
def test_hip_nms_keeps_what_the_oracle_keeps(n, thresh, seed):
    from mssvt_amd import iou3d_nms_utils
    boxes, scores = _random_boxes(n, seed)
    want = nms_ref.nms(boxes, scores, thresh)
    got, _ = iou3d_nms_utils.nms_gpu(torch.from_numpy(boxes).to(DEV), torch.from_numpy(scores).to(DEV), thresh)
    assert got.cpu().numpy().tolist() == want.tolist()
    got2, _ = iou3d_nms_utils.nms_gpu(torch.from_numpy(boxes).to(DEV), torch.from_numpy(scores).to(DEV), thresh, pre_maxsize=40)
    assert got2.cpu().numpy().tolist() == nms_ref.nms(boxes, scores, thresh, 40).tolist()


def test_hip_nms_empty_and_many():
    from mssvt_amd import iou3d_nms_utils
    e, _ = iou3d_nms_utils.nms_gpu(torch.zeros((0, 7), device=DEV), torch.zeros(0, device=DEV), 0.5)
    assert e.numel() == 0
    boxes, scores = _random_boxes(4096, 9, spread=75.0)  # NMS_PRE_MAXSIZE of the Waymo recipe
    got, _ = iou3d_nms_utils.nms_gpu(torch.from_numpy(boxes).to(DEV), torch.from_numpy(scores).to(DEV), 0.7)
    got = got.cpu().numpy()
    assert 1000 < got.size <= 4096 and np.all(np.diff(scores[got]) <= 0)  # best first
    # kept boxes do not suppress each other (spot check on the oracle, 40 of them)
    k = boxes[got[:40]]
    for i in range(40):
        for j in range(i + 1, 40):
            assert float(nms_ref.iou_bev(k[i], k[j])) <= 0.7 + 1e-5


def test_center_head_with_the_hip_nms_matches_the_reference_run(golden_dir):
    from tests.test_detector_cpu import build_from_golden, check_against_golden
    d, bev, head = build_from_golden(golden_dir)
    check_against_golden(d, bev.to(DEV), head.to(DEV), dev=DEV, tol=5e-5)


def test_centerpoint_detector_end_to_end():
    from mssvt_amd import centerpoint
    torch.manual_seed(0)
    det = centerpoint.build_detector().to(DEV).eval()
    with torch.no_grad():
        for h in det.dense_head.heads_list:  # random init: lift the heat map above the score threshold
            h.hm[-1].bias.fill_(0.5)
    B = 2
    pts = torch.from_numpy(synthetic.make_batch_points(40000, B, 123)).to(DEV)
    with torch.no_grad():
        preds, _ = det(dict(points=pts, batch_size=B))
        preds2, _ = det(dict(points=pts, batch_size=B))
    assert len(preds) == B
    for p, q in zip(preds, preds2):
        n = p["pred_boxes"].shape[0]
        assert 0 < n <= 500 and p["pred_boxes"].shape[1] == 7 and p["pred_scores"].shape == (n,) and p["pred_labels"].shape == (n,)
        assert bool(torch.isfinite(p["pred_boxes"]).all()) and int(p["pred_labels"].min()) >= 1 and int(p["pred_labels"].max()) <= 3
        assert torch.equal(p["pred_boxes"], q["pred_boxes"]) and torch.equal(p["pred_scores"], q["pred_scores"])  # deterministic
        assert bool((p["pred_scores"][:-1] >= p["pred_scores"][1:]).all())


def test_centerpoint_detector_training_step():
    """The whole detector in train mode (ref detectors/centerpoint.py:9-32): points + ground-truth boxes -> DynamicVFE (batch
    statistics, differentiable reductions) -> MsSVT backbone (compact training path) -> BEV backbone -> CenterHead targets and
    losses -> backward.  Every stage's parameters receive a finite gradient; two identical steps agree up to
    the order of the framework's atomic sums (the differentiable VFE reductions, MIOpen's convolutions)."""
    from mssvt_amd import centerpoint
    B = 2
    pts = torch.from_numpy(synthetic.make_batch_points(20000, B, 77)).to(DEV)
    rng = np.random.default_rng(3)
    gt = np.zeros((B, 10, 8), np.float32)
    for b in range(B):
        n = 8 - 2 * b
        gt[b, :n, 0:2] = rng.uniform(-40, 40, (n, 2))
        gt[b, :n, 2] = rng.uniform(-1, 1, n)
        gt[b, :n, 3:6] = rng.uniform([1.5, 0.8, 1.0], [5.0, 2.5, 2.0], (n, 3))
        gt[b, :n, 6] = rng.uniform(-3.1, 3.1, n)
        gt[b, :n, 7] = rng.integers(1, 4, n)
    losses = []
    for _ in range(2):
        torch.manual_seed(0)
        det = centerpoint.build_detector().to(DEV).train()
        ret, tb, _ = det(dict(points=pts, batch_size=B, gt_boxes=torch.from_numpy(gt).to(DEV)))
        loss = ret["loss"]
        assert bool(torch.isfinite(loss)) and loss.item() > 0 and abs(tb["loss_rpn"] - loss.item()) < 1e-4 * loss.item()
        loss.backward()
        groups = {"vfe": 0, "backbone_3d": 0, "backbone_2d": 0, "dense_head": 0}
        for k, v in det.named_parameters():
            g = k.split(".")[0]
            if v.grad is not None and g in groups:
                assert bool(torch.isfinite(v.grad).all()), k
                groups[g] += int(v.grad.abs().sum() > 0)
        assert all(n > 0 for n in groups.values()), groups
        losses.append(loss.item())
    assert abs(losses[0] - losses[1]) <= 1e-5 * abs(losses[0])
