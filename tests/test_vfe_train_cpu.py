"""The training branch of the DynamicVFE mirror (mssvt_amd/dynamic_vfe.py: train mode, autograd on) against ONE TRAINING STEP of
the reference's own module (pcdet/models/backbones_3d/vfe/dynamic_vfe.py:71-131, run on the CPU by oracle/gen_golden_vfe.py
with torch_scatter's two reductions restated): output, every parameter gradient, the BatchNorm buffers after the step."""
import os

import numpy as np
import pytest
import torch


def _step(golden_dir, device):
    from mssvt_amd.dynamic_vfe import DynamicVFE
    d = np.load(os.path.join(golden_dir, "dynamic_vfe_train_32_64.npz"))
    cfg = dict(NUM_FILTERS=d["num_filters"].tolist(), WITH_CLUSTER_CENTER=True, WITH_VOXEL_CENTER=True)
    vfe = DynamicVFE(cfg, 5, d["voxel_size"].tolist(), d["grid_size"].tolist(), d["point_cloud_range"].tolist())
    missing, unexpected = vfe.load_state_dict({k[3:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("sd.")}, strict=True)
    assert not missing and not unexpected  # identical state-dict keys
    vfe = vfe.to(device).train()
    out = vfe(dict(points=torch.from_numpy(d["points"]).to(device), batch_size=int(d["batch_size"])))
    (out["voxel_features"] * torch.from_numpy(d["weight_of_the_loss"]).to(device)).sum().backward()
    return d, vfe, out


def _check(d, vfe, out):
    np.testing.assert_array_equal(out["voxel_coords"].cpu().numpy(), d["voxel_coords"])
    np.testing.assert_allclose(out["voxel_features"].detach().cpu().numpy(), d["voxel_features"], rtol=1e-5, atol=1e-5)
    scale = max(float(np.abs(d[k]).max()) for k in d.files if k.startswith("grad."))
    for k, v in vfe.named_parameters():
        want = d["grad." + k]
        assert v.grad is not None, k
        if k.endswith(".0.bias"):
            # a Linear bias in front of a BatchNorm on batch statistics has NO gradient (the mean is removed): what both
            # runs hold is the rounding residue of sums of terms of the weights' gradient scale
            assert float(v.grad.abs().max()) <= 1e-6 * scale and float(np.abs(want).max()) <= 1e-6 * scale, k
            continue
        # gradients are sums over 6 000 points: tolerance relative to the tensor's scale
        np.testing.assert_allclose(v.grad.cpu().numpy(), want, rtol=1e-4, atol=1e-5 * max(1.0, float(np.abs(want).max())), err_msg=k)
    for k, v in vfe.state_dict().items():
        if "after." + k in d.files:
            np.testing.assert_allclose(v.cpu().numpy(), d["after." + k], rtol=1e-5, atol=1e-6, err_msg=k)


def test_training_step_matches_the_reference_run(golden_dir):
    _check(*_step(golden_dir, "cpu"))


def test_eval_mode_stays_forward_only_and_train_mode_moves_the_statistics(golden_dir):
    d, vfe, _ = _step(golden_dir, "cpu")
    before = vfe.pfn[0][1].running_mean.clone()
    with torch.no_grad():  # train mode without autograd: the batch statistics still update (as in the reference)
        vfe(dict(points=torch.from_numpy(d["points"]), batch_size=int(d["batch_size"])))
    assert not torch.equal(before, vfe.pfn[0][1].running_mean)


@pytest.mark.gpu
@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs the MI355X")
def test_training_step_matches_the_reference_run_on_the_gpu(golden_dir):
    _check(*_step(golden_dir, "cuda"))
