"""Training path: the compact differentiable forward + deterministic segmented-sum backward
(mssvt_amd/train_path.py, csrc/segment_reduce.hip) against the padded operator-level path (the reference's own
structure, whose gradients are pinned to the reference's backward run in tests/test_module_gpu.py), and run to run."""
import numpy as np
import pytest
import torch

from mssvt_amd import synthetic

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _net(C, params, H, seed=3):
    from mssvt_amd.config import Config
    from mssvt_amd.mssvt_backbone import MixedScaleSparseTransformer
    torch.manual_seed(seed)
    cfg = Config.wrap(dict(NAME="MixedScaleSparseTransformer", HASH_SIZE=H, NUM_OUTPUT_FEATURES=C, PARAMS=params))
    return MixedScaleSparseTransformer(cfg, C, synthetic.GRID_SIZE, synthetic.VOXEL_SIZE,
                                       synthetic.POINT_CLOUD_RANGE).to(DEV).eval()  # eval: DropPath off, grads on


def _params(C, heads, cheads, interp=(True, True)):
    blk = dict(name="MixedScaleSparseTransformerBlock", channels=[C, 2 * C, C], num_heads=heads,
               window_size=[[3, 3, 5], [7, 7, 7]], max_num_win1=45, max_num_win2=343, cbs_mode="odd_even",
               key_num_sample=32)
    return [dict(blk, cbs_pattern=1, use_feature_interpolation=interp[0]),
            dict(blk, cbs_pattern=0, use_feature_interpolation=interp[1]),
            dict(name="MixedScaleSparseTransformerCompressBlock", channels=[C, 2 * C, C], num_heads=cheads,
                 window_size=[[1, 1, 32]], max_num_win1=32)]


def _grads(net, x, coords, B, w):
    for p in net.parameters():
        p.grad = None
    x = x.detach().clone().requires_grad_(True)
    out = net(dict(voxel_features=x, voxel_coords=coords, batch_size=B))["encoded_spconv_tensor"].features
    (out * w).sum().backward()
    return out.detach(), x.grad.detach().clone(), {k: p.grad.detach().clone() for k, p in net.named_parameters()}


@pytest.mark.parametrize("C,heads,cheads,interp", [(32, [2, 2], [4], (True, False)), (64, [4, 4], [8], (True, True))])
def test_compact_training_path_matches_the_operator_path(C, heads, cheads, interp):
    from mssvt_amd import fused
    B, H = 2, 40009
    net = _net(C, _params(C, heads, cheads, interp), H)
    vc, _, _ = synthetic.voxelize_numpy(synthetic.make_batch_points(4000, B, 31))
    coords = torch.from_numpy(vc).to(DEV)
    x = torch.randn(vc.shape[0], C, device=DEV)
    g = torch.Generator(device="cpu").manual_seed(7)
    res = {}
    for compact in (True, False):
        fused.TRAIN_COMPACT = compact
        try:
            out_shape = net(dict(voxel_features=x, voxel_coords=coords, batch_size=B))["encoded_spconv_tensor"].features.shape
            w = torch.randn(out_shape, generator=torch.Generator().manual_seed(7)).to(DEV)
            res[compact] = _grads(net, x, coords, B, w)
        finally:
            fused.TRAIN_COMPACT = True
    (o1, gx1, gp1), (o2, gx2, gp2) = res[True], res[False]

    def close(a, b, what):
        scale = max(1.0, float(b.abs().max()))
        err = float((a - b).abs().max())
        assert err <= 2e-4 * scale, "%s: %.3e vs scale %.3e" % (what, err, scale)

    close(o1, o2, "output")
    close(gx1, gx2, "input gradient")
    for k in gp2:
        close(gp1[k], gp2[k], k)


def test_training_gradients_are_bit_identical_run_to_run():
    """Segmented sums in a fixed order instead of atomics: two backward passes of the same step agree bit for bit
    (input gradient and every parameter gradient), at a size where atomics would reorder (20k points x 2)."""
    C, B, H = 64, 2, 400009
    net = _net(C, _params(C, [4, 4], [8]), H)
    vc, _, _ = synthetic.voxelize_numpy(synthetic.make_batch_points(20000, B, 5))
    coords = torch.from_numpy(vc).to(DEV)
    x = torch.randn(vc.shape[0], C, device=DEV)
    n_out = net(dict(voxel_features=x, voxel_coords=coords, batch_size=B))["encoded_spconv_tensor"].features.shape
    w = torch.randn(n_out, generator=torch.Generator().manual_seed(1)).to(DEV)
    a = _grads(net, x, coords, B, w)
    junk = torch.randn(64 << 20, device=DEV)  # different allocator state / cache contents between the runs
    del junk
    b = _grads(net, x, coords, B, w)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for k in a[2]:
        assert torch.equal(a[2][k], b[2][k]), k


def test_segment_sum_rows_matches_index_add():
    from mssvt_amd import train_path
    torch.manual_seed(0)
    R, N, C = 5000, 700, 48
    src = torch.randn(R, C, device=DEV)
    idx = torch.randint(0, N, (R,), device=DEV)
    csr = train_path.Csr.gather(idx, N)  # dst[i] = table[idx[i]]; its transpose sums rows of src per table row
    got = train_path.segment_sum_rows(src, csr.t_off, csr.t_idx, csr.t_w, N)
    want = torch.zeros(N, C, device=DEV, dtype=torch.float64).index_add_(0, idx, src.double())
    assert float((got.double() - want).abs().max()) < 1e-4
    # weighted form with a row that nothing maps to
    w = torch.rand(R, device=DEV)
    off = torch.arange(R + 1, dtype=torch.int32, device=DEV)
    c2 = train_path.Csr(off, idx.int(), w, N + 3)
    got2 = train_path.segment_sum_rows(src, c2.t_off, c2.t_idx, c2.t_w, N + 3)
    want2 = torch.zeros(N + 3, C, device=DEV, dtype=torch.float64).index_add_(0, idx, src.double() * w.double()[:, None])
    assert float((got2.double() - want2).abs().max()) < 1e-4 and float(got2[N:].abs().max()) == 0.0
