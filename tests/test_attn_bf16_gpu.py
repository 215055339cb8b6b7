"""BASELINE configs[2]: window attention with bf16 matrix-core operands (csrc/block_attn_bf16.hip).

The reference computes in fp32 (mssvt_utils.py:112-150); bf16 operands are this build's extension, so fp32 stays the
parity path (tests/test_module_gpu.py, 1e-3 ceiling) and the bf16 variant gets its OWN stated tolerance
(DESIGN.md section 2): indices stay bit-exact (index work is untouched); features of a whole backbone forward

    max |err|  <=  BF16_MAX * max |ref|          and          rms(err)  <=  BF16_RMS * rms(ref)

against the fp32 result (oracle or fp32 fused path).  Only MFMA operands are rounded (tokens, weights, Q', K', V',
P, O: relative 2^-9 each); accumulation, softmax, positional MLP, interpolation, FFN and LayerNorm are fp32.
"""
import json
import os

import numpy as np
import pytest
import torch

from mssvt_amd import synthetic
from oracle import block_ref

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF16_MAX, BF16_RMS = 3e-2, 6e-3  # observed: see DESIGN.md section 2 (about a third of these)


def bf16_close(got, want, what=""):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape
    err = got - want
    mx, rms = np.abs(err).max() / max(np.abs(want).max(), 1e-30), np.sqrt((err ** 2).mean() / max((want ** 2).mean(), 1e-30))
    print("bf16 %s: max err / max ref = %.3e, rms err / rms ref = %.3e" % (what, mx, rms))
    assert mx <= BF16_MAX and rms <= BF16_RMS, (what, mx, rms)
    return mx, rms


def _cfg(params, hash_size, nout):
    from mssvt_amd.config import Config
    return Config.wrap(dict(NAME="MixedScaleSparseTransformer", HASH_SIZE=hash_size, NUM_OUTPUT_FEATURES=nout,
                            PARAMS=params))


def test_bf16_backbone_small_scene_vs_oracle():
    """The benchmark configuration's shapes (C = 128, heads [4,4] / [8], FF 256: the (64, 16) instantiation) on 20k-point
    scenes, batch 2, against the fp32 CPU oracle."""
    from mssvt_amd import config
    B = 2
    pts = synthetic.make_batch_points(20000, B, 300)
    vc, _, _ = synthetic.voxelize_numpy(pts)
    feats = torch.randn(vc.shape[0], 128, generator=torch.Generator().manual_seed(5)).numpy()
    torch.manual_seed(0)
    cfg = config.load_yaml(config.DEFAULT_CFG)
    net = config.build_backbone_from_cfg(cfg).eval()
    sd = {k: v.numpy() for k, v in net.state_dict().items()}
    want = block_ref.backbone_forward(sd, [dict(p) for p in cfg.MODEL.BACKBONE_3D.PARAMS], feats, vc, B,
                                      synthetic.GRID_SIZE, synthetic.VOXEL_SIZE, synthetic.POINT_CLOUD_RANGE, 400000)
    net = net.to(DEV).set_attn_dtype("bf16")
    from mssvt_amd import fused
    assert fused.attn_uses_bf16(net.backbone[0])
    with torch.no_grad():
        batch = dict(voxel_features=torch.from_numpy(feats).to(DEV), voxel_coords=torch.from_numpy(vc).to(DEV), batch_size=B)
        sp = net(dict(batch))["encoded_spconv_tensor"]
        sp2 = net(dict(batch))["encoded_spconv_tensor"]
    np.testing.assert_array_equal(sp.indices.cpu().numpy(), want.indices)  # index work is untouched: bit-exact
    assert torch.equal(sp.features, sp2.features), "run-to-run deterministic"
    bf16_close(sp.features.cpu().numpy(), want.features, "backbone 20k x 2 vs oracle")
    # and it really is a different arithmetic from the fp32 path
    with torch.no_grad():
        f32 = net.set_attn_dtype("f32")(dict(batch))["encoded_spconv_tensor"]
    assert not torch.equal(f32.features, sp.features)


def test_bf16_batch8_vs_oracle():
    """configs[2]'s batch shape against the fp32 CPU ORACLE itself (not the fp32 fused path): 8 scenes of 20k points in one
    batch -- the size the oracle finishes in seconds -- with the bf16 tolerance above; indices bit-exact."""
    from mssvt_amd import config, fused
    B = 8
    pts = synthetic.make_batch_points(20000, B, 800)
    vc, _, _ = synthetic.voxelize_numpy(pts)
    feats = torch.randn(vc.shape[0], 128, generator=torch.Generator().manual_seed(8)).numpy()
    torch.manual_seed(0)
    cfg = config.load_yaml(config.DEFAULT_CFG)
    net = config.build_backbone_from_cfg(cfg).eval()
    sd = {k: v.numpy() for k, v in net.state_dict().items()}
    want = block_ref.backbone_forward(sd, [dict(p) for p in cfg.MODEL.BACKBONE_3D.PARAMS], feats, vc, B,
                                      synthetic.GRID_SIZE, synthetic.VOXEL_SIZE, synthetic.POINT_CLOUD_RANGE, 400000)
    net = net.to(DEV).set_attn_dtype("bf16")
    assert fused.attn_uses_bf16(net.backbone[0])
    with torch.no_grad():
        sp = net(dict(voxel_features=torch.from_numpy(feats).to(DEV), voxel_coords=torch.from_numpy(vc).to(DEV),
                      batch_size=B))["encoded_spconv_tensor"]
    np.testing.assert_array_equal(sp.indices.cpu().numpy(), want.indices)
    assert sp.v_bs_cnt.tolist() == [int((want.indices[:, 0] == b).sum()) for b in range(B)]
    bf16_close(sp.features.cpu().numpy(), want.features, "backbone 20k x 8 vs oracle")


def test_bf16_batch8_full_size_vs_fp32_fused():
    """BASELINE configs[2] at full size: 8 x 160k-point scenes in one batch, bf16-operand attention against the fp32
    fused path on the same frame (itself pinned to the oracle at full size in tests/test_module_gpu.py)."""
    from mssvt_amd import config
    from mssvt_amd.dist import scene_seeds
    B = 8
    pts = synthetic.make_batch_points(160000, B, seed0=scene_seeds(0, B)[0])
    vc, _, _ = synthetic.voxelize_numpy(pts)
    feats = torch.randn(vc.shape[0], 128, generator=torch.Generator().manual_seed(1000)).to(DEV)
    vct = torch.from_numpy(vc).to(DEV)
    torch.manual_seed(0)
    net = config.build_backbone_from_cfg().to(DEV).eval()
    with torch.no_grad():
        want = net.set_attn_dtype("f32")(dict(voxel_features=feats, voxel_coords=vct, batch_size=B))["encoded_spconv_tensor"]
        got = net.set_attn_dtype("bf16")(dict(voxel_features=feats, voxel_coords=vct, batch_size=B))["encoded_spconv_tensor"]
    assert got.features.shape[0] > 8 * 30000 and torch.equal(got.indices, want.indices)
    bf16_close(got.features.cpu().numpy(), want.features.cpu().numpy(), "8 x 160k vs fp32 fused")


@pytest.mark.parametrize("name", ["block_odd_interp", "block_even_interp", "block_all_interp", "block_trunc",
                                  "block_k64_heads44", "block_enlarged_stride1", "block_empty_sample"])
def test_bf16_block_vs_reference_golden(golden_dir, name):
    """Single Blocks of the reference-run goldens (small C: the (16,8) / (8,8)... instantiations, K = 64, stride-1
    queries, an empty sample) with bf16 operands where the shape is instantiated."""
    from tests.test_module_gpu import build_block, load, make_sp
    from mssvt_amd import fused
    d, sd = load(golden_dir, name)
    blk = build_block(d, sd, "block")
    blk.attn_dtype = "bf16"
    if not fused.attn_uses_bf16(blk):
        pytest.skip("shape not instantiated for bf16: the fp32 kernels run")
    with torch.no_grad():
        out = blk(make_sp(d))
    bf16_close(out.features.cpu().numpy(), d["out_features"], name)


def test_bf16_attention_rows_vs_fp32_kernels():
    """Kernel level: the attention rows of one Block (before interpolation / FFN) from the bf16 launch against the
    three fp32 launches on the same plan: max |err| <= 5e-2 * max |row values|, rms(err) <= 1e-2 * rms (observed 2.2e-2 / see DESIGN.md)."""
    from mssvt_amd import config, fused
    from mssvt_amd.mssvt_utils import SparseTensor
    pts = synthetic.make_batch_points(40000, 1, 9)
    vc, _, _ = synthetic.voxelize_numpy(pts)
    torch.manual_seed(0)
    net = config.build_backbone_from_cfg().to(DEV).eval()
    blk = net.backbone[0]
    feats = torch.randn(vc.shape[0], 128, generator=torch.Generator().manual_seed(2)).to(DEV)
    with torch.no_grad():
        sp = SparseTensor(features=feats, indices=torch.from_numpy(vc).to(DEV).int().contiguous(),
                          spatial_shape=net.grid_size, voxel_size=net.voxel_size,
                          point_cloud_range=net.point_cloud_range, batch_size=1, hash_size=net.hash_size)
        p = fused.two_scale_plan(blk, sp)
        xhat = fused.layer_norm(feats, blk.norm1)
        q_ind, nq, _ = fused._query(blk, p)
        od = fused._work_order(blk, p, nq, feats.shape[0])
        qbuf = fused._query_scratch(p, od["row_cap"], blk.ms_attn, feats.device)
        rows = {}
        blk.attn_kv16 = blk.attn_qo16 = False  # "f32" = the fp32 matrix instruction
        for dt in ("f32", "bf16"):
            blk.attn_dtype = dt
            attn = torch.zeros((p.cap * nq + 1, 128), dtype=torch.float32, device=DEV)
            fused._attention_call(blk, p, od, 128, nq, xhat, qbuf, attn)
            rows[dt] = attn
        nw = int(p.num_wins.item())
        valid = (q_ind[:nw] >= 0).reshape(-1)
    a, b = rows["f32"][:nw * nq][valid], rows["bf16"][:nw * nq][valid]
    assert a.shape[0] > 5000 and float(a.abs().max()) > 0
    assert torch.equal(rows["f32"][:nw * nq][~valid], rows["bf16"][:nw * nq][~valid])  # untouched rows stay untouched
    err = float((a - b).abs().max()) / float(a.abs().max())
    rms = float((a - b).pow(2).mean().sqrt() / a.pow(2).mean().sqrt())
    print("attention rows: max err / max = %.3e, rms err / rms = %.3e" % (err, rms))
    assert err <= 5e-2 and rms <= 1e-2


@pytest.mark.parametrize("seed,points,batch", [(9, 40000, 1), (4, 30000, 2)])
def test_split_f16_window_launch_attention_rows_vs_fp32_kernels(seed, points, batch):
    """mssvt_block_attention_kv16 (the default of the fp32 path): launch B with split-fp16 matrix operands (k_attn_kvh),
    Qt pre-split by launch A.  Same plan, against the three fp32-MFMA launches, with the fp32 parity path's tolerance
    |diff| <= 1e-5 max|ref| + 1e-4 |ref|; rows of invalid queries stay untouched; both query patterns."""
    from mssvt_amd import config, fused
    from mssvt_amd.mssvt_utils import SparseTensor
    pts = synthetic.make_batch_points(points, batch, seed)
    vc, _, _ = synthetic.voxelize_numpy(pts)
    torch.manual_seed(0)
    net = config.build_backbone_from_cfg().to(DEV).eval()
    feats = torch.randn(vc.shape[0], 128, generator=torch.Generator().manual_seed(2)).to(DEV)
    with torch.no_grad():
        sp = SparseTensor(features=feats, indices=torch.from_numpy(vc).to(DEV).int().contiguous(),
                          spatial_shape=net.grid_size, voxel_size=net.voxel_size,
                          point_cloud_range=net.point_cloud_range, batch_size=batch, hash_size=net.hash_size)
        for blk in (net.backbone[0], net.backbone[1]):
            p = fused.two_scale_plan(blk, sp)
            xhat = fused.layer_norm(feats, blk.norm1)
            q_ind, nq, _ = fused._query(blk, p)
            od = fused._work_order(blk, p, nq, feats.shape[0])
            qbuf = fused._query_scratch(p, od["row_cap"], blk.ms_attn, feats.device)
            assert fused._attn_kv16_ok(blk, fused._attn_refs(blk, None), p)
            assert fused._attn_refs(blk, None)["kv16_packed"] is not None
            rows = {}
            for mode in ("f32", "kv16", "kv16+qo16"):  # window launch alone / all three launches on split-fp16 operands
                blk.attn_kv16, blk.attn_qo16 = mode != "f32", mode == "kv16+qo16"
                attn = torch.zeros((p.cap * nq + 1, 128), dtype=torch.float32, device=DEV)
                fused._attention_call(blk, p, od, 128, nq, xhat, qbuf, attn)
                rows[mode] = attn
            nw = int(p.num_wins.item())
            valid = (q_ind[:nw] >= 0).reshape(-1)
            a = rows["f32"][:nw * nq][valid]
            assert a.shape[0] > 1000
            tol = 1e-5 * float(a.abs().max()) + 1e-4 * a.abs()
            for mode in ("kv16", "kv16+qo16"):
                b = rows[mode][:nw * nq][valid]
                assert not torch.equal(a, b)
                assert torch.equal(rows["f32"][:nw * nq][~valid], rows[mode][:nw * nq][~valid])
                assert bool(((a - b).abs() <= tol).all()), (mode, float(((a - b).abs() / tol).max()))
                print("%s vs fp32 launches: max |diff| / max |ref| = %.2e" % (mode, float((a - b).abs().max()) / float(a.abs().max())))


def test_split_f16_window_launch_range_guard():
    """Key tokens or Qt that could leave the fp16 range (bound from the parameters) keep the fp32 launch."""
    from mssvt_amd import config, fused
    torch.manual_seed(0)
    net = config.build_backbone_from_cfg().to(DEV).eval()
    blk = net.backbone[0]

    class P(object):
        coord_bound = 80.0
    r = fused._attn_refs(blk, None)
    assert fused._attn_kv16_ok(blk, r, P)
    with torch.no_grad():
        blk.ms_attn.to_qs[0].weight.mul_(3.0e4)
    assert not fused._attn_kv16_ok(blk, fused._attn_refs(blk, None), P)


def test_split_f16_attention_with_64_keys_matches_fp32_kernels():
    """key_num_sample = 64 at the benchmark's head shape (Cg 64, head dim 16): the window launch's four-tile
    instantiation, fed with Qt fragments by launch A (the Q' hand-off is the K <= 32 form) -- against the fp32-instruction
    launches, fp32 tolerance."""
    from mssvt_amd import config, fused
    from mssvt_amd.mssvt_utils import SparseTensor
    cfg = config.load_yaml(config.DEFAULT_CFG)
    params = [dict(p) for p in cfg.MODEL.BACKBONE_3D.PARAMS]
    for p_ in params:
        p_["key_num_sample"] = 64
    cfg.MODEL.BACKBONE_3D.PARAMS = params
    pts = synthetic.make_batch_points(60000, 1, 21)
    vc, _, _ = synthetic.voxelize_numpy(pts)
    torch.manual_seed(0)
    net = config.build_backbone_from_cfg(cfg).to(DEV).eval()
    feats = torch.randn(vc.shape[0], 128, generator=torch.Generator().manual_seed(3)).to(DEV)
    with torch.no_grad():
        sp = SparseTensor(features=feats, indices=torch.from_numpy(vc).to(DEV).int().contiguous(),
                          spatial_shape=net.grid_size, voxel_size=net.voxel_size,
                          point_cloud_range=net.point_cloud_range, batch_size=1, hash_size=net.hash_size)
        blk = net.backbone[0]
        assert blk.key_num_sample == 64
        p = fused.two_scale_plan(blk, sp)
        xhat = fused.layer_norm(feats, blk.norm1)
        q_ind, nq, _ = fused._query(blk, p)
        od = fused._work_order(blk, p, nq, feats.shape[0])
        qbuf = fused._query_scratch(p, od["row_cap"], blk.ms_attn, feats.device)
        assert fused._attn_kv16_ok(blk, fused._attn_refs(blk, None), p)
        rows = {}
        for mode in ("f32", "kv16+qo16"):
            blk.attn_kv16 = blk.attn_qo16 = mode != "f32"
            attn = torch.zeros((p.cap * nq + 1, 128), dtype=torch.float32, device=DEV)
            fused._attention_call(blk, p, od, 128, nq, xhat, qbuf, attn)
            rows[mode] = attn
        nw = int(p.num_wins.item())
        valid = (q_ind[:nw] >= 0).reshape(-1)
        a, b = rows["f32"][:nw * nq][valid], rows["kv16+qo16"][:nw * nq][valid]
        assert a.shape[0] > 1000 and not torch.equal(a, b)
        assert int((p.k_mask[1][:nw] == 0).sum(1).max()) > 32  # windows that really use more than two key tiles
        tol = 1e-5 * float(a.abs().max()) + 1e-4 * a.abs()
        assert bool(((a - b).abs() <= tol).all()), float(((a - b).abs() / tol).max())
