#!/usr/bin/env python3
"""Outputs of every matrix-instruction kernel family of the library on fixed inputs, as one .npz -- run once per BUILD
of the library (MSSVT_LIB=<variant>, mssvt_amd/build.py::VARIANTS) by tests/test_mfma_schedules_gpu.py, which compares the
files.  The arithmetic of a kernel is fixed by its source (explicit FMAs, -ffp-contract=off, association written out):
another optimisation level or scheduler only moves instructions, so every build must give the SAME numbers; a sum that is
read before its last matrix instruction has landed gives different ones in at least one of them.

    MSSVT_LIB=mssvt_amd/lib/variants/libmssvt_hip_O2.so python tools/schedule_probe.py out.npz
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from mssvt_amd import _lib, config, fused, synthetic  # noqa: E402

DEV = "cuda"


def _frame(net, feats, coords, B):
    with torch.no_grad():
        sp = net(dict(voxel_features=feats, voxel_coords=coords, batch_size=B))["encoded_spconv_tensor"]
    return sp.features.float().cpu().numpy()


def main(out_path):
    out = {"lib": np.array(os.path.basename(_lib.LIB_PATH))}
    B = 2
    vc, _, _ = synthetic.voxelize_numpy(synthetic.make_batch_points(20000, B, 77))
    coords = torch.from_numpy(vc).to(DEV)
    # --- the benchmark network (C = 128, FF = 256, heads [4,4] / [8]): the instantiations bench.py runs
    cfg = config.load_yaml(config.DEFAULT_CFG)
    torch.manual_seed(0)
    net = config.build_backbone_from_cfg(cfg).eval().to(DEV)
    feats = torch.randn(vc.shape[0], 128, generator=torch.Generator().manual_seed(5)).to(DEV)
    out["c128_default"] = _frame(net, feats, coords, B)  # k_attn_q16/kvh/o16, k_ffn_ws, k_cmp_ws
    keep = (fused.ATTN_KV16, fused.ATTN_QO16, fused.FFN_ARITH, fused.CMP_WS)
    try:
        fused.CMP_WS = False
        out["c128_cmp3"] = _frame(net, feats, coords, B)  # compress_fused.hip, split-fp16 form
        fused.ATTN_KV16 = fused.ATTN_QO16 = False
        fused.FFN_ARITH = "f32"
        for b in net.backbone:
            b.refresh_weights()
        out["c128_f32"] = _frame(net, feats, coords, B)  # k_attn_q/kv/o, k_ffn_up/down, compress_fused fp32 form
    finally:
        fused.ATTN_KV16, fused.ATTN_QO16, fused.FFN_ARITH, fused.CMP_WS = keep
        for b in net.backbone:
            b.refresh_weights()
    net.set_attn_dtype("bf16")
    out["c128_bf16"] = _frame(net, feats, coords, B)  # block_attn_bf16.hip
    net.set_attn_dtype("f32")
    # --- the training step: linear_rows.hip, linear_wgrad.hip, pair attention (no MFMA), on the same network
    net.train(False)
    x = feats.detach().clone().requires_grad_(True)
    for p in net.parameters():
        p.grad = None
    y = net(dict(voxel_features=x, voxel_coords=coords, batch_size=B))["encoded_spconv_tensor"].features
    w = torch.randn(y.shape, generator=torch.Generator().manual_seed(9)).to(DEV)
    (y * w).sum().backward()
    out["train_out"] = y.detach().cpu().numpy()
    out["train_dx"] = x.grad.cpu().numpy()
    for k, p in net.named_parameters():
        if p.grad is not None and ("linear" in k or "to_" in k or "projs" in k):
            out["grad." + k] = p.grad.cpu().numpy()
    # --- a narrow network (C = 64, heads [4,4] / [8]): other instantiations of the same templates
    blk = dict(name="MixedScaleSparseTransformerBlock", channels=[64, 128, 64], num_heads=[4, 4],
               window_size=[[3, 3, 5], [7, 7, 7]], max_num_win1=45, max_num_win2=343, cbs_mode="odd_even", key_num_sample=32,
               use_feature_interpolation=True)
    params = [dict(blk, cbs_pattern=1), dict(blk, cbs_pattern=0),
              dict(name="MixedScaleSparseTransformerCompressBlock", channels=[64, 128, 64], num_heads=[8],
                   window_size=[[1, 1, 32]], max_num_win1=32)]
    from mssvt_amd.config import Config
    from mssvt_amd.mssvt_backbone import MixedScaleSparseTransformer
    torch.manual_seed(1)
    net64 = MixedScaleSparseTransformer(Config.wrap(dict(HASH_SIZE=400009, NUM_OUTPUT_FEATURES=64, PARAMS=params)), 64,
                                        synthetic.GRID_SIZE, synthetic.VOXEL_SIZE, synthetic.POINT_CLOUD_RANGE).eval().to(DEV)
    f64 = torch.randn(vc.shape[0], 64, generator=torch.Generator().manual_seed(6)).to(DEV)
    out["c64_default"] = _frame(net64, f64, coords, B)
    # --- csrc/linear_rows_h.hip directly: K = 256 (the 8-wave form) and K = 128 (16 waves), forward and dX forms
    from mssvt_amd import train_path
    g = torch.Generator().manual_seed(11)
    for K, N in ((256, 128), (128, 256), (64, 128)):
        xr = (torch.randn(5000, K, generator=g) * torch.exp2(torch.randint(-20, 20, (5000, 1), generator=g).float())).to(DEV)
        wr = torch.randn(N, K, generator=g).to(DEV)
        br = torch.randn(N, generator=g).to(DEV)
        out["linear_rows_h_%d_%d" % (K, N)] = train_path._linear_rows(xr, wr, False, br, True, N, split16=True).cpu().numpy()
        dy = torch.randn(5000, N, generator=g).to(DEV)
        out["linear_rows_h_%d_%d_dx" % (K, N)] = train_path._linear_rows(dy, wr, True, None, False, K, split16=True).cpu().numpy()
    # --- csrc/pfn_sorted.hip (k_ps_pfn2) and csrc/pfn_fused.hip (k_pfn2_h): the DynamicVFE eval forward on the default configuration
    from mssvt_amd import dynamic_vfe
    pts = synthetic.make_batch_points(20000, B, 78)
    torch.manual_seed(2)
    vfe = dynamic_vfe.DynamicVFE(config.Config.wrap({}), 5, synthetic.VOXEL_SIZE, synthetic.GRID_SIZE,
                                 synthetic.POINT_CLOUD_RANGE).eval()
    with torch.no_grad():
        for m in vfe.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.normal_(0, 0.3)
                m.running_var.uniform_(0.5, 1.5)
        assert dynamic_vfe.FUSED_PFN and dynamic_vfe.SORTED_PFN
        vfe = vfe.to(DEV)
        out["vfe_fused"] = vfe(dict(points=torch.from_numpy(pts).to(DEV), batch_size=B))["voxel_features"].cpu().numpy()  # pfn_sorted.hip
        dynamic_vfe.SORTED_PFN = False
        try:
            out["vfe_atomic"] = vfe(dict(points=torch.from_numpy(pts).to(DEV), batch_size=B))["voxel_features"].cpu().numpy()  # pfn_fused.hip
        finally:
            dynamic_vfe.SORTED_PFN = True
    np.savez(out_path, **out)
    print("schedule probe: %d arrays from %s -> %s" % (len(out) - 1, _lib.LIB_PATH, out_path))


if __name__ == "__main__":
    main(sys.argv[1])
