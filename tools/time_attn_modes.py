"""Attention call of Blocks 0 / 1 on the bench frame: fp32 trio vs the single-launch split-fp16 kernel vs bf16."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mssvt_amd import config, fused  # noqa: E402
from mssvt_amd.mssvt_utils import SparseTensor  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = config.build_backbone_from_cfg().to(dev).eval()
_, _, vc, feats = bench.make_inputs(160000, batch, 0, dev)
with torch.no_grad():
    sp = SparseTensor(features=feats, indices=vc.int().contiguous(), spatial_shape=net.grid_size, voxel_size=net.voxel_size,
                      point_cloud_range=net.point_cloud_range, batch_size=batch, hash_size=net.hash_size)
    for bi in (0, 1):
        blk = net.backbone[bi]
        p = fused.two_scale_plan(blk, sp)
        xhat = fused.layer_norm(feats, blk.norm1)
        q_ind, nq, _ = fused._query(blk, p)
        od = fused._work_order(blk, p, nq, feats.shape[0])
        qbuf = fused._query_scratch(p, od["row_cap"], blk.ms_attn, dev)
        nw = int(p.num_wins.item())
        valid = (q_ind[:nw] >= 0).reshape(-1)
        rows = {}
        for mode in ("f32", "f16x3", "bf16"):
            blk.attn_dtype = "bf16" if mode == "bf16" else "f32"
            blk.attn_arith = "f16x3" if mode == "f16x3" else "f32"
            attn = torch.zeros((p.cap * nq + 1, 128), dtype=torch.float32, device=dev)
            ms = bench.event_time_ms(lambda: fused._attention_call(blk, p, od, 128, nq, xhat, qbuf, attn), 20)
            rows[mode] = attn[:nw * nq][valid]
            print("block %d %-6s %.1f us" % (bi, mode, ms * 1e3))
        a = rows["f32"]
        for mode in ("f16x3", "bf16"):
            b = rows[mode]
            print("   %s vs f32: max |diff| %.3e (max |f32| %.2f), rms rel %.3e" % (
                mode, float((a - b).abs().max()), float(a.abs().max()), float((a - b).pow(2).mean().sqrt() / a.pow(2).mean().sqrt())))
