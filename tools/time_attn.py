"""Time the window-attention launches of a Block on the bench scene: the fp32 trio (fp32 MFMA / split-fp16 window launch) against the bf16 kernel,
both query patterns (investigation helper; usage: python tools/time_attn.py [batch] [points])."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mssvt_amd import config, fused  # noqa: E402
from mssvt_amd.mssvt_utils import SparseTensor  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
points = int(sys.argv[2]) if len(sys.argv) > 2 else 160000
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = config.build_backbone_from_cfg().to(dev).eval()
_, _, vc, feats = bench.make_inputs(points, batch, 0, dev)
with torch.no_grad():
    sp = SparseTensor(features=feats, indices=vc.int().contiguous(), spatial_shape=net.grid_size,
                      voxel_size=net.voxel_size, point_cloud_range=net.point_cloud_range, batch_size=batch,
                      hash_size=net.hash_size)
    for bi in (0, 1):
        blk = net.backbone[bi]
        p = fused.two_scale_plan(blk, sp)
        xhat = fused.layer_norm(feats, blk.norm1)
        q_ind, nq, _ = fused._query(blk, p)
        od = fused._work_order(blk, p, nq, feats.shape[0])
        qbuf = fused._query_scratch(p, od["row_cap"], blk.ms_attn, dev)
        outs = {}
        for dt in ("f32", "kv16", "kv16+qo16", "bf16"):
            blk.attn_dtype = "bf16" if dt == "bf16" else "f32"
            blk.attn_kv16 = dt.startswith("kv16")
            blk.attn_qo16 = dt == "kv16+qo16"
            attn = torch.zeros((p.cap * nq + 1, 128), dtype=torch.float32, device=dev)
            ms = bench.event_time_ms(lambda: fused._attention_call(blk, p, od, 128, nq, xhat, qbuf, attn), 20)
            outs[dt] = attn
            print("block %d (cbs_pattern %d, %d windows, %d query rows) %s: %.1f us" % (
                bi, blk.cbs_pattern, int(p.num_wins.item()), int(od["n_rows"].item()), dt, ms * 1e3))
        a, b = outs["f32"], outs["kv16+qo16"]
        print("   kv16+qo16 vs f32: max err / max = %.3e, rms err / rms = %.3e" % (
            float((a - b).abs().max() / a.abs().max()), float((a - b).pow(2).mean().sqrt() / a.pow(2).mean().sqrt())))
        a, b = outs["f32"], outs["kv16"]
        print("   kv16 vs f32: max err / max = %.3e, rms err / rms = %.3e" % (
            float((a - b).abs().max() / a.abs().max()), float((a - b).pow(2).mean().sqrt() / a.pow(2).mean().sqrt())))
        a, b = outs["f32"], outs["bf16"]
        print("   bf16 vs f32: max err / max = %.3e, rms err / rms = %.3e" % (
            float((a - b).abs().max() / a.abs().max()), float((a - b).pow(2).mean().sqrt() / a.pow(2).mean().sqrt())))
