"""Time the FFN tail of one Block at bench size: split-fp16 single launch vs the two fp32 launches; max difference."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mssvt_amd import config, fused  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = config.build_backbone_from_cfg().to(dev).eval()
_, _, vc, feats = bench.make_inputs(160000, batch, 0, dev)
with torch.no_grad():
    res = {}
    for arith in ("f32", "f16x3"):
        fused.FFN_ARITH = arith
        out = net(dict(voxel_features=feats, voxel_coords=vc, batch_size=batch))["encoded_spconv_tensor"].features
        res[arith] = out
        for _ in range(3):
            net(dict(voxel_features=feats, voxel_coords=vc, batch_size=batch))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            net(dict(voxel_features=feats, voxel_coords=vc, batch_size=batch))
        e1.record()
        torch.cuda.synchronize()
        print("%s: %.3f ms / frame-batch" % (arith, e0.elapsed_time(e1) / 20))
    a, b = res["f32"], res["f16x3"]
    print("max |f16x3 - f32| = %.3e, max |f32| = %.3e" % (float((a - b).abs().max()), float(a.abs().max())))
