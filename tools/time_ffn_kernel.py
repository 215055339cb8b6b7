"""Time the FFN tail launches of Block 0 (and the CompressBlock) alone at bench size, both arithmetics."""
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
blk = net.backbone[0]
for n in (7400 * batch, 74270 * batch):
    x = torch.randn(n, 128, device=dev)

    class SP(object):
        _next_norm1 = net.backbone[1].norm1
    with torch.no_grad():
        outs = {}
        for arith in ("f32", "f16x3"):
            fused.FFN_ARITH = arith
            for _ in range(3):
                y = fused._ffn_tail(blk, SP(), x)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                y = fused._ffn_tail(blk, SP(), x)
            e1.record()
            torch.cuda.synchronize()
            outs[arith] = y
            print("rows %d %s: %.1f us" % (n, arith, e0.elapsed_time(e1) / 20 * 1e3))
        print("   max diff %.3e" % float((outs["f32"] - outs["f16x3"]).abs().max()))
