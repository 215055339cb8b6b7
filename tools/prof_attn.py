"""Launch the fused attention kernel a few times on the bench inputs (for rocprofv3 --pmc runs)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mssvt_amd import config, fused  # noqa: E402

torch.manual_seed(0)
dev = torch.device("cuda", 0)
net = config.build_backbone_from_cfg().to(dev).eval()
points = int(sys.argv[1]) if len(sys.argv) > 1 else 160000
_, _, vc, feats = bench.make_inputs(points, 1, 0, dev)
r = fused.roofline(net, vc, feats, 1, bench.event_time_ms, bench.HBM_PEAK_GBS)
print(r)
