"""The roofline section of bench.py once (it launches the timing-only ceiling kernels of csrc/ceiling.hip 20 x each beside the
kernels they stand for): run under `rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU ...` to check that a ceiling
launch issues the instruction counts of its product kernel (tools/pmc_ceiling.sh)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mssvt_amd import config, fused  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = config.build_backbone_from_cfg().to(dev).eval()
_, _, vc, feats = bench.make_inputs(160000, 1, 0, dev)
r = fused.roofline(net, vc, feats, 1, bench.event_time_ms, 8000.0)
print("ceiling us", r.get("ceiling_us"), [o.get("ceiling_us_window_launch") for o in r["other_kernels"]])
