"""A few whole forwards of the bench frame (for rocprofv3 --pmc passes over every kernel of the frame)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mssvt_amd import config  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = config.build_backbone_from_cfg().to(dev).eval()
if len(sys.argv) > 2:
    net.set_attn_dtype(sys.argv[2])
_, _, vc, feats = bench.make_inputs(160000, batch, 0, dev)
with torch.no_grad():
    for _ in range(12):
        net(dict(voxel_features=feats, voxel_coords=vc, batch_size=batch))
torch.cuda.synchronize()
print("done")
