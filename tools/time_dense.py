"""Developer tool: dense()/HeightCompression input of the bench frame, gather kernel vs the torch formulation."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mssvt_amd import config
from mssvt_amd.mssvt_utils import scatter_nd
torch.manual_seed(0)
dev = torch.device("cuda", 0)
net = config.build_backbone_from_cfg().to(dev).eval()
_, _, vc, feats = bench.make_inputs(160000, 1, 0, dev)
with torch.no_grad():
    sp = net(dict(voxel_features=feats, voxel_coords=vc, batch_size=1))["encoded_spconv_tensor"]
print("output voxels", tuple(sp.features.shape), "grid", sp.spatial_shape)
def ref():
    zyx = list(sp.spatial_shape[::-1])
    return scatter_nd(sp.indices.long(), sp.features, [sp.batch_size] + zyx + [sp.features.shape[1]]).permute(0, 4, 1, 2, 3).contiguous()
assert torch.equal(sp.dense(), ref())
print("kernel  %.1f us" % (1e3 * bench.event_time_ms(sp.dense, 50)))
print("torch   %.1f us" % (1e3 * bench.event_time_ms(ref, 50)))
