"""Developer tool: where does the host time of one forward go (tiny scene => GPU time negligible)."""
import cProfile, pstats, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mssvt_amd import config
torch.manual_seed(0)
dev = torch.device("cuda", 0)
net = config.build_backbone_from_cfg().to(dev).eval()
_, _, vc, feats = bench.make_inputs(2000, 1, 0, dev)
def step():
    with torch.no_grad():
        return net(dict(voxel_features=feats, voxel_coords=vc, batch_size=1))
for _ in range(20): step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(200): step()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
