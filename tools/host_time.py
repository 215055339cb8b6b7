"""How long the host spends per forward (Python + launches) against the GPU time per frame: if the two are close the
frame is launch-bound and faster kernels do not show (investigation helper)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mssvt_amd import config
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = config.build_backbone_from_cfg().to(dev).eval()
_, _, vc, feats = bench.make_inputs(160000, batch, 0, dev)
def step():
    with torch.no_grad():
        return net(dict(voxel_features=feats, voxel_coords=vc, batch_size=batch))
for _ in range(10): step()
torch.cuda.synchronize()
n = 100
host = 0.0
t0 = time.perf_counter()
for _ in range(n):
    a = time.perf_counter(); step(); host += time.perf_counter() - a
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print("batch %d: %.3f ms per frame wall, %.3f ms of it inside the forward call on the host" % (batch, tot / n * 1e3, host / n * 1e3))
# host-only cost: same calls with the GPU far behind?  time the call when the queue is already deep
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
