"""Host cost of one forward by function (cumulative), GPU queue kept deep so that nothing waits (investigation helper)."""
import os, sys, time, cProfile, pstats
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mssvt_amd import config
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = config.build_backbone_from_cfg().to(dev).eval()
frames = [bench.make_inputs(160000, 1, 0, dev, frame=f) for f in range(4)]
def step(i):
    _, _, vc, feats = frames[i % 4]
    with torch.no_grad():
        return net(dict(voxel_features=feats, voxel_coords=vc, batch_size=1))
for i in range(10): step(i)
torch.cuda.synchronize()
# pure host time: issue 30 frames back to back and time only the calls
t = []
for i in range(30):
    a = time.perf_counter(); step(i); t.append(time.perf_counter() - a)
torch.cuda.synchronize()
print("host per frame (first 5 calls, queue empty -> no back-pressure): %s us" % [round(x * 1e6) for x in t[:5]])
print("median over 30: %d us" % (sorted(t)[15] * 1e6))
pr = cProfile.Profile(); pr.enable()
for i in range(20): step(i)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(32)
st = pstats.Stats(pr)
st.sort_stats("tottime").print_callers("__getattr__")
st.print_callers("torch.empty")
st.print_callers("decorate_context")
