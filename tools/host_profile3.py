"""Host time of a forward split into: everything, without the kernel launches (the C entry points stubbed out), and
without launches + torch allocations cached (investigation helper)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mssvt_amd import config, _lib
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = config.build_backbone_from_cfg().to(dev).eval()
frames = [bench.make_inputs(160000, 1, 0, dev, frame=f) for f in range(4)]
def step(i):
    _, _, vc, feats = frames[i % 4]
    with torch.no_grad():
        return net(dict(voxel_features=feats, voxel_coords=vc, batch_size=1))
def timeit(label, n=40):
    for i in range(5): step(i)
    torch.cuda.synchronize()
    t = []
    for i in range(n):
        a = time.perf_counter(); step(i); t.append(time.perf_counter() - a)
        if i % 4 == 3: torch.cuda.synchronize()   # keep the queue shallow: no back-pressure in the timings
    t.sort()
    print("%-40s median %4d us  min %4d us" % (label, t[len(t) // 2] * 1e6, t[0] * 1e6))
timeit("full forward (host side)")
real_call = _lib.call
calls = [0]
def fake(name, *a):
    calls[0] += 1
    if name == "mssvt_level_setup_sorted":  # keeps the status / count words the host reads valid
        real_call(name, *a)
_lib.call = fake
import mssvt_amd.fused as fused, mssvt_amd.mssvt_ops as ops
timeit("C entry points stubbed out")
print("entry-point calls per frame:", calls[0] / 45.0)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(40):
    step(i)
    if i % 4 == 3: torch.cuda.synchronize()
pr.disable()
print("per frame under the profiler: %.0f us" % (sum(v[2] for v in pstats.Stats(pr).stats.values()) / 40 * 1e6))
pstats.Stats(pr).sort_stats("tottime").print_stats(45)
