"""aten / runtime op census of one backbone forward (where do fills, copies and syncs come from?)."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mssvt_amd import config  # noqa: E402

torch.manual_seed(0)
dev = torch.device("cuda", 0)
net = config.build_backbone_from_cfg().to(dev).eval()
_, _, vc, feats = bench.make_inputs(160000, 1, 0, dev)


def step():
    with torch.no_grad():
        return net(dict(voxel_features=feats, voxel_coords=vc, batch_size=1))


for _ in range(5):
    step()
torch.cuda.synchronize()
N = 4
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(N):
        step()
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_stack_n=6)
rows = []
for e in ka:
    if e.key.startswith("aten::") or "hip" in e.key.lower() or "Memcpy" in e.key or "Memset" in e.key:
        rows.append((e.count / N, e.key, e.self_cpu_time_total / N, e.self_device_time_total / N,
                     [s for s in e.stack if "mssvt_amd" in s or "bench" in s][:3]))
rows.sort(key=lambda r: -r[3])
for r in rows[:70]:
    print("%6.1f %-34s cpu %7.1f us  gpu %7.1f us  %s" % r)
