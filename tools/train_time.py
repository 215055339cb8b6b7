"""Time a training step (forward + backward + SGD) of the backbone on the bench scene, compact path vs operator
path, with peak memory (investigation helper; usage: python tools/train_time.py [batch] [points])."""
import os
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # as bench.py (mssvt_amd.use_device_kernargs)
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mssvt_amd import config, fused  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
points = int(sys.argv[2]) if len(sys.argv) > 2 else 160000
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = config.build_backbone_from_cfg().to(dev).train()
_, _, vc, feats = bench.make_inputs(points, batch, 0, dev)
opt = torch.optim.SGD(net.parameters(), lr=1e-4)


def step():
    opt.zero_grad(set_to_none=True)
    out = net(dict(voxel_features=feats, voxel_coords=vc, batch_size=batch))["encoded_spconv_tensor"].features
    out.square().mean().backward()
    opt.step()


for compact in ((True,) if os.environ.get("MSSVT_TRAIN_ONLY_COMPACT") else (True, False)):
    fused.TRAIN_COMPACT = compact
    torch.cuda.reset_peak_memory_stats()
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    print("%s path: %.1f ms / training step, peak memory %.2f GB" % (
        "compact" if compact else "operator", (time.perf_counter() - t0) / n * 1e3, torch.cuda.max_memory_allocated() / 2**30))
