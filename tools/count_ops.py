"""Elementwise framework ops of one compact training step, by name and input shapes (torch profiler, host side):
which full-width adds / muls / copies autograd still runs around the hand-written kernels (investigation helper)."""
import os, sys, collections
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from mssvt_amd import config
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = config.build_backbone_from_cfg().to(dev).train()
_, _, vc, feats = bench.make_inputs(160000, 1, 0, dev)
def step():
    for p in net.parameters(): p.grad = None
    out = net(dict(voxel_features=feats, voxel_coords=vc, batch_size=1))["encoded_spconv_tensor"].features
    out.square().mean().backward()
step(); step()
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    step()
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::add", "aten::add_", "aten::mul", "aten::where", "aten::addcmul", "aten::copy_", "aten::clone", "aten::contiguous", "aten::fill_", "aten::zero_"):
        shp = str(e.input_shapes[:2])
        cnt[(e.name, shp)] += 1
for (k, shp), v in sorted(cnt.items(), key=lambda kv: -kv[1])[:40]:
    print(v, k, shp)
