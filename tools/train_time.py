"""Developer tool: one training step (forward + backward + SGD) of the backbone on a 160k-point scene --
autograd on means the differentiable operator path (K6 / K11 backward kernels); prints ms per step and peak memory."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mssvt_amd import config
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = config.build_backbone_from_cfg().to(dev).train()
_, _, vc, feats = bench.make_inputs(160000, 1, 0, dev)
feats = feats.clone().requires_grad_(True)
opt = torch.optim.SGD(net.parameters(), lr=1e-4)
def step():
    opt.zero_grad(set_to_none=True)
    out = net(dict(voxel_features=feats, voxel_coords=vc, batch_size=1))["encoded_spconv_tensor"].features
    loss = out.square().mean()
    loss.backward()
    opt.step()
    return float(loss)
for _ in range(2): step()
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(5): l=step()
torch.cuda.synchronize(); print("train step ms", (time.perf_counter()-t)/5*1e3, "loss", l, "peak GB", torch.cuda.max_memory_allocated()/2**30)
