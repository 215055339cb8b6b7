"""cProfile of the compact training step (host side): where the Python / launch time goes."""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mssvt_amd import config  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = config.build_backbone_from_cfg().to(dev).train()
_, _, vc, feats = bench.make_inputs(160000, 1, 0, dev)
opt = torch.optim.SGD(net.parameters(), lr=1e-4)


def fwd():
    opt.zero_grad(set_to_none=True)
    return net(dict(voxel_features=feats, voxel_coords=vc, batch_size=1))["encoded_spconv_tensor"].features.square().mean()


for _ in range(3):
    fwd().backward(); opt.step()
torch.cuda.synchronize()
# host time of forward / backward / optimizer without waiting for the GPU
tf = tb = to = 0.0
for _ in range(5):
    t0 = time.perf_counter(); loss = fwd(); t1 = time.perf_counter(); loss.backward(); t2 = time.perf_counter(); opt.step(); t3 = time.perf_counter()
    tf += t1 - t0; tb += t2 - t1; to += t3 - t2
    torch.cuda.synchronize()
print("host ms per step: forward %.2f  backward %.2f  optimizer %.2f" % (tf / 5 * 1e3, tb / 5 * 1e3, to / 5 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    fwd().backward(); opt.step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
