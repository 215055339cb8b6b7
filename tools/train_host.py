"""Host side of the compact training step: time to ISSUE a step (no sync inside), and a cProfile of five steps
(investigation helper; usage: python tools/train_host.py)."""
import cProfile
import os
import pstats
import sys
import time

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
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
    out = net(dict(voxel_features=feats, voxel_coords=vc, batch_size=1))["encoded_spconv_tensor"].features
    return out.square().mean()


for _ in range(3):
    fwd().backward()
    opt.step()
torch.cuda.synchronize()
tf = tb = 0.0
n = 5
for _ in range(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss = fwd()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    loss.backward()
    opt.step()
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    tf += t1 - t0
    tb += t3 - t2
    print("forward issue %.2f ms (+%.2f to drain), backward issue %.2f ms (+%.2f to drain)" % (
        (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3))
pr = cProfile.Profile()
for _ in range(n):
    torch.cuda.synchronize()
    pr.enable()
    loss = fwd()
    pr.disable()
    loss.backward()
    opt.step()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(40)
st.sort_stats("cumtime").print_stats("mssvt_amd", 45)
