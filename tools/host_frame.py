"""Host time of a frame: the whole-frame C call (mssvt_amd/frame.py) against the Python-driven fused path.
On a tiny scene (2k points) the GPU is never the limiter, so the wall time per frame of a free-running loop IS the host's
own time per frame; the 160k-point rows show what the benchmark frame gets."""
import os, sys, time
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # as bench.py (mssvt_amd.use_device_kernargs)
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mssvt_amd import config, frame
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = config.build_backbone_from_cfg().to(dev).eval()
for points in (2000, 160000):
    frames = [bench.make_inputs(points, 1, 0, dev, frame=f) for f in range(4)]
    for on in (True, False, True, False):
        frame.ENABLED = on
        def step(i):
            _, _, vc, feats = frames[i % 4]
            with torch.no_grad():
                return net(dict(voxel_features=feats, voxel_coords=vc, batch_size=1))
        for i in range(20): step(i)
        torch.cuda.synchronize()
        N = 300
        t0 = time.perf_counter()
        for i in range(N): step(i)
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print("%6d points, frame call %-5s: %7.1f us / frame issued, %7.1f us / frame completed" % (points, on, t_host / N * 1e6, t_all / N * 1e6))
