#!/usr/bin/env python3
"""When does each frame of a pipelined run complete?  The bench loop (frames resident, inputs_ready, deferred host waits)
with a timing event behind every frame: completion times relative to the first submission, per stream kind.
python tools/pipe_trace.py [--steps 50] [--kinds priority,cumask]"""
import argparse
import os
import sys
import time

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mssvt_amd import config  # noqa: E402
from mssvt_amd.pipeline import FramePipeline  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--kinds", default="priority,cumask")
a = ap.parse_args()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = config.build_backbone_from_cfg(config.load_yaml(config.DEFAULT_CFG)).to(dev).eval()
frames = [bench.make_inputs(160000, 1, 0, dev, frame=f) for f in range(4)]
for kind in a.kinds.split(","):
    os.environ["MSSVT_PIPE_STREAMS"] = kind
    pipe = FramePipeline(net, depth=4)
    for rep in range(3):
        for i in range(8):
            f = frames[i % 4]
            pipe(dict(voxel_features=f[3], voxel_coords=f[2], batch_size=1), inputs_ready=True)
        pipe.synchronize()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        evs, host = [], []
        t0 = time.perf_counter()
        for i in range(a.steps):
            f = frames[i % 4]
            p = pipe(dict(voxel_features=f[3], voxel_coords=f[2], batch_size=1), inputs_ready=True)
            host.append((time.perf_counter() - t0) * 1e3)
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(p.stream)
            evs.append(ev)
        pipe.synchronize()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
        done = [e0.elapsed_time(ev) for ev in evs]
        print("%s rep %d: %d steps in %.2f ms wall (%.0f frames/s); last frame done at %.2f ms" % (kind, rep, a.steps, wall, a.steps / wall * 1e3, max(done)))
        print("   submitted at (ms):", " ".join("%.2f" % t for t in host[:12]), "...", " ".join("%.2f" % t for t in host[-4:]))
        print("   completed at (ms):", " ".join("%.2f" % t for t in done[:12]), "...", " ".join("%.2f" % t for t in done[-4:]))
    pipe.close()
