"""Frame rate with the index work on a side stream (async_index) against the single-stream forward; outputs compared."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mssvt_amd import config  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
bf16 = len(sys.argv) > 2 and sys.argv[2] == "bf16"
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = config.build_backbone_from_cfg().to(dev).eval()
if bf16:
    net.set_attn_dtype("bf16")
_, _, vc, feats = bench.make_inputs(160000, batch, 0, dev)
outs = {}
for mode in ("sync", "async within frame", "async across frames"):
    net.async_index = mode != "sync"
    net.async_inputs_resident = mode == "async across frames"
    with torch.no_grad():
        for _ in range(10):
            out = net(dict(voxel_features=feats, voxel_coords=vc, batch_size=batch))["encoded_spconv_tensor"]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 100 if batch == 1 else 30
        for _ in range(n):
            out = net(dict(voxel_features=feats, voxel_coords=vc, batch_size=batch))["encoded_spconv_tensor"]
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
    outs[mode] = (out.features.clone(), out.indices.clone())
    print("%-22s %.3f ms / step, %.0f frames/s" % (mode, ms, batch / ms * 1e3))
for mode in list(outs)[1:]:
    print(mode, "features equal:", torch.equal(outs[mode][0], outs["sync"][0]), "indices equal:", torch.equal(outs[mode][1], outs["sync"][1]))
