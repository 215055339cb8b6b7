"""Developer tool: per-wave phase stamps of the split FFN kernels (build with MSSVT_EXTRA_HIPCC_FLAGS=-DMSSVT_STAMPS)."""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mssvt_amd import config, fused, _lib
torch.manual_seed(0)
dev = torch.device("cuda", 0)
net = config.build_backbone_from_cfg().to(dev).eval()
_, _, vc, feats = bench.make_inputs(160000, 1, 0, dev)
r = fused.roofline(net, vc, feats, 1, bench.event_time_ms, bench.HBM_PEAK_GBS)
torch.cuda.synchronize()
buf = np.zeros(2 * 4 * 8 * 64, dtype=np.uint64)
_lib.lib().mssvt_debug_read_stamps(buf.ctypes.data_as(ctypes.c_void_p))
s = buf.reshape(2, 4, 8, 64).astype(np.int64)
for kid in range(1):
    for b in range(2):
        base = s[kid, b, :, 0].min()
        for w in range(8):
            r_ = s[kid, b, w]
            nz = int((r_ != 0).sum())
            print("k", kid, "blk", b, "wave", w, [int(v - base) // 100 for v in r_[:nz]])
print(r["second_kernel"]["avg_launch_us"])
