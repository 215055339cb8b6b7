"""Developer tool: phase stamps (build with MSSVT_EXTRA_HIPCC_FLAGS=-DMSSVT_STAMPS)."""
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
with torch.no_grad():
    for _ in range(3):
        net(dict(voxel_features=feats, voxel_coords=vc, batch_size=1))
torch.cuda.synchronize()
buf = np.zeros(64 * 16, dtype=np.uint64)
_lib.lib().mssvt_debug_read_plan_stamps(buf.ctypes.data_as(ctypes.c_void_p))
s = buf.reshape(64, 16).astype(np.int64)
print("phases: start, init, cols, phase1, K3 end, win1meta, fps1, out1, fps2, out2   (100 cycles)")
for w in range(0, 64, 4):
    r_ = s[w]
    nz = int((r_ != 0).sum())
    print("win", w, [int(v - r_[0]) // 100 for v in r_[:nz]])
sp = np.zeros(32768 * 2, dtype=np.uint64)
_lib.lib().mssvt_debug_read_plan_span(sp.ctypes.data_as(ctypes.c_void_p))
sp = sp.reshape(-1, 2).astype(np.int64)
live = sp[:, 1] > 0
t0 = sp[live, 0].min()
st, en = (sp[live, 0] - t0) / 100.0, (sp[live, 1] - t0) / 100.0
dur = en - st
print("windows", live.sum(), "kernel span (100 cyc)", en.max(), "dur mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f" % (
    dur.mean(), np.percentile(dur, 50), np.percentile(dur, 90), np.percentile(dur, 99), dur.max()))
print("start p50 %.0f p90 %.0f max %.0f" % (np.percentile(st, 50), np.percentile(st, 90), st.max()))
order = np.argsort(-en)[:8]
print("last finishers: start, dur", [(int(st[i]), int(dur[i])) for i in order])
print("sum dur / span = avg concurrency", dur.sum() / en.max())
