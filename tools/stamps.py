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
buf = np.zeros(32768 * 16, dtype=np.uint64)
_lib.lib().mssvt_debug_read_plan_stamps(buf.ctypes.data_as(ctypes.c_void_p))
s = buf.reshape(32768, 16).astype(np.int64)[:19307, :8]
d = np.diff(s, axis=1)
ok = (d >= 0).all(1) & (d < 10_000_000).all(1)
print("windows with sane stamps", ok.sum())
d = d[ok]
print("mean cycles per phase: columns, K3 + indices, lists out, fps1, out1, fps2, out2:", [int(v) for v in d.mean(0)], "sum", int(d.sum(1).mean()))
tot = d.sum(1)
for lo, hi in ((0, 50), (50, 90), (90, 100)):
    a_, b_ = np.percentile(tot, lo), np.percentile(tot, hi)
    sel = (tot >= a_) & (tot <= b_)
    print("  windows p%d-p%d of total time: phases" % (lo, hi), [int(v) for v in d[sel].mean(0)], "sum", int(tot[sel].mean()))
sp = np.zeros(32768 * 2, dtype=np.uint64)
_lib.lib().mssvt_debug_read_plan_span(sp.ctypes.data_as(ctypes.c_void_p))
sp = sp.reshape(-1, 2).astype(np.int64)
live = (sp[:, 1] > sp[:, 0]) & (sp[:, 0] > 0)
t1 = sp[live, 1].max()
live &= sp[:, 0] > t1 - 400_000  # the last launch only
t0 = sp[live, 0].min()
st, en = (sp[live, 0] - t0), (sp[live, 1] - t0)
dur = en - st
print("windows", live.sum(), "kernel span (ticks)", en.max(), "dur mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f" % (
    dur.mean(), np.percentile(dur, 50), np.percentile(dur, 90), np.percentile(dur, 99), dur.max()))
print("start p10 %.0f p50 %.0f p90 %.0f max %.0f" % (np.percentile(st, 10), np.percentile(st, 50), np.percentile(st, 90), st.max()))
print("sum dur / span = avg concurrency %.0f waves" % (dur.sum() / en.max()))
# concurrency over time (20 bins)
edges = np.linspace(0, en.max(), 21)
conc = [(np.minimum(en, b) - np.maximum(st, a)).clip(min=0).sum() / (b - a) for a, b in zip(edges[:-1], edges[1:])]
print("concurrency per 5% of the span:", [int(c) for c in conc])
order = np.argsort(-en)[:8]
print("last finishers: start, dur", [(int(st[i]), int(dur[i])) for i in order])
