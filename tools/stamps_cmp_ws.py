"""per-phase shader clocks of k_cmp_ws (library built with -DCW_STAMPS), 160k-point frame"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mssvt_amd import _lib, config, synthetic
torch.manual_seed(0)
net = config.build_backbone_from_cfg().to("cuda").eval()
pts = synthetic.make_batch_points(160000, 1, 0)
vc, _, _ = synthetic.voxelize_numpy(pts)
feats = torch.randn(vc.shape[0], 128, generator=torch.Generator().manual_seed(0)).cuda()
d = dict(voxel_features=feats, voxel_coords=torch.from_numpy(vc).cuda(), batch_size=1)
with torch.no_grad():
    for _ in range(3):
        net(dict(d))
torch.cuda.synchronize()
L = _lib.lib()
buf = (ctypes.c_ulonglong * (8 * 8 * 16))()
L.mssvt_debug_cmp_ws_stamps(buf)
a = np.array(buf, dtype=np.float64).reshape(8, 8, 16)
names = ["prologue", "first max", "Q", "wait0", "S1", "wait1", "S2", "wait2", "KV+score", "scan", "atomics", "tail max", "wait3", "O", "tiles", "subtiles"]
for b in range(8):
    w = a[b].mean(axis=0)
    tot = w[:14].sum()
    print("wg %3d: total %.0f clk (%.1f us at 2.1 GHz) tiles %d sub %d | " % (b * 32, tot, tot / 2100, w[14], w[15]) +
          " ".join("%s %.0f" % (n, v) for n, v in zip(names[:14], w[:14])))
w = a.mean(axis=(0, 1))
print("per subtile: " + " ".join("%s %.0f" % (n, w[i] / max(w[15], 1)) for i, n in enumerate(names[:14]) if i in (4, 5, 6, 7, 8, 9, 10)))
print("per tile: " + " ".join("%s %.0f" % (n, w[i] / max(w[14], 1)) for i, n in enumerate(names[:14]) if i in (2, 3, 11, 12, 13)))
