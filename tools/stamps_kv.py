"""Developer tool: where a wave of the persistent k_attn_kv spends its cycles (build with MSSVT_EXTRA_HIPCC_FLAGS=-DMSSVT_STAMPS).
Runs ONE Block attention of the bench frame per query pattern."""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MSSVT_ATTN_KV", "1")
import bench
from mssvt_amd import config, fused, _lib
torch.manual_seed(0)
dev = torch.device("cuda", 0)
net = config.build_backbone_from_cfg().to(dev).eval()
_, _, vc, feats = bench.make_inputs(160000, 1, 0, dev)
for nblk in (1, 2):
    sub = torch.nn.ModuleList(list(net.backbone[:nblk]))
    with torch.no_grad():
        for _ in range(2):
            kw = dict(features=feats, indices=vc.int().contiguous(), spatial_shape=net.grid_size, voxel_size=net.voxel_size,
                      point_cloud_range=net.point_cloud_range, batch_size=1, hash_size=net.hash_size, gather_dict=None)
            sp = fused.setup_input_level(net.backbone, kw, True)
            for i, b in enumerate(sub):
                sp._next_norm1 = net.backbone[i + 1].norm1
                sp._plan_group = list(net.backbone[:4])
                sp._next_compress = net.backbone[4]
                sp = b(sp)
    torch.cuda.synchronize()
    buf = np.zeros(8192 * 12, dtype=np.uint64)
    _lib.lib().mssvt_debug_read_attn_kv_stamps(buf.ctypes.data_as(ctypes.c_void_p))
    s = buf.reshape(8192, 12).astype(np.int64)
    s = s[s[:, 3] > 0]
    t0 = s[:, 0].min()
    print("block %d (pattern %d): waves with work %d, windows/wave mean %.2f" % (nblk - 1, net.backbone[nblk - 1].cbs_pattern, len(s), s[:, 3].mean()))
    print("  entry spread p50 %d max %d | prologue (entry -> loop) mean %d | loop mean %d | whole wave mean %d max %d, last exit %d" % (
        np.percentile(s[:, 0] - t0, 50), (s[:, 0] - t0).max(), (s[:, 1] - s[:, 0]).mean(), (s[:, 2] - s[:, 1]).mean(),
        (s[:, 2] - s[:, 0]).mean(), (s[:, 2] - s[:, 0]).max(), (s[:, 2] - t0).max()))
    per = s[:, 4:9].sum(0) / s[:, 3].sum()
    print("  cycles per window: token build %d, prefetch issue %d, tile->LDS %d, scores+softmax %d, PV+store %d | passes/window %.2f" % (
        per[0], per[1], per[2], per[3], per[4], s[:, 9].sum() / s[:, 3].sum()))
