"""Developer tool: where a wave of k_attn_bf16 spends its cycles (build with MSSVT_EXTRA_HIPCC_FLAGS=-DMSSVT_STAMPS)."""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mssvt_amd import config, fused, _lib
from mssvt_amd.mssvt_utils import SparseTensor
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = config.build_backbone_from_cfg().to(dev).eval()
_, _, vc, feats = bench.make_inputs(160000, 1, 0, dev)
with torch.no_grad():
    sp = SparseTensor(features=feats, indices=vc.int().contiguous(), spatial_shape=net.grid_size,
                      voxel_size=net.voxel_size, point_cloud_range=net.point_cloud_range, batch_size=1, hash_size=net.hash_size)
    for bi in (0, 1):
        blk = net.backbone[bi]
        blk.attn_dtype = "bf16"
        p = fused.two_scale_plan(blk, sp)
        xhat = fused.layer_norm(feats, blk.norm1)
        q_ind, nq, _ = fused._query(blk, p)
        od = fused._work_order(blk, p, nq, feats.shape[0])
        attn = torch.zeros((p.cap * nq + 1, 128), dtype=torch.float32, device=dev)
        for _ in range(3):
            fused._attention_call(blk, p, od, 128, nq, xhat, None, attn)
        torch.cuda.synchronize()
        buf = np.zeros(64 * 4 * 8, dtype=np.uint64)
        _lib.lib().mssvt_debug_read_attn_bf_stamps(buf.ctypes.data_as(ctypes.c_void_p))
        s = buf.reshape(-1, 8).astype(np.float64)
        names = ["prologue", "tokens(wait rows)", "issue next", "kv proj", "q tokens(wait xq)", "q..store", "-", "windows"]
        tot = s[:, :6].sum(1)
        print("block", bi, "waves", s.shape[0], "mean cycles per wave", tot.mean(), "windows per wave", s[:, 7].mean())
        for k in range(6):
            print("   %-20s %8.0f cycles/wave  %5.1f%%   per window %7.0f" % (names[k], s[:, k].mean(), 100 * s[:, k].mean() / tot.mean(), s[:, k].mean() / max(s[:, 7].mean(), 1)))
