"""Developer tool: phase stamps of k_attn_q (build with MSSVT_EXTRA_HIPCC_FLAGS=-DMSSVT_STAMPS)."""
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
        p = fused.two_scale_plan(blk, sp)
        xhat = fused.layer_norm(feats, blk.norm1)
        q_ind, nq, _ = fused._query(blk, p)
        od = fused._work_order(blk, p, nq, feats.shape[0])
        qbuf = fused._query_scratch(p, od["row_cap"], blk.ms_attn, dev)
        attn = torch.zeros((p.cap * nq + 1, 128), dtype=torch.float32, device=dev)
        for _ in range(3):
            fused._attention_call(blk, p, od, 128, nq, xhat, qbuf, attn)
        torch.cuda.synchronize()
        buf = np.zeros(64 * 8, dtype=np.uint64)
        _lib.lib().mssvt_debug_read_attn_q_stamps(buf.ctypes.data_as(ctypes.c_void_p))
        s = buf.reshape(64, 8).astype(np.int64)
        d = np.diff(s[:, :5], axis=1)
        print("block", bi, "rows", int(od["n_rows"].item()), "mean cycles: num_rows %d, staging %d, rows+barrier %d, tiles %d; span of starts %d" % (
            d[:, 0].mean(), d[:, 1].mean(), d[:, 2].mean(), d[:, 3].mean(), s[:, 0].max() - s[:, 0].min()))
