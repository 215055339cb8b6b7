"""Time the plan kernels on the bench scene (investigation helper)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mssvt_amd import config, fused, mssvt_ops
from mssvt_amd.mssvt_utils import SparseTensor

dev = torch.device("cuda", 0)
net = config.build_backbone_from_cfg().to(dev).eval()
_, _, vc, feats = bench.make_inputs(160000, 1, 0, dev)

def mk():
    return SparseTensor(features=feats, indices=vc.int().contiguous(), spatial_shape=net.grid_size,
                        voxel_size=net.voxel_size, point_cloud_range=net.point_cloud_range, batch_size=1,
                        hash_size=net.hash_size)

def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

sp = mk()
blk, cblk = net.backbone[0], net.backbone[4]
print("SparseTensor (K1)      us", t(mk))
print("window_partition_device", t(lambda: mssvt_ops.window_partition_device(blk.win1_size, 90000, 1, net.hash_size, [156,156,6], sp.indices)))
def p2():
    sp._level = None
    fused.two_scale_plan(blk, sp)
print("two_scale_plan total   us", t(p2))
def p1():
    sp._level = None
    fused.one_scale_plan(cblk, sp)
print("one_scale_plan total   us", t(p1))
