"""Time the window-plan kernel ALONE on the bench scene (investigation helper; safe for timing-only ablation builds:
nothing consumes its outputs here).  usage: python tools/time_plan.py [batch]"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mssvt_amd import config, fused, _lib

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda", 0)
net = config.build_backbone_from_cfg().to(dev).eval()
_, _, vc, feats = bench.make_inputs(160000, batch, 0, dev)
kw = dict(features=feats, indices=vc.int().contiguous(), spatial_shape=net.grid_size, voxel_size=net.voxel_size,
          point_cloud_range=net.point_cloud_range, batch_size=batch, hash_size=net.hash_size, gather_dict=None)
with torch.no_grad():
    sp = fused.setup_input_level(net.backbone, kw, True)
    sp._plan_group = [b for b in net.backbone[:4]]
    sp._next_compress = net.backbone[4]
    p = fused.two_scale_plan(net.backbone[0], sp)
torch.cuda.synchronize()
print("windows", int(p.num_wins.item()), "status", int(sp._level["level_status"].item()))


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print("mssvt_window_plan_two alone: %.1f us" % t(lambda: _lib.call("mssvt_window_plan_two", *p._plan_args, ctypes.c_int(0), None, None, None, None, None, _lib.stream())))
