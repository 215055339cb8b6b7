"""Which instruction mix could the FFN tail run on?  csrc/ceiling.hip::k_ceiling_ffn_mix -- the launch's real bytes (the
bench frame's tables) with the matrix-instruction shape and the vector-instruction count as parameters -- timed with HIP
events beside k_ffn_ws and the round-5 ceiling kernel.  Run it under `rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA` to read
each variant's real vector-instruction count (tools/pmc_ceiling_mix.sh).   python tools/ceiling_mix.py [--reps 20]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mssvt_amd import _lib, config, fused  # noqa: E402
from mssvt_amd.mssvt_utils import SparseTensor  # noqa: E402

reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 20
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = config.build_backbone_from_cfg().to(dev).eval()
_, _, vc, feats = bench.make_inputs(160000, 1, 0, dev)
blk = net.backbone[0]
_i, _P = ctypes.c_int, fused._P
with torch.no_grad():
    sp = SparseTensor(features=feats, indices=vc.int().contiguous(), spatial_shape=net.grid_size, voxel_size=net.voxel_size,
                      point_cloud_range=net.point_cloud_range, batch_size=1, hash_size=net.hash_size)
    sp._plan_group = [b for b in net.backbone if b.plan_key() == blk.plan_key()]
    p = fused.two_scale_plan(blk, sp)
    x_in = sp.features.contiguous()
    C = x_in.shape[1]
    q_ind, nq, _ = fused._query(blk, p)
    vs3, mn3 = fused._f3(sp.voxel_size), fused._f3(sp.point_cloud_range[0:3])
    tab = fused._interp_table(blk, sp, p, q_ind, nq, p.ind_win1, blk.max_num_win1, p.owner_win1, 1, vs3, mn3)
    abuf = fused._attn_buffer(p, nq, C, dev)
    abuf.normal_()
    sp._next_norm1 = net.backbone[1].norm1
    frag = fused._ffn_f16_weights(fused._ffn_refs(blk))
    yc, ync = torch.empty_like(x_in), torch.empty_like(x_in)
    N = x_in.shape[0]
    ms_ws = bench.event_time_ms(lambda: fused._ffn_tail(blk, sp, None, x_in, None, table=(tab, abuf)), reps)
    ms_c5 = bench.event_time_ms(lambda: _lib.call("mssvt_ceiling_ffn_ws", _i(N), _P(x_in), _P(tab[0]), _P(tab[1]), _P(abuf), _P(frag),
                                                  _P(yc), _P(ync), _lib.stream()), reps)
    print("rows %d   k_ffn_ws %.1f us   k_ceiling_ffn_ws (round 5) %.1f us" % (N, ms_ws * 1e3, ms_c5 * 1e3))
    fill = [0, 72, 144, 216, 288, 400]
    names = {0: "no matrix instructions, 16-row tiles", 1: "48 x 16x16x32 per wave and 16-row tile",
             2: "48 x 32x32x16 per wave and 32-row tile"}
    for mode in (0, 1, 2):
        for k, nv in enumerate(fill):
            ms = bench.event_time_ms(lambda: _lib.call("mssvt_ceiling_ffn_mix", _i(100 * mode + k), _i(N), _P(x_in), _P(tab[0]), _P(tab[1]),
                                                       _P(abuf), _P(frag), _P(yc), _P(ync), _lib.stream()), reps)
            per16 = nv if mode != 2 else nv / 2.0
            print("mix mode %d (%s) filler %3d per wave-tile (%5.1f per 16 rows): %.1f us" % (mode, names[mode], nv, per16, ms * 1e3))
