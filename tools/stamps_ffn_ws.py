"""Developer tool: where a wave of k_ffn_ws spends its cycles (build with MSSVT_EXTRA_HIPCC_FLAGS=-DMSSVT_STAMPS)."""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mssvt_amd import config, fused, _lib
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = config.build_backbone_from_cfg().to(dev).eval()
blk = net.backbone[0]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 74270
x = torch.randn(n, 128, device=dev)
class SP(object):
    _next_norm1 = net.backbone[1].norm1
fused.FFN_ARITH = "f16x3"
with torch.no_grad():
    for _ in range(3):
        fused._ffn_tail(blk, SP(), x)
torch.cuda.synchronize()
buf = np.zeros(4 * 8 * 16, dtype=np.uint64)
_lib.lib().mssvt_debug_read_ffn_ws_stamps(buf.ctypes.data_as(ctypes.c_void_p))
s = buf.reshape(32, 16).astype(np.float64)
names = ["prologue (weights, A(0), GEMM1(0))", "loop top", "I2: GEMM2(t) | A(t+1), issue t+2", "barrier", "I1: D(t) | GEMM1(t+1), split u -> LDS",
         "barrier", "exit"]
tiles = s[:, 9].mean()
tot = s[:, :7].sum(1).mean()
print("waves", s.shape[0], "tiles per wave", tiles, "cycles per wave", tot, "(100 MHz counter: x24 = shader clocks)")
for k in range(7):
    print("   %-36s %9.0f   %5.1f%%   per tile %7.1f" % (names[k], s[:, k].mean(), 100 * s[:, k].mean() / tot, s[:, k].mean() / (tiles if 0 < k < 6 else 1)))

# ---- launch skew and tail of the grid: start / end of every workgroup on the 100 MHz wall clock (round 4)
if hasattr(_lib.lib(), "mssvt_debug_read_ffn_ws_spans"):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.no_grad():
        e0.record()
        fused._ffn_tail(blk, SP(), x)
        e1.record()
    torch.cuda.synchronize()
    sp = np.zeros(1024 * 2, dtype=np.uint64)
    _lib.lib().mssvt_debug_read_ffn_ws_spans(sp.ctypes.data_as(ctypes.c_void_p))
    grid = min(256, (n + 15) // 16)
    sp = sp.reshape(1024, 2)[:grid].astype(np.float64) / 100.0  # us
    t0 = sp[:, 0].min()
    st, en = sp[:, 0] - t0, sp[:, 1] - t0
    print("grid %d: event time of the launch %.1f us; first start -> last end %.1f us" % (grid, e0.elapsed_time(e1) * 1e3, en.max()))
    pct = lambda v: " ".join("%.1f" % np.percentile(v, p) for p in (0, 10, 50, 90, 100))
    print("   start offsets (us, p0 p10 p50 p90 p100): ", pct(st))
    print("   end   offsets (us):                      ", pct(en))
    print("   workgroup durations (us):                ", pct(en - st))
    tiles = (n + 15) // 16
    per = np.array([len(range(b, tiles, grid)) for b in range(grid)])
    for k in sorted(set(per)):
        sel = per == k
        print("   workgroups with %d tiles: %d, mean duration %.1f us, %.3f us per tile" % (k, sel.sum(), (en - st)[sel].mean(), (en - st)[sel].mean() / k))
    # by XCD (workgroup b runs on XCD b % 8)
    print("   mean duration by XCD:", " ".join("%.1f" % (en - st)[x::8].mean() for x in range(8)))
