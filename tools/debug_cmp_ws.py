"""developer check of csrc/compress_ws.hip against float64 and the three-launch form (run on the GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from _pytest.monkeypatch import MonkeyPatch
from tests import test_compress_ws_gpu as T

mp = MonkeyPatch()
pts, B, ws, ns = int(os.environ.get("PTS", 20000)), int(os.environ.get("B", 1)), (1, 1, int(os.environ.get("WZ", 32))), int(os.environ.get("NS", 32))
blk = T._compress_block(ws=ws, ns=ns)
sp = T._sp(pts, B, 3)
new, p, xhat, calls = T._attention_only(blk, sp, True, mp)
nw = int(p.num_wins.item())
want = T._reference_f64(blk, sp, p, xhat)
sp2 = T._sp(pts, B, 3)
old, p2, _, _ = T._attention_only(blk, sp2, False, mp)
print("calls", [c for c in calls if "compress" in c], "nw", nw, "N", xhat.shape[0])
eo = (old[:nw].double() - want).abs()
en = (new[:nw].double() - want).abs()
print("old vs f64 max", float(eo.max()), " new vs f64 max", float(en.max()), "scale", float(want.abs().max()))
bad = en.max(dim=1).values > 1e-3
print("bad windows", int(bad.sum()), "of", nw)
cnt = p.win_cnt[:nw]
for c in sorted(set(cnt.tolist()))[:12]:
    m = cnt == c
    print("  cnt", c, "windows", int(m.sum()), "bad", int((bad & m).sum()))
idx = torch.arange(nw, device=bad.device)
print("bad by window %16:", [int((bad & (idx % 16 == k)).sum()) for k in range(16)])
print("bad by head (channel/16):", [int((en[:, 16 * h:16 * h + 16].max(dim=1).values > 1e-3).sum()) for h in range(8)])
print("bad by channel%16:", [int((en[:, k::16].max(dim=1).values > 1e-3).sum()) for k in range(16)])
if bad.any():
    w = int(idx[bad][0])
    print("first bad window", w, "cnt", int(cnt[w]), "got", new[w, :8].tolist(), "want", want[w, :8].tolist())
    tile = w // 16
    print("tile", tile, "bad flags in tile", bad[tile * 16:tile * 16 + 16].int().tolist(), "cnts", cnt[tile * 16:tile * 16 + 16].tolist())

# intermediate values (library built with -DCW_DEBUG)
import ctypes
from mssvt_amd import _lib
L = _lib.lib()
if hasattr(L, "mssvt_debug_cmp_ws"):
    C = 128
    N = xhat.shape[0]
    qtok = torch.zeros(nw + 16, C, device="cuda"); qpd = torch.zeros(nw + 16, C, device="cuda"); scd = torch.zeros(N, 8, device="cuda")
    L.mssvt_debug_cmp_ws(ctypes.c_void_p(qtok.data_ptr()), ctypes.c_void_p(qpd.data_ptr()), ctypes.c_void_p(scd.data_ptr()))
    sp3 = T._sp(pts, B, 3)
    new3, p3, xhat3, _ = T._attention_only(blk, sp3, True, mp)
    torch.cuda.synchronize()
    ma = blk.ms_attn
    d = lambda t: t.detach().double()
    k_ind = p3.k_ind[:nw].long(); valid = k_ind >= 0
    rows = p3.win_vstart[:nw].long()[:, None] + k_ind.clamp(min=0)
    x = d(xhat3)[rows] * valid[..., None]
    q_tok = x.max(dim=1).values
    print("q_tok err", float((qtok[:nw].double() - q_tok).abs().max()))
    q = (q_tok @ d(ma.to_qs[0].weight).T + d(ma.to_qs[0].bias)) * ma.scale * 1.4426950408889634
    print("q' err", float((qpd[:nw].double() - q).abs().max()), "scale", float(q.abs().max()))
    # scores per voxel row
    vs = torch.tensor(sp3.voxel_size, dtype=torch.float32, device="cuda"); mn = torch.tensor(sp3.point_cloud_range[:3], dtype=torch.float32, device="cuda")
    wsz = torch.tensor(p3.win_size_m, dtype=torch.float32, device="cuda")
    pw = p3.pair_win[:N].long()
    vxyz = sp3.indices[:, [3, 2, 1]].float(); wxyz = p3.win_ind[:nw][:, [3, 2, 1]].float()
    vc = (vxyz + 0.5) * vs + mn; wc = ((wxyz + 0.5) * wsz + mn)[pw.clamp(min=0)]
    geo = torch.cat([vc - wc, wc], dim=-1).double()
    W1, b1 = d(blk.pos_proj[0].weight).reshape(C, 6), d(blk.pos_proj[0].bias)
    W2, b2 = d(blk.pos_proj[2].weight).reshape(C, C), d(blk.pos_proj[2].bias)
    ktok = d(xhat3) + torch.relu(torch.relu(geo @ W1.T + b1) @ W2.T + b2)
    K = ktok @ d(ma.to_kvs[0].weight)[:C].T + d(ma.to_kvs[0].bias)[:C]
    s = (q[pw.clamp(min=0)] * K).reshape(N, 8, 16).sum(-1)
    ok = pw >= 0
    print("score err", float((scd.double() - s)[ok].abs().max()), "scale", float(s[ok].abs().max()))
