"""Developer tool: edge cases (window-count overflow, empty / tiny scenes, voxelizer on non-finite points) on garbage-filled allocator memory."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mssvt_amd import config, synthetic, voxelize
from mssvt_amd._lib import MssvtHipError
dev = torch.device("cuda", 0)
def junk(g=0x7f7f7f7f):
    j = [torch.full((32 << 20,), g, dtype=torch.int32, device=dev) for _ in range(8)]
    del j
torch.manual_seed(0)
# 1. window overflow
net = config.build_backbone_from_cfg().to(dev).eval()
for b in net.backbone: b.max_num_wins = 100
vc, _, _ = synthetic.voxelize_numpy(synthetic.make_batch_points(20000, 2, 5))
feats, vct = torch.randn(vc.shape[0], 128, device=dev), torch.from_numpy(vc).to(dev)
for g in (0x7f7f7f7f, -1):
    junk(g)
    try:
        with torch.no_grad(): net(dict(voxel_features=feats, voxel_coords=vct, batch_size=2))
        print("win overflow: returned")
    except MssvtHipError as e:
        print("win overflow raised:", str(e)[:50])
    torch.cuda.synchronize()
# 2. empty / tiny scenes on garbage
net = config.build_backbone_from_cfg().to(dev).eval()
for n in (0, 1, 7):
    v = torch.zeros((n, 4), dtype=torch.int32)
    if n: v[:, 1] = torch.arange(n) % 32; v[:, 2] = 100 + 3 * torch.arange(n); v[:, 3] = 200
    junk(0x7fc00000)
    with torch.no_grad():
        sp = net(dict(voxel_features=torch.randn(n, 128).to(dev), voxel_coords=v.to(dev), batch_size=1))["encoded_spconv_tensor"]
    torch.cuda.synchronize()
    print("tiny", n, tuple(sp.features.shape), bool(torch.isfinite(sp.features).all()))
# 3. voxelizer with NaN / inf / far points
pts = synthetic.make_batch_points(20000, 2, 1).copy()
pts[10, 1] = np.nan; pts[11, 2] = np.inf; pts[12, 3] = -np.inf; pts[13, 1] = 1e30; pts[14, 0] = 5
junk()
import inspect
print(inspect.signature(voxelize.voxelize))
want_vc, want_inv, kept = synthetic.voxelize_numpy(pts)
vc2, pv = voxelize.voxelize(torch.from_numpy(pts).to(dev), synthetic.POINT_CLOUD_RANGE, synthetic.VOXEL_SIZE, synthetic.GRID_SIZE, 2)
torch.cuda.synchronize()
pvn = pv.cpu().numpy()
print("voxelizer non-finite: coords equal", np.array_equal(vc2.cpu().numpy(), want_vc), "dropped", (pvn[~kept] == -1).all(), "kept map equal", np.array_equal(pvn[kept], want_inv), "rows 10..14:", pvn[10:15], kept[10:15])
