"""Developer tool: invalid voxel coordinates (outside the grid, negative, duplicated, batch index out of range) on a
garbage-filled allocator: the forward must either raise or return -- never fault."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mssvt_amd import config, synthetic
from mssvt_amd._lib import MssvtHipError
dev = torch.device("cuda", 0)
impl = sys.argv[1] if len(sys.argv) > 1 else "fused"
torch.manual_seed(0)
net = config.build_backbone_from_cfg().to(dev).eval().set_impl(impl)
X, Y, Z = synthetic.GRID_SIZE
for case in ("oob_x", "neg_z", "dup", "b_high", "b_neg", "all"):
    vc, _, _ = synthetic.voxelize_numpy(synthetic.make_batch_points(20000, 2, 5))
    vc = vc.copy()
    n = vc.shape[0]
    rows = list(range(100, n, n // 50))
    for j, r in enumerate(rows):
        kind = case if case != "all" else ("oob_x", "neg_z", "dup", "b_high", "b_neg")[j % 5]
        if kind == "oob_x": vc[r, 3] = X + 5
        if kind == "neg_z": vc[r, 1] = -1
        if kind == "dup": vc[r, 1:] = vc[r - 1, 1:]
        if kind == "b_high": vc[r, 0] = 7
        if kind == "b_neg": vc[r, 0] = -1
    feats, vct = torch.randn(n, 128, device=dev), torch.from_numpy(vc).to(dev)
    junk = [torch.full((32 << 20,), 0x7f7f7f7f, dtype=torch.int32, device=dev) for _ in range(16)]
    del junk
    try:
        with torch.no_grad():
            out = net(dict(voxel_features=feats, voxel_coords=vct, batch_size=2))["encoded_spconv_tensor"]
        torch.cuda.synchronize()
        print(case, "returned", tuple(out.features.shape), "finite:", bool(torch.isfinite(out.features).all()), flush=True)
    except (MssvtHipError, AssertionError, RuntimeError) as e:
        torch.cuda.synchronize()
        print(case, "raised", type(e).__name__, str(e)[:80], flush=True)
print("done")
