"""Developer tool: the hash-overflow scene on DIRTY allocator memory (freed blocks full of garbage), per feature toggle."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mssvt_amd import config, fused, synthetic
from mssvt_amd._lib import MssvtHipError
dev = torch.device("cuda", 0)
for name in sys.argv[1:]:
    k, v = name.split("=")
    setattr(fused, k, v == "1")
torch.manual_seed(0)
net = config.build_backbone_from_cfg().to(dev).eval()
net.hash_size = 1000
pts = synthetic.make_batch_points(20000, 1, 3)
vc, _, _ = synthetic.voxelize_numpy(pts)
feats = torch.randn(vc.shape[0], 128, device=dev)
vct = torch.from_numpy(vc).to(dev)
for rep in range(3):
    junk = [torch.full((64 << 20,), 0x7f7f7f7f, dtype=torch.int32, device=dev) for _ in range(8)]  # 2 GiB of garbage
    del junk
    try:
        with torch.no_grad():
            net(dict(voxel_features=feats, voxel_coords=vct, batch_size=1))
        print("no error raised")
    except MssvtHipError as e:
        print("raised:", str(e)[:60])
    torch.cuda.synchronize()
print("done", sys.argv[1:])
