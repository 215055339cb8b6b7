"""Developer tool: 3000 forwards over four scenes of different sizes -- allocator footprint must stay flat and
the output of a scene bit-identical every time it comes round."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mssvt_amd import config
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = config.build_backbone_from_cfg().to(dev).eval()
net.async_inputs_resident = net.async_index  # MSSVT_ASYNC_INDEX=1: the next frame's index work under this frame's features
ins = [bench.make_inputs(p, 1, s, dev)[2:] for p, s in ((160000, 0), (120000, 1), (200000, 2), (60000, 3))]
ref = None
with torch.no_grad():
    for it in range(3000):
        vc, feats = ins[it % 4]
        out = net(dict(voxel_features=feats, voxel_coords=vc, batch_size=1))["encoded_spconv_tensor"].features
        if it % 4 == 0:
            if ref is None: ref = out.clone()
            elif it % 400 == 0: assert torch.equal(ref, out), it
        if it % 500 == 0:
            torch.cuda.synchronize()
            print(it, "alloc GB %.2f reserved GB %.2f" % (torch.cuda.memory_allocated()/2**30, torch.cuda.memory_reserved()/2**30), flush=True)
torch.cuda.synchronize(); print("soak ok")
