"""Host microseconds per frame by section (wall-clock wrappers, C entry points stubbed except the level set-up):
investigation helper for the launch-bound front of a one-scene frame."""
import os, sys, time, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mssvt_amd import config, _lib, fused, mssvt_backbone
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = config.build_backbone_from_cfg().to(dev).eval()
frames = [bench.make_inputs(160000, 1, 0, dev, frame=f) for f in range(4)]
def step(i):
    _, _, vc, feats = frames[i % 4]
    with torch.no_grad():
        return net(dict(voxel_features=feats, voxel_coords=vc, batch_size=1))
for i in range(8): step(i)
torch.cuda.synchronize()
real_call = _lib.call
def fake(name, *a):
    if name in ("mssvt_level_setup_sorted", "mssvt_fill_two"):
        real_call(name, *a)
_lib.call = fake
acc = collections.defaultdict(float); cnt = collections.defaultdict(int)
def wrap(mod, name):
    fn = getattr(mod, name)
    def w(*a, **k):
        t = time.perf_counter(); r = fn(*a, **k); acc[name] += time.perf_counter() - t; cnt[name] += 1; return r
    setattr(mod, name, w)
for n in ("block_forward", "compress_forward", "setup_input_level", "two_scale_plan", "prepare_group", "_attention_call",
          "_ffn_tail", "one_scale_plan", "_work_order", "_interp_table", "_query_scratch", "_attn_buffer", "_norm1", "layer_norm",
          "_compress_finish", "_plan_tables", "_sorted_level", "level_state", "window_partition"):
    wrap(fused, n)
N = 60
t0 = time.perf_counter()
for i in range(N):
    step(i)
    if i % 4 == 3: torch.cuda.synchronize()
tot = time.perf_counter() - t0
print("host per frame %.0f us (includes the periodic synchronize)" % (tot / N * 1e6))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("  %-22s %6.1f us/frame  (%4.1f calls/frame, %5.1f us/call)" % (k, v / N * 1e6, cnt[k] / N, v / cnt[k] * 1e6))
