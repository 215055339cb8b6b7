"""Developer tool: random small backbone configurations, fused path against the operator path."""
import os, sys, random
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mssvt_amd import synthetic
from mssvt_amd.config import Config
from mssvt_amd.mssvt_backbone import MixedScaleSparseTransformer

dev = "cuda"
if len(sys.argv) > 2 and sys.argv[1] == "--each":  # one subprocess per seed: a GPU fault only kills that seed
    import subprocess
    first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    for sd in range(first, first + int(sys.argv[2])):
        r = subprocess.run([sys.executable, __file__, "--one", str(sd)], capture_output=True, text=True, timeout=300)
        out = [l for l in r.stdout.splitlines() if l and not l.startswith("failures")]
        print(out[-1] if out else "%d CRASH rc=%d %s" % (sd, r.returncode, r.stderr.strip().splitlines()[-1][:120] if r.stderr.strip() else ""), flush=True)
    sys.exit(0)
seeds = [int(sys.argv[2])] if len(sys.argv) > 2 and sys.argv[1] == "--one" else range(int(sys.argv[1]) if len(sys.argv) > 1 else 24)
bad = 0
for seed in seeds:
    rng = random.Random(seed)
    wide = seed >= 1000  # seeds >= 1000: the wider space (C = 128, K = 64, big windows, pooling compress windows)
    C = rng.choice([32, 64, 128] if wide else [32, 64])
    heads = rng.choice({32: [[2, 2], [1, 1], [1, 3]], 64: [[2, 2], [1, 3], [4, 4], [2, 6]], 128: [[4, 4], [8, 8], [2, 2]]}[C]) \
        if wide else (rng.choice([[2, 2], [1, 3], [4, 4]]) if C == 64 else rng.choice([[2, 2], [1, 1]]))
    if C // sum(heads) not in (8, 16, 32):
        continue
    w1 = rng.choice([[3, 3, 5], [3, 3, 3], [2, 2, 2], [5, 5, 3]] + ([[4, 4, 2], [1, 1, 3], [7, 7, 3]] if wide else []))
    w2 = [w1[i] + rng.choice([2, 4]) for i in range(3)]
    full1 = (w1[0] + (1 - w1[0] % 2)) * (w1[1] + (1 - w1[1] % 2)) * (w1[2] + (1 - w1[2] % 2))
    m1 = rng.choice([full1, max(4, full1 // 3)])
    m2 = rng.choice([w2[0] * w2[1] * w2[2], 40])
    K = rng.choice([8, 16, 32, 64] if wide else [8, 16, 32])
    B = rng.choice([1, 2, 3])
    pts = rng.choice([2000, 8000, 20000])
    blocks = []
    for i in range(rng.choice([1, 2, 3])):
        blocks.append(dict(name="MixedScaleSparseTransformerBlock", channels=[C, 2 * C, C], num_heads=heads,
                           window_size=[w1, w2], max_num_win1=m1, max_num_win2=m2, cbs_mode="odd_even",
                           cbs_pattern=rng.choice([0, 1, 2]), key_num_sample=K,
                           use_feature_interpolation=rng.choice([True, False])))
    cw = rng.choice([[1, 1, 32], [1, 1, 16], [1, 1, 8]] + ([[3, 3, 5], [2, 2, 4], [1, 1, 2]] if wide else []))
    cz = cw[2]
    cfull = (cw[0] + (1 - cw[0] % 2)) * (cw[1] + (1 - cw[1] % 2)) * (cw[2] + (1 - cw[2] % 2))
    blocks.append(dict(name="MixedScaleSparseTransformerCompressBlock", channels=[C, 2 * C, rng.choice([C, C]) if not wide else rng.choice([C, 48])],
                       num_heads=[rng.choice([2, 4])] if not wide else rng.choice([[1], [2], [4], [2, 2]]),
                       window_size=[cw], max_num_win1=cfull if wide else cz))
    if C % sum(blocks[-1]["num_heads"]):
        continue
    torch.manual_seed(seed)
    try:
        net = MixedScaleSparseTransformer(Config.wrap(dict(HASH_SIZE=rng.choice([200003, 30011]) if wide else 200003,
                                                           NUM_OUTPUT_FEATURES=blocks[-1]["channels"][2], PARAMS=blocks)), C,
                                          synthetic.GRID_SIZE, synthetic.VOXEL_SIZE, synthetic.POINT_CLOUD_RANGE).to(dev).eval()
        vc, _, _ = synthetic.voxelize_numpy(synthetic.make_batch_points(pts, B, seed))
        f = torch.randn(vc.shape[0], C, generator=torch.Generator().manual_seed(seed)).to(dev)
        bd = lambda: dict(voxel_features=f, voxel_coords=torch.from_numpy(vc).to(dev), batch_size=B)
        def junk():  # MSSVT_FUZZ_GARBAGE=1: the allocator's free blocks hold garbage (reads of unwritten memory show)
            if os.environ.get("MSSVT_FUZZ_GARBAGE", "0") == "1":
                j = [torch.full((32 << 20,), rng.choice([0x7f7f7f7f, -1, 0x7fc00000]), dtype=torch.int32, device=dev)
                     for _ in range(8)]
                del j
        with torch.no_grad():
            junk()
            a = net.set_impl("fused")(bd())["encoded_spconv_tensor"]
            junk()
            b = net.set_impl("ops")(bd())["encoded_spconv_tensor"]
        ok = torch.equal(a.indices, b.indices)
        err = float(((a.features - b.features).abs() / b.features.abs().clamp(min=1.0)).max()) if ok else float("nan")
        status = "ok" if ok and err < 1e-3 else "MISMATCH"
    except Exception as e:  # noqa: BLE001
        status, err = "ERROR %s: %s" % (type(e).__name__, str(e)[:100]), float("nan")
    bad += status != "ok"
    print(seed, status, flush=True) if False else None
    print(seed, status, "err %.2e" % err, dict(C=C, heads=heads, w1=w1, w2=w2, m1=m1, m2=m2, K=K, B=B, pts=pts,
                                             pats=[(b_.get("cbs_pattern"), b_.get("use_feature_interpolation")) for b_ in blocks[:-1]], cmp=(cw, blocks[-1]["num_heads"], blocks[-1]["channels"][2])))
print("failures:", bad)
