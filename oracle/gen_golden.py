"""Generate tests/golden/*.npz by running the REFERENCE's own Python (BUILD CONTAINER ONLY).

    python -m oracle.gen_golden               # from /root/repo: everything
    python -m oracle.gen_golden --backbone-only  # tests/golden/backbone.npz alone
    python -m oracle.gen_golden --even-only   # only the cases added after the first set (even windows, empty
                                              # sample, K = 64, enlarged windows, two-level backbone): the
                                              # committed files of the first set stay byte-identical

What is pinned:
  * attention_block.npz / attention_compress.npz -- true reference arithmetic of
    ``MixedScaleAttention.forward`` (mssvt_utils.py:88-157), batch-first with query
    and key masks (Block) and sequence-first nq=1 with a key mask (CompressBlock);
  * query_tables.npz -- ``get_vox_query_table`` (mssvt_backbone.py:73-122) for four
    window configurations (tie order = this CPU run's torch.sort; F7c);
  * block_*.npz / compress_*.npz / backbone*.npz -- the reference's
    ``MixedScaleSparseTransformerBlock/CompressBlock/MixedScaleSparseTransformer``
    ``forward`` executed unmodified with the C oracle underneath (oracle/ref_import.py),
    with every index intermediate recorded.

The fixtures hold data only (inputs, weights, intermediates, outputs).
"""
import os
import sys

import numpy as np
import torch

from mssvt_amd import synthetic
from . import ref_import

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

PC_RANGE = [-9.6, -9.6, -2.0, 9.6, 9.6, 4.0]
VOXEL = [0.32, 0.32, 0.1875]
GRID = [60, 60, 32]
HASH = 4099


def sd_to_np(sd, prefix="sd."):
    return {prefix + k: v.detach().cpu().numpy() for k, v in sd.items()}


def toy_scene(batch_size, pts_per_scene, seed0, C, empty_sample=None):
    pts = synthetic.make_batch_points(pts_per_scene, batch_size, seed0)
    if empty_sample is not None:  # a sample of the batch without a single point
        pts = pts[pts[:, 0] != empty_sample]
    vc, _, _ = synthetic.voxelize_numpy(pts, PC_RANGE, VOXEL, GRID)
    g = torch.Generator().manual_seed(1234 + seed0)
    feats = torch.randn(vc.shape[0], C, generator=g)
    return vc, feats.numpy()


class Recorder:
    """Wraps the reference op entry points used by Block.forward and records outputs."""

    def __init__(self, mods):
        self.bb, self.utils, self.ops, self.pn2 = mods
        self.rec = {}
        self._orig = {}

    def __enter__(self):
        def wrap(mod, name, keys):
            orig = getattr(mod, name)
            self._orig[(mod, name)] = orig
            counter = {"n": 0}

            def f(*a, **k):
                out = orig(*a, **k)
                outs = out if isinstance(out, tuple) else (out,)
                i = counter["n"]
                counter["n"] += 1
                for kk, o in zip(keys, outs):
                    self.rec["%s.%d.%s" % (name, i, kk)] = o.detach().cpu().numpy().copy()
                return out

            setattr(mod, name, f)

        wrap(self.ops, "get_non_empty_window_center", ["win_ind", "win_table"])
        wrap(self.ops, "gather_two_window_voxels",
             ["ind_odd", "ind_even", "ind_win1", "ind_win2", "coord_odd", "coord_even",
              "coord_win1", "coord_win2"])
        wrap(self.ops, "gather_one_window_voxels", ["ind_win1", "coord_win1"])
        wrap(self.pn2, "farthest_point_sample", ["fps_ind"])
        wrap(self.pn2, "three_nn", ["dist", "idx"])
        return self

    def __exit__(self, *a):
        for (mod, name), orig in self._orig.items():
            setattr(mod, name, orig)


def gen_attention(mods):
    _, utils, _, _ = mods
    torch.manual_seed(7)
    attn = utils.MixedScaleAttention(embed_dim=32, num_heads=[2, 2]).eval()
    b, nq, nk = 5, 6, 8
    q = torch.randn(b, nq, 32)
    k = torch.randn(b, 2 * nk, 32)
    qm = torch.rand(b, nq) < 0.4
    km = torch.rand(b, 2 * nk) < 0.5
    km[:, 0] = False
    km[:, nk] = False
    km[2, nk:] = True  # an all-masked key group: softmax over uniform -100 logits
    with torch.no_grad():
        out = attn(query=q, keys=k, query_mask=qm, key_masks=km, batch_first=True)
    np.savez_compressed(os.path.join(OUT, "attention_block.npz"), query=q.numpy(), keys=k.numpy(),
                        query_mask=qm.numpy(), key_masks=km.numpy(), out=out.numpy(),
                        embed_dim=32, num_heads=np.array([2, 2]), **sd_to_np(attn.state_dict()))

    torch.manual_seed(8)
    attn = utils.MixedScaleAttention(embed_dim=32, num_heads=[4]).eval()
    b, ns = 7, 10
    q = torch.randn(1, b, 32)
    k = torch.randn(ns, b, 32)
    km = torch.rand(b, ns) < 0.5
    km[:, 0] = False
    with torch.no_grad():
        out = attn(query=q, keys=k, key_masks=km)
    np.savez_compressed(os.path.join(OUT, "attention_compress.npz"), query=q.numpy(), keys=k.numpy(),
                        key_masks=km.numpy(), out=out.numpy(), embed_dim=32,
                        num_heads=np.array([4]), **sd_to_np(attn.state_dict()))


def make_block(bb, cls, C, ff, Cout, heads, window_size, max1, max2, cbs_pattern=1,
               interp=True, key_num_sample=32):
    if cls == "block":
        return bb.MixedScaleSparseTransformerBlock(
            cfg=None, in_channels=C, ff_channels=ff, out_channels=Cout, num_heads=heads,
            drop_path=0.0, window_size=window_size, max_num_win1=max1, max_num_win2=max2,
            cbs_mode="odd_even", cbs_pattern=cbs_pattern, key_num_sample=key_num_sample,
            use_feature_interpolation=interp).eval()
    return bb.MixedScaleSparseTransformerCompressBlock(
        cfg=None, in_channels=C, ff_channels=ff, out_channels=Cout, num_heads=heads,
        drop_path=0.0, window_size=window_size, max_num_win1=max1).eval()


def gen_query_tables(mods):
    bb = mods[0]
    out = {}
    cfgs = {
        "w335_777": ([[3, 3, 5], [7, 7, 7]], None, None),
        "w222_444": ([[2, 2, 2], [4, 4, 4]], 8, 64),
        "w557_bbb": ([[5, 5, 7], [11, 11, 11]], None, None),
        "w115": ([[1, 1, 5]], None, None),
        "w3316": ([[3, 3, 16]], None, None),
    }
    for name, (ws, m1, m2) in cfgs.items():
        blk = make_block(bb, "block" if len(ws) == 2 else "compress", 16, 32, 16,
                         [1, 1] if len(ws) == 2 else [2], ws, m1, m2)
        for k, v in blk.vox_query_table.items():
            out["%s.%s" % (name, k)] = v.numpy()
        out["%s.window_size" % name] = np.array(ws)
        if blk.max_num_odd is not None:
            out["%s.max_num_odd" % name] = blk.max_num_odd
            out["%s.max_num_even" % name] = blk.max_num_even
    np.savez_compressed(os.path.join(OUT, "query_tables.npz"), **out)


def run_block(mods, name, cls, window_size, heads, max1, max2, cbs_pattern, interp, seed,
              C=32, ff=64, Cout=32, B=2, pts=1500, key_num_sample=32, empty_sample=None):
    bb, utils, _, _ = mods
    vc, feats = toy_scene(B, pts, seed, C, empty_sample)
    torch.manual_seed(100 + seed)
    blk = make_block(bb, cls, C, ff, Cout, heads, window_size, max1, max2, cbs_pattern, interp,
                     key_num_sample)
    # non-trivial LayerNorm affine + biases so that every parameter is exercised
    with torch.no_grad():
        for p in blk.parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    sp = utils.SparseTensor(features=torch.from_numpy(feats), indices=torch.from_numpy(vc),
                            spatial_shape=list(GRID), voxel_size=list(VOXEL),
                            point_cloud_range=list(PC_RANGE), batch_size=B, hash_size=HASH)
    map_table = sp.map_table.numpy().copy()
    with Recorder(mods) as r, torch.no_grad():
        out = blk(sp)
    d = dict(voxel_coords=vc, voxel_features=feats, map_table=map_table,
             out_features=out.features.numpy(), out_indices=out.indices.numpy(),
             out_spatial_shape=np.array(out.spatial_shape),
             out_voxel_size=np.array(out.voxel_size, dtype=np.float64),
             window_size=np.array(window_size), num_heads=np.array(heads),
             max_num_win1=-1 if max1 is None else max1, max_num_win2=-1 if max2 is None else max2,
             cbs_pattern=cbs_pattern, use_feature_interpolation=interp, channels=np.array([C, ff, Cout]),
             batch_size=B, hash_size=HASH, grid_size=np.array(GRID), voxel_size=np.array(VOXEL),
             point_cloud_range=np.array(PC_RANGE), key_num_sample=key_num_sample)
    if cls == "compress":
        d["out_map_table"] = out.map_table.numpy()
    for k, v in blk.vox_query_table.items():
        d["qt." + k] = v.numpy()
    d.update(sd_to_np(blk.state_dict()))
    d.update({"rec." + k: v for k, v in r.rec.items()})
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print(name, "N=%d nw=%d" % (vc.shape[0], r.rec["get_non_empty_window_center.0.win_ind"].shape[0]),
          "out", out.features.shape)


def gen_backbone(mods, name="backbone", params=None, out_features=48, seed=40, C=32, pts=1500):
    bb, utils, _, _ = mods
    params = params or [
        dict(name="MixedScaleSparseTransformerBlock", channels=[C, 64, C], num_heads=[2, 2],
             window_size=[[3, 3, 5], [7, 7, 7]], max_num_win1=45, max_num_win2=343,
             cbs_mode="odd_even", cbs_pattern=1, key_num_sample=32, use_feature_interpolation=True),
        dict(name="MixedScaleSparseTransformerBlock", channels=[C, 64, C], num_heads=[2, 2],
             window_size=[[3, 3, 5], [7, 7, 7]], max_num_win1=45, max_num_win2=343,
             cbs_mode="odd_even", cbs_pattern=0, key_num_sample=32, use_feature_interpolation=True),
        dict(name="MixedScaleSparseTransformerCompressBlock", channels=[C, 64, 48], num_heads=[4],
             window_size=[[1, 1, 16]], max_num_win1=16),
    ]
    cfg = ref_import.AttrDict.wrap(dict(HASH_SIZE=HASH, NUM_OUTPUT_FEATURES=out_features, PARAMS=params))
    B = 2
    vc, feats = toy_scene(B, pts, seed, C)
    torch.manual_seed(4000 + seed)
    net = bb.MixedScaleSparseTransformer(cfg, C, list(GRID), list(VOXEL), list(PC_RANGE)).eval()
    with torch.no_grad():
        for p in net.parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
        bd = net(dict(voxel_features=torch.from_numpy(feats),
                      voxel_coords=torch.from_numpy(vc).float(), batch_size=B))
    sp = bd["encoded_spconv_tensor"]
    dense = sp.dense()
    d = dict(voxel_coords=vc, voxel_features=feats, out_features=sp.features.numpy(),
             out_indices=sp.indices.numpy(), out_spatial_shape=np.array(sp.spatial_shape),
             dense_shape=np.array(dense.shape), dense_sum=dense.sum().item(),
             dense_nonzero_rows=(dense.abs().sum(1) > 0).sum().item(),
             batch_size=B, hash_size=HASH, grid_size=np.array(GRID), voxel_size=np.array(VOXEL),
             point_cloud_range=np.array(PC_RANGE))
    # keep a slice of the dense tensor (the whole thing is order-invariant scatter of out_features)
    d["dense_b0_z0"] = dense[0, :, 0].numpy()
    d.update(sd_to_np(net.state_dict()))
    import json
    d["params_json"] = json.dumps(params)
    d["out_features_dim"] = out_features
    d["in_channels"] = C
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print(name, "out", sp.features.shape, "dense", tuple(dense.shape))


def gen_even_windows(mods):
    """Even window sizes: (w+1)-cell overlapping win1 lists (ref mssvt_backbone.py:94-97), a voxel in the lists of
    several windows; the reference's scatter `select_v[win1_ind.flatten()] = ...` (:320-322) then has duplicate
    rows -- on the CPU the last (highest flat slot) write wins, which is the canonical order of DESIGN.md."""
    WE = [[2, 2, 2], [4, 4, 4]]
    # torch documents an index_put with duplicate indices as undefined; its CPU kernel splits the rows over the
    # intra-op threads, so the winner depends on the thread count.  ONE thread = sequential execution = the
    # last (highest flat slot) write wins: the canonical order the oracle and the HIP paths implement.
    nthreads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        run_block(mods, "block_evenwin_odd_interp", "block", WE, [2, 2], 27, 64, 1, True, seed=30)
        run_block(mods, "block_evenwin_all_nointerp", "block", WE, [1, 3], 27, 64, 2, False, seed=31, key_num_sample=16)
        run_block(mods, "block_evenwin_trunc", "block", WE, [2, 2], 9, 40, 0, True, seed=32)
        # CompressBlocks with even windows: a voxel is a key of up to 8 windows
        run_block(mods, "compress_2x2x4", "compress", [[2, 2, 4]], [4], 45, None, 1, True, seed=33, Cout=48)
        run_block(mods, "compress_2x2x2_groups", "compress", [[2, 2, 2]], [2, 2], 27, None, 1, True, seed=34)
    finally:
        torch.set_num_threads(nthreads)
    # a batch of three whose middle sample is empty (per-sample loops, sample offsets, hash tables)
    run_block(mods, "block_empty_sample", "block", [[3, 3, 5], [7, 7, 7]], [2, 2], 45, 343, 1, True, seed=35, B=3,
              pts=1000, empty_sample=1)
    run_block(mods, "compress_empty_sample", "compress", [[1, 1, 16]], [4], 16, None, 1, True, seed=36, B=3, pts=1000,
              empty_sample=1, Cout=48)
    # more keys than list slots (FPS runs past the list: repeated-0 picks), head dimension 8
    run_block(mods, "block_k64_heads44", "block", [[3, 3, 5], [7, 7, 7]], [4, 4], 45, 343, 1, True, seed=37, C=64, ff=128,
              Cout=64, key_num_sample=64)
    # enlarged windows, every win1 voxel a query (BASELINE configs[4] shape)
    run_block(mods, "block_enlarged_stride1", "block", [[5, 5, 7], [11, 11, 11]], [2, 2], 175, 1331, 2, True, seed=38,
              pts=2500)
    # the benchmark configuration's shapes (mssvt_amd/cfgs/mssvt.yaml: C = 128, FF = 256, heads [4,4] / [8]): the
    # kernel instantiations bench.py runs, on a toy scene
    gen_backbone(mods, "backbone_c128", [
        dict(name="MixedScaleSparseTransformerBlock", channels=[128, 256, 128], num_heads=[4, 4],
             window_size=[[3, 3, 5], [7, 7, 7]], max_num_win1=45, max_num_win2=343, cbs_mode="odd_even",
             cbs_pattern=1, key_num_sample=32, use_feature_interpolation=True),
        dict(name="MixedScaleSparseTransformerBlock", channels=[128, 256, 128], num_heads=[4, 4],
             window_size=[[3, 3, 5], [7, 7, 7]], max_num_win1=45, max_num_win2=343, cbs_mode="odd_even",
             cbs_pattern=0, key_num_sample=32, use_feature_interpolation=True),
        dict(name="MixedScaleSparseTransformerCompressBlock", channels=[128, 256, 128], num_heads=[8],
             window_size=[[1, 1, 32]], max_num_win1=32)], out_features=128, seed=42, C=128, pts=1000)
    # two resolution levels: a Block on the output of a CompressBlock (window table -> voxel table, scaled voxels)
    C = 32
    blk = dict(name="MixedScaleSparseTransformerBlock", channels=[C, 64, C], num_heads=[2, 2],
               window_size=[[3, 3, 5], [7, 7, 7]], max_num_win1=45, max_num_win2=343, cbs_mode="odd_even",
               key_num_sample=32, use_feature_interpolation=True)
    gen_backbone(mods, "backbone_two_levels", [
        dict(blk, cbs_pattern=1),
        dict(name="MixedScaleSparseTransformerCompressBlock", channels=[C, 64, C], num_heads=[4],
             window_size=[[2, 2, 4]], max_num_win1=45),
        dict(blk, cbs_pattern=0, window_size=[[3, 3, 3], [5, 5, 5]], max_num_win1=27, max_num_win2=125),
        dict(name="MixedScaleSparseTransformerCompressBlock", channels=[C, 64, 48], num_heads=[2, 2],
             window_size=[[1, 1, 8]], max_num_win1=8)], out_features=48, seed=41)


def main():
    os.makedirs(OUT, exist_ok=True)
    mods = ref_import.load()
    if "--grad-only" in sys.argv:
        gen_gradients(mods)
        return
    if "--backbone-only" in sys.argv:  # the 3-block backbone alone (metadata keys added after the first set)
        gen_backbone(mods)
        return
    if "--even-only" in sys.argv:  # add the even-window cases without touching the committed files
        gen_even_windows(mods)
        return
    gen_attention(mods)
    gen_query_tables(mods)
    W2 = [[3, 3, 5], [7, 7, 7]]
    run_block(mods, "block_odd_interp", "block", W2, [2, 2], 45, 343, 1, True, seed=10)
    run_block(mods, "block_even_interp", "block", W2, [2, 2], 45, 343, 0, True, seed=11)
    run_block(mods, "block_all_interp", "block", W2, [2, 2], 45, 343, 2, True, seed=12)
    run_block(mods, "block_odd_nointerp", "block", W2, [2, 2], 45, 343, 1, False, seed=13, Cout=48)
    # truncating max_num_* (lists overflow) + asymmetric head groups
    run_block(mods, "block_trunc", "block", W2, [1, 3], 6, 20, 1, True, seed=14, pts=3000,
              key_num_sample=8)
    run_block(mods, "compress_1x1x16", "compress", [[1, 1, 16]], [4], 16, None, 1, True, seed=20, Cout=48)
    run_block(mods, "compress_3x3x5", "compress", [[3, 3, 5]], [2, 2], 45, None, 1, True, seed=21)
    gen_even_windows(mods)
    gen_gradients(mods)
    gen_backbone(mods)




def gen_gradients(mods):
    """Backward of the reference's Block / CompressBlock (its autograd Functions GroupingOperation / GatherOperation /
    pointnet2 GroupingOperation, mssvt_ops.py:172-190, pointnet2_utils.py:53-73,180-197, with the C oracle's K6 / K11
    underneath): gradients of a fixed linear functional of the output w.r.t. the input features and every
    parameter.  Pins row R15 (backward) of the scope table."""
    bb, utils, _, _ = mods
    C = 32
    cases = [("grad_block_odd_interp", "block", [[3, 3, 5], [7, 7, 7]], [2, 2], 45, 343, 1, True, 60),
             ("grad_block_even_nointerp", "block", [[3, 3, 5], [7, 7, 7]], [1, 3], 45, 343, 0, False, 61),
             ("grad_compress_1x1x16", "compress", [[1, 1, 16]], [4], 16, None, 1, True, 62)]
    for name, cls, ws, heads, m1, m2, pattern, interp, seed in cases:
        vc, feats = toy_scene(2, 1200, seed, C)
        torch.manual_seed(100 + seed)
        blk = make_block(bb, cls, C, 64, C, heads, ws, m1, m2, pattern, interp, 32)
        with torch.no_grad():
            for p in blk.parameters():
                if p.dim() == 1:
                    p.add_(0.1 * torch.randn_like(p))
        x = torch.from_numpy(feats).clone().requires_grad_(True)
        sp = utils.SparseTensor(features=x, indices=torch.from_numpy(vc), spatial_shape=list(GRID),
                                voxel_size=list(VOXEL), point_cloud_range=list(PC_RANGE), batch_size=2, hash_size=HASH)
        out = blk(sp)
        g = torch.Generator().manual_seed(seed)
        w = torch.randn(out.features.shape, generator=g)
        (out.features * w).sum().backward()
        d = dict(voxel_coords=vc, voxel_features=feats, loss_weights=w.numpy(), out_features=out.features.detach().numpy(),
                 grad_input=x.grad.numpy(), window_size=np.array(ws), num_heads=np.array(heads),
                 max_num_win1=m1, max_num_win2=-1 if m2 is None else m2, cbs_pattern=pattern,
                 use_feature_interpolation=interp, channels=np.array([C, 64, C]), batch_size=2, hash_size=HASH,
                 grid_size=np.array(GRID), voxel_size=np.array(VOXEL), point_cloud_range=np.array(PC_RANGE),
                 key_num_sample=32)
        for k, v in blk.vox_query_table.items():
            d["qt." + k] = v.numpy()
        d.update(sd_to_np(blk.state_dict()))
        for k, p in blk.named_parameters():
            d["grad." + k] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
        print(name, "N=%d" % vc.shape[0], "|grad_input|", float(x.grad.abs().mean()))


if __name__ == "__main__":
    main()
