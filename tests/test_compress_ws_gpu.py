"""mssvt_compress_ws (csrc/compress_ws.hip): the CompressBlock attention of a sorted pillar level in one launch, against a
float64 evaluation of the reference formulas (mssvt_backbone.py:351-383, mssvt_utils.py:112-150) on the same K4 lists, and
against the three-launch form it replaces (mssvt_compress_fused)."""
import numpy as np
import pytest
import torch

from mssvt_amd import synthetic

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _compress_block(C=128, heads=8, ws=(1, 1, 32), ns=32, seed=0):
    from mssvt_amd.mssvt_backbone import MixedScaleSparseTransformerCompressBlock
    torch.manual_seed(seed)
    return MixedScaleSparseTransformerCompressBlock(cfg=None, in_channels=C, ff_channels=2 * C, out_channels=C, num_heads=[heads],
                                                    drop_path=0.0, window_size=[list(ws)], max_num_win1=ns).to(DEV).eval()


def _sp(points, B, seed, C=128, scale=1.0):
    from mssvt_amd.mssvt_utils import SparseTensor
    pts = synthetic.make_batch_points(points, B, seed)
    vc, _, _ = synthetic.voxelize_numpy(pts)
    feats = torch.randn(vc.shape[0], C, generator=torch.Generator().manual_seed(seed)) * scale
    return SparseTensor(features=feats.to(DEV), indices=torch.from_numpy(vc).to(DEV), spatial_shape=synthetic.GRID_SIZE,
                        voxel_size=synthetic.VOXEL_SIZE, point_cloud_range=synthetic.POINT_CLOUD_RANGE, batch_size=B,
                        hash_size=200003)


def _reference_f64(blk, sp, p, xhat):
    """Attention output (before the FFN tail) of every window in float64, from the K4 lists of the plan."""
    nw = int(p.num_wins.item())
    ns, C = blk.max_num_win1, xhat.shape[1]
    ma = blk.ms_attn
    d = lambda t: t.detach().double()  # noqa: E731
    k_ind = p.k_ind[:nw].long()
    valid = k_ind >= 0
    rows = (p.win_vstart[:nw].long()[:, None] + k_ind.clamp(min=0))  # (nw, ns) global rows
    x = d(xhat)[rows] * valid[..., None]                              # zero padded key features
    q_tok = x.max(dim=1).values                                       # ref :370 (zeros take part when a slot is empty)
    vs = torch.tensor(sp.voxel_size, dtype=torch.float32, device=DEV)
    mn = torch.tensor(sp.point_cloud_range[:3], dtype=torch.float32, device=DEV)
    wsz = torch.tensor(p.win_size_m, dtype=torch.float32, device=DEV)
    vxyz = sp.indices[:, [3, 2, 1]].float()
    wxyz = p.win_ind[:nw][:, [3, 2, 1]].float()
    vc = ((vxyz + 0.5) * vs + mn)[rows]                               # fp32 exactly as the kernels form them
    wc = ((wxyz + 0.5) * wsz + mn)[:, None, :].expand(-1, ns, -1)
    geo = torch.cat([vc - wc, wc], dim=-1).double()                   # NOT masked (ref :372)
    W1, b1 = d(blk.pos_proj[0].weight).reshape(C, 6), d(blk.pos_proj[0].bias)
    W2, b2 = d(blk.pos_proj[2].weight).reshape(C, C), d(blk.pos_proj[2].bias)
    pos = torch.relu(torch.relu(geo @ W1.T + b1) @ W2.T + b2)
    k_tok = x + pos
    q = (q_tok @ d(ma.to_qs[0].weight).T + d(ma.to_qs[0].bias)) * ma.scale
    kv = k_tok @ d(ma.to_kvs[0].weight).T + d(ma.to_kvs[0].bias)
    K, V = kv[..., :C], kv[..., C:]
    hd = ma.per_head_dim
    s = (q[:, None, :] * K).reshape(nw, ns, C // hd, hd).sum(-1)      # (nw, ns, heads)
    s = s.masked_fill(~valid[..., None], float("-inf"))
    pr = torch.softmax(s, dim=1)
    o = (pr[..., None] * V.reshape(nw, ns, C // hd, hd)).sum(1).reshape(nw, C)
    return o @ d(ma.projs[0].weight).T + d(ma.projs[0].bias)


def _attention_only(blk, sp, ws_on, monkeypatch):
    """The attention output `new` (nw, C) of the fused CompressBlock path with / without mssvt_compress_ws."""
    from mssvt_amd import fused
    monkeypatch.setattr(fused, "CMP_WS", ws_on)
    grabbed = {}
    real = fused._compress_fused_tail

    def spy(block, sp_, p, new):
        grabbed["new"], grabbed["p"] = new.clone(), p
        return real(block, sp_, p, new)
    monkeypatch.setattr(fused, "_compress_fused_tail", spy)
    calls = []
    real_call = fused._lib.call

    def call_spy(name, *a):
        calls.append(name)
        return real_call(name, *a)
    monkeypatch.setattr(fused._lib, "call", call_spy)
    with torch.no_grad():
        xhat = fused._norm1(blk, sp, sp.features)
        fused._compress_forward_fused(blk, sp, xhat, sp.features.contiguous())
    torch.cuda.synchronize()
    monkeypatch.setattr(fused._lib, "call", real_call)
    monkeypatch.setattr(fused, "_compress_fused_tail", real)
    return grabbed["new"], grabbed["p"], xhat, calls


@pytest.mark.parametrize("points,B,ws,ns", [(20000, 1, (1, 1, 32), 32), (20000, 3, (1, 1, 32), 32), (60000, 2, (1, 1, 16), 16),
                                            (160000, 1, (1, 1, 8), 8), (3000, 1, (1, 1, 32), 32)])
def test_one_launch_compress_attention_matches_float64(points, B, ws, ns, monkeypatch):
    blk = _compress_block(ws=ws, ns=ns)
    sp = _sp(points, B, 3)
    new, p, xhat, calls = _attention_only(blk, sp, True, monkeypatch)
    assert "mssvt_compress_ws" in calls and "mssvt_compress_fused" not in calls
    nw = int(p.num_wins.item())
    want = _reference_f64(blk, sp, p, xhat)
    got = new[:nw].double()
    scale = max(1.0, float(want.abs().max()))
    err = float((got - want).abs().max())
    assert err <= 2e-5 * scale, "max err %.3e (scale %.3f)" % (err, scale)
    # ... and the three-launch form on the same level: the same numbers up to the association of the softmax sums
    sp2 = _sp(points, B, 3)
    old, p2, _, calls2 = _attention_only(blk, sp2, False, monkeypatch)
    assert "mssvt_compress_fused" in calls2 and "mssvt_compress_ws" not in calls2
    assert int(p2.num_wins.item()) == nw
    assert float((old[:nw].double() - want).abs().max()) <= 2e-5 * scale
    assert float((old[:nw] - new[:nw]).abs().max()) <= 2e-5 * scale


def test_one_launch_compress_with_scores_far_apart(monkeypatch):
    """Scores of a window more than 2^100 apart (queries scaled up 60 x): the pieces that hold such rows take the merge form of
    the fold (the reference exponent moves), the others the plain sums -- one result, the float64 softmax."""
    blk = _compress_block()
    with torch.no_grad():
        blk.ms_attn.to_qs[0].weight.mul_(60.0)
        blk.ms_attn.to_qs[0].bias.mul_(60.0)
    sp = _sp(60000, 1, 8)
    new, p, xhat, calls = _attention_only(blk, sp, True, monkeypatch)
    assert "mssvt_compress_ws" in calls
    nw = int(p.num_wins.item())
    want = _reference_f64(blk, sp, p, xhat)
    scale = max(1.0, float(want.abs().max()))
    # (scores of magnitude ~1e3 carry an absolute error ~1e-4 in fp32: the weights of near-tied rows move by that much)
    err = (new[:nw].double() - want).abs()
    assert float(err.max()) <= 2e-3 * scale and float(err.mean()) <= 2e-5 * scale


def test_one_launch_compress_is_deterministic_and_ignores_stale_memory(monkeypatch):
    blk = _compress_block()
    outs = []
    for fill in (0.0, float("nan")):
        junk = torch.full((64 << 20,), fill, device=DEV)  # what the allocator hands out next
        del junk
        sp = _sp(20000, 2, 9)
        new, p, _, _ = _attention_only(blk, sp, True, monkeypatch)
        outs.append(new[:int(p.num_wins.item())].clone())
    assert torch.equal(outs[0], outs[1])


def test_one_launch_compress_does_not_depend_on_the_batch_composition(monkeypatch):
    """a scene's rows are bit-identical whether it is processed alone or behind another scene -- the softmax sums
    are a left fold in row order, independent of how the kernel cuts the level into chunks and 16-row pieces."""
    from mssvt_amd.mssvt_utils import SparseTensor
    blk = _compress_block()
    both = _sp(60000, 2, 21)
    new, p, _, _ = _attention_only(blk, both, True, monkeypatch)
    nw = int(p.num_wins.item())
    wb = p.win_ind[:nw, 0]
    for b in range(2):
        sel = both.indices[:, 0] == b
        idx = both.indices[sel].clone()
        idx[:, 0] = 0
        one = SparseTensor(features=both.features[sel].clone(), indices=idx, spatial_shape=synthetic.GRID_SIZE,
                           voxel_size=synthetic.VOXEL_SIZE, point_cloud_range=synthetic.POINT_CLOUD_RANGE, batch_size=1,
                           hash_size=200003)
        got, p1, _, _ = _attention_only(blk, one, True, monkeypatch)
        n1 = int(p1.num_wins.item())
        assert n1 == int((wb == b).sum())
        assert torch.equal(got[:n1], new[:nw][wb == b])


def test_shapes_outside_the_one_launch_form_keep_the_three_launch_form(monkeypatch):
    # a list capacity below the slab height can truncate a list: a window is then no run of rows
    blk = _compress_block(ws=(1, 1, 32), ns=8)
    sp = _sp(20000, 1, 4)
    _, _, _, calls = _attention_only(blk, sp, True, monkeypatch)
    assert "mssvt_compress_fused" in calls and "mssvt_compress_ws" not in calls
    # 4 heads of 32 channels: head = two waves
    blk = _compress_block(heads=4)
    sp = _sp(20000, 1, 4)
    _, _, _, calls = _attention_only(blk, sp, True, monkeypatch)
    assert "mssvt_compress_fused" in calls and "mssvt_compress_ws" not in calls


def test_backbone_with_and_without_the_one_launch_compress(monkeypatch):
    from mssvt_amd import config, fused
    from tests.test_module_gpu import assert_feat_close
    torch.manual_seed(0)
    net = config.build_backbone_from_cfg().to(DEV).eval()
    pts = synthetic.make_batch_points(20000, 2, 5)
    vc, _, _ = synthetic.voxelize_numpy(pts)
    feats = torch.randn(vc.shape[0], 128, generator=torch.Generator().manual_seed(5)).to(DEV)
    outs = []
    for on in (True, False):
        monkeypatch.setattr(fused, "CMP_WS", on)
        net.__dict__.pop("_frame_state", None)
        with torch.no_grad():
            sp = net(dict(voxel_features=feats, voxel_coords=torch.from_numpy(vc).to(DEV), batch_size=2))["encoded_spconv_tensor"]
        outs.append(sp)
    assert torch.equal(outs[0].indices, outs[1].indices) and torch.equal(outs[0].map_table, outs[1].map_table)
    assert_feat_close(outs[0].features.cpu().numpy(), outs[1].features.cpu().numpy())
