"""The whole-frame C entry point (mssvt_frame_forward, csrc/frame.hip; mssvt_amd/frame.py) against the Python-driven
fused path it replaces: same kernels, same arguments -> bit-identical frames; and its fall-backs."""
import numpy as np
import pytest
import torch

from mssvt_amd import synthetic

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _scene(points, B, seed, C=128):
    pts = synthetic.make_batch_points(points, B, seed)
    vc, _, _ = synthetic.voxelize_numpy(pts)
    feats = torch.randn(vc.shape[0], C, generator=torch.Generator().manual_seed(seed))
    return feats.to(DEV), torch.from_numpy(vc).to(DEV)


def _net(seed=0):
    from mssvt_amd import config
    torch.manual_seed(seed)
    return config.build_backbone_from_cfg().to(DEV).eval()


def _run(net, feats, vc, B, frame_on, monkeypatch):
    from mssvt_amd import frame
    monkeypatch.setattr(frame, "ENABLED", frame_on)
    calls = []
    real = frame.forward

    def spy(*a, **k):
        r = real(*a, **k)
        calls.append(r is not None)
        return r
    monkeypatch.setattr(frame, "forward", spy)
    with torch.no_grad():
        sp = net(dict(voxel_features=feats, voxel_coords=vc, batch_size=B))["encoded_spconv_tensor"]
    torch.cuda.synchronize()
    monkeypatch.setattr(frame, "forward", real)
    return sp, (bool(calls) and calls[0])


def _same(a, b):
    assert a.features.shape == b.features.shape and a.indices.shape == b.indices.shape
    assert torch.equal(a.indices, b.indices)
    assert torch.equal(a.features, b.features)  # bit-identical: the same launches with the same arguments
    assert torch.equal(a.map_table, b.map_table)
    assert torch.equal(a.v_bs_cnt, b.v_bs_cnt)
    assert list(a.spatial_shape) == list(b.spatial_shape) and list(a.voxel_size) == list(b.voxel_size)
    assert torch.equal(a.dense(), b.dense())


@pytest.mark.parametrize("points,B", [(20000, 1), (20000, 3), (160000, 1)])
def test_frame_call_is_bit_identical_to_the_python_driven_path(points, B, monkeypatch):
    net = _net()
    feats, vc = _scene(points, B, 5)
    for _ in range(2):  # second frame: persistent workspace reused
        got, used = _run(net, feats, vc, B, True, monkeypatch)
        assert used
        want, used = _run(net, feats, vc, B, False, monkeypatch)
        assert not used
        _same(got, want)


def test_frame_call_with_the_side_stream_gives_the_same_frames(monkeypatch):
    """MSSVT_FRAME_OVERLAP: first norm1 + pillar plan on the frame object's second stream (auto for large frames)."""
    from mssvt_amd import frame
    net = _net()
    feats, vc = _scene(20000, 2, 21)
    want, _ = _run(net, feats, vc, 2, False, monkeypatch)
    monkeypatch.setattr(frame, "OVERLAP", "1")
    for _ in range(3):
        got, used = _run(net, feats, vc, 2, True, monkeypatch)
        assert used
        _same(got, want)
    monkeypatch.setattr(frame, "OVERLAP", "0")
    got, used = _run(net, feats, vc, 2, True, monkeypatch)
    assert used
    _same(got, want)


def test_frame_call_bf16_attention_and_other_patterns(monkeypatch):
    net = _net().set_attn_dtype("bf16")
    feats, vc = _scene(20000, 2, 7)
    got, used = _run(net, feats, vc, 2, True, monkeypatch)
    want, _ = _run(net, feats, vc, 2, False, monkeypatch)
    assert used
    _same(got, want)
    net = _net().set_attn_dtype("f32")
    net.backbone[1].cbs_pattern = 2            # stride-1 queries (whole win1 list)
    net.backbone[2].use_feature_interpolation = False
    got, used = _run(net, feats, vc, 2, True, monkeypatch)
    want, _ = _run(net, feats, vc, 2, False, monkeypatch)
    assert used
    _same(got, want)


def test_frame_call_follows_parameter_updates(monkeypatch):
    net = _net()
    feats, vc = _scene(20000, 1, 9)
    a, used = _run(net, feats, vc, 1, True, monkeypatch)
    assert used
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(1.01)  # in place: the version counters move
    b, used = _run(net, feats, vc, 1, True, monkeypatch)
    want, _ = _run(net, feats, vc, 1, False, monkeypatch)
    assert used and not torch.equal(a.features, b.features)
    _same(b, want)
    sd = {k: v.clone() for k, v in _net(3).state_dict().items()}
    net.load_state_dict(sd)
    c, used = _run(net, feats, vc, 1, True, monkeypatch)
    want, _ = _run(net, feats, vc, 1, False, monkeypatch)
    assert used
    _same(c, want)


def test_frame_call_declines_what_it_does_not_cover(monkeypatch):
    from mssvt_amd import fused
    net = _net()
    feats, vc = _scene(20000, 1, 11)
    # operator-level implementation requested
    net.set_impl("ops")
    _, used = _run(net, feats, vc, 1, True, monkeypatch)
    assert not used
    net.set_impl("fused")
    # fp32-instruction FFN requested
    monkeypatch.setattr(fused, "FFN_ARITH", "f32")
    a, used = _run(net, feats, vc, 1, True, monkeypatch)
    assert not used
    monkeypatch.setattr(fused, "FFN_ARITH", "f16x3")
    # parameters outside the fp16 range of the split products: the Python path (fp32-instruction kernels) runs
    with torch.no_grad():
        net.backbone[0].linear1.weight.mul_(1e6)
    _, used = _run(net, feats, vc, 1, True, monkeypatch)
    assert not used


def test_frame_call_unsorted_input_is_redone_on_the_order_agnostic_path(monkeypatch):
    """A voxel list in another order: the frame call reports MSSVT_ST_UNSORTED (its outputs are empty), the module redoes
    the frame on the order-agnostic kernels -- the same frame the Python-driven path produces for that list (results DO
    depend on the list order: an FPS-picked empty slot becomes voxel 0 of the sample, ref mssvt_backbone.py:253-256)."""
    feats, vc = _scene(20000, 2, 13)
    g = torch.Generator().manual_seed(0)
    perm = torch.cat([idx[torch.randperm(idx.numel(), generator=g)] for idx in
                      [(vc[:, 0].cpu() == b).nonzero().flatten() for b in range(2)]]).to(DEV)
    f2, v2 = feats[perm].contiguous(), vc[perm].contiguous()
    want, _ = _run(_net(), f2, v2, 2, False, monkeypatch)
    net = _net()
    got, used = _run(net, f2, v2, 2, True, monkeypatch)
    assert not used  # the frame call raised UnsortedVoxels
    _same(got, want)
    assert net._unsorted_skip > 0  # ... and the module backs off from the sorted attempt for the next frames
    got, _ = _run(net, f2, v2, 2, True, monkeypatch)
    _same(got, want)
    # sorted frames afterwards (back-off over): the frame call serves them again
    net._unsorted_skip = 0
    got, used = _run(net, feats, vc, 2, True, monkeypatch)
    want, _ = _run(_net(), feats, vc, 2, False, monkeypatch)
    assert used
    _same(got, want)


def test_frame_call_hash_overflow_is_loud(monkeypatch):
    from mssvt_amd import frame
    from mssvt_amd._lib import MssvtHipError
    net = _net()
    net.hash_size = 1000
    feats, vc = _scene(20000, 1, 3)
    monkeypatch.setattr(frame, "ENABLED", True)
    with pytest.raises(MssvtHipError), torch.no_grad():
        net(dict(voxel_features=feats, voxel_coords=vc, batch_size=1))
    torch.cuda.synchronize()


def test_pillar_lists_inside_the_level_setup_equal_the_plan_kernel():
    """mssvt_level_setup_sorted_pillars (the CompressBlock's K4 lists written where k_col_emit numbers the window) against
    mssvt_level_setup_sorted + mssvt_window_plan_one: k_ind / win_vstart / win_cnt / pair_win bit-identical."""
    import ctypes
    from mssvt_amd import _lib, fused, mssvt_ops
    from mssvt_amd.mssvt_utils import SparseTensor
    net = _net()
    B = 3
    feats, vc = _scene(30000, B, 31)
    cmp_blk = net.backbone[-1]
    with torch.no_grad():
        # reference: the Python-driven path's level + one_scale_plan
        sp = fused.setup_input_level(net.backbone, dict(features=feats, indices=vc.int().contiguous(), spatial_shape=net.grid_size,
                                                        voxel_size=net.voxel_size, point_cloud_range=net.point_cloud_range,
                                                        batch_size=B, hash_size=net.hash_size, gather_dict=None))
        assert sp is not None and sp._level.get("sorted")
        p = fused.one_scale_plan(cmp_blk, sp, sync=False)
        p.host_ev.synchronize()
        nw = p.host_words()[1]
        assert p.disjoint == 2 and nw > 1000
        # the merged form, fresh buffers
        n = vc.shape[0]
        X, Y, Z = (int(v) for v in net.grid_size)
        H = int(net.hash_size)
        dev = feats.device
        blk = net.backbone[0]
        al = lambda v: (int(v) + 63) // 64 * 64  # noqa: E731
        sizes = [64, 128, al(B + 1), 2 * B * X * Y]
        offs = [sum(sizes[:i]) for i in range(4)]
        zero = torch.zeros(sum(sizes), dtype=torch.int32, device=dev)
        i32 = lambda *shape: torch.empty(shape, dtype=torch.int32, device=dev)  # noqa: E731
        cnt, vbase = i32(B), i32(B * X * Y)
        scratch = i32(int(_lib.lib().mssvt_level_sorted_scratch_ints(B, X, Y)))
        wins = [i32(n, 4), i32(n, 4)]
        table = torch.full((B, H, 2), -1, dtype=torch.int32, device=dev)
        vcounts = i32(2, B)
        ns = cmp_blk.max_num_win1
        k_ind, vstart, wcnt = torch.full((n, ns), -1, dtype=torch.int32, device=dev), i32(n), i32(n)
        pair_win = torch.full((n,), -1, dtype=torch.int32, device=dev)
        t1 = cmp_blk._tables_on(dev)['win1']
        ia = lambda v: (ctypes.c_int * len(v))(*[int(x) for x in v])  # noqa: E731
        pa = lambda ts: (ctypes.c_void_p * len(ts))(*[0 if t is None else t.data_ptr() for t in ts])  # noqa: E731
        parts = [blk, cmp_blk]
        _lib.call("mssvt_level_setup_sorted_pillars", n, B, X, Y, Z, H, vc.int().contiguous().data_ptr(), zero.data_ptr(),
                  ctypes.c_longlong(-zero.numel() * 4), cnt.data_ptr(), zero[offs[2]:].data_ptr(), zero[offs[3]:].data_ptr(),
                  vbase.data_ptr(), zero.data_ptr(), 2, ia([[X, Y, Z][i] // b.win1_size[i] for b in parts for i in range(3)]),
                  ia([w for b in parts for w in b.win1_size]), ia([b.max_num_wins for b in parts]), pa(wins), pa([None, table]),
                  pa([vcounts[0], vcounts[1]]), pa([zero[64:128], zero[128:192]]), scratch.data_ptr(), 1, ns, int(t1.shape[0]),
                  t1.data_ptr(), k_ind.data_ptr(), vstart.data_ptr(), wcnt.data_ptr(), None, pair_win.data_ptr(), None,
                  _lib.stream())
        torch.cuda.synchronize()
        assert int(zero[129]) == nw and int(zero[0]) == 0
        assert torch.equal(wins[1][:nw], p.win_ind[:nw])
        assert torch.equal(k_ind[:nw], p.k_ind[:nw])
        assert torch.equal(vstart[:nw], p.win_vstart[:nw]) and torch.equal(wcnt[:nw], p.win_cnt[:nw])
        assert torch.equal(pair_win, p.pair_win[:n])
        assert torch.equal(table, p.win_table)
