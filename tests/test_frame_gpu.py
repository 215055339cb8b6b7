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
