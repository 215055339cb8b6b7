"""Frames in flight (mssvt_amd/pipeline.py): every frame of a pipelined run is bit-identical to the same frame run alone,
whatever shares the GPU with it -- the per-stream frame objects share nothing they write."""
import pytest
import torch

from mssvt_amd import synthetic

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("depth", [2, 4])
def test_pipelined_frames_are_bit_identical_to_frames_run_alone(depth):
    from mssvt_amd import config, frame
    from mssvt_amd.pipeline import FramePipeline
    torch.manual_seed(0)
    net = config.build_backbone_from_cfg(config.load_yaml(config.DEFAULT_CFG)).to(DEV).eval()
    scenes = []
    for f in range(5):
        vc, _, _ = synthetic.voxelize_numpy(synthetic.make_batch_points(20000 + 3000 * f, 1, 100 + f))
        feats = torch.randn(vc.shape[0], 128, generator=torch.Generator().manual_seed(f)).to(DEV)
        scenes.append(dict(voxel_features=feats, voxel_coords=torch.from_numpy(vc).to(DEV), batch_size=1))
    alone = []
    with torch.no_grad():
        for sc in scenes:
            sp = net(dict(sc))["encoded_spconv_tensor"]
            alone.append((sp.features.clone(), sp.indices.clone()))
    torch.cuda.synchronize()
    pipe = FramePipeline(net, depth=depth)
    assert pipe.own_queues and len({s.cuda_stream for s in pipe.streams}) == depth  # a hardware queue of its own per stream
    outs = [pipe(dict(scenes[i % len(scenes)])) for i in range(4 * len(scenes))]  # several rounds: workspaces are reused
    pipe.synchronize()
    # ... and with the host wait deferred (what bench.py runs): a frame's result is fetched `depth` submissions later at the latest
    pend = [pipe(dict(scenes[i % len(scenes)]), inputs_ready=True, defer=True) for i in range(4 * len(scenes))]
    assert sum(p is not None for p in pipe.pending) == depth  # the last `depth` frames have not been waited for yet
    pipe.synchronize()
    assert all(p is None for p in pipe.pending)
    outs += [p.get() for p in pend]
    for i, out in enumerate(outs):
        sp = out["encoded_spconv_tensor"]
        f, idx = alone[(i % (4 * len(scenes))) % len(scenes)]
        assert torch.equal(sp.indices, idx) and torch.equal(sp.features, f), i
    # one frame object (workspace) per stream, all on the whole-frame C call
    st = net.__dict__["_frame_state"]
    assert len(st["frames"]) == depth + 1 and all(fr is not None for fr in st["frames"].values())
    assert len({fr.workspace.data_ptr() for fr in st["frames"].values()}) == depth + 1


def test_pipeline_over_any_callable_and_close():
    """The pipeline also takes a whole chain (bench.py --from-points: VFE -> backbone -> BEV): every frame's ops run on that
    frame's stream, results equal the chain run alone; close() gives the per-stream frame objects back."""
    from mssvt_amd import config
    from mssvt_amd.dynamic_vfe import DynamicVFE
    from mssvt_amd.pipeline import FramePipeline
    torch.manual_seed(0)
    cfg = config.load_yaml(config.DEFAULT_CFG)
    net = config.build_backbone_from_cfg(cfg).to(DEV).eval()
    vfe = DynamicVFE(cfg.MODEL.VFE, 5, synthetic.VOXEL_SIZE, synthetic.GRID_SIZE, synthetic.POINT_CLOUD_RANGE).to(DEV).eval()
    clouds = [torch.from_numpy(synthetic.make_batch_points(20000, 1, 40 + f)).to(DEV) for f in range(3)]

    def chain(bd):
        with torch.no_grad():
            bd = net(vfe(bd))
        bd["bev"] = bd["encoded_spconv_tensor"].dense()
        return bd
    alone = [chain(dict(points=c, batch_size=1))["bev"].clone() for c in clouds]
    torch.cuda.synchronize()
    pipe = FramePipeline(chain, depth=3, device=torch.device(DEV, 0))
    outs = [pipe(dict(points=clouds[i % 3], batch_size=1)) for i in range(9)]
    pipe.synchronize()
    for i, o in enumerate(outs):
        assert torch.equal(o["bev"], alone[i % 3]), i
    n_before = len(net.__dict__["_frame_state"]["frames"])
    pipe.close()
    assert pipe.streams == [] and len(net.__dict__["_frame_state"]["frames"]) <= n_before
