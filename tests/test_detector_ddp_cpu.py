"""The detector's training wrap (ref tools/train.py:118-119,143-144: SyncBatchNorm conversion, then DistributedDataParallel
around the WHOLE model) on two gloo ranks: the modules either side of the 3-D backbone -- DynamicVFE (BatchNorm1d, train
branch), BaseBEVBackbone (BatchNorm2d) and CenterHead (targets + losses) -- run for real on the CPU; the MsSVT backbone
between them has no CPU path (the product fails loudly without the HIP library), so a dense-scatter stand-in written here
takes its place and the whole detector under DDP runs in tests/test_bench_gpu.py.  Each rank trains on its OWN scene;
after one step both ranks hold the same parameters (gradients averaged) and the same BatchNorm buffers (broadcast)."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp
from torch import nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


GRID, VOXEL, PCR = [32, 32, 8], [1.0, 1.0, 0.5], [-16.0, -16.0, -2.0, 16.0, 16.0, 2.0]
CLASSES = ["Vehicle", "Pedestrian", "Cyclist"]
HEAD = dict(CLASS_AGNOSTIC=False, CLASS_NAMES_EACH_HEAD=[CLASSES], SHARED_CONV_CHANNEL=16, USE_BIAS_BEFORE_NORM=True, NUM_HM_CONV=2,
            SEPARATE_HEAD_CFG=dict(HEAD_ORDER=["center", "center_z", "dim", "rot"],
                                   HEAD_DICT=dict(center=dict(out_channels=2, num_conv=2), center_z=dict(out_channels=1, num_conv=2),
                                                  dim=dict(out_channels=3, num_conv=2), rot=dict(out_channels=2, num_conv=2))),
            TARGET_ASSIGNER_CONFIG=dict(FEATURE_MAP_STRIDE=1, NUM_MAX_OBJS=20, GAUSSIAN_OVERLAP=0.1, MIN_RADIUS=2),
            LOSS_CONFIG=dict(LOSS_WEIGHTS=dict(cls_weight=1.0, loc_weight=2.0, code_weights=[1.0] * 8)),
            POST_PROCESSING=dict(SCORE_THRESH=0.1, POST_CENTER_LIMIT_RANGE=[-20, -20, -5, 20, 20, 5], MAX_OBJ_PER_SAMPLE=50,
                                 NMS_CONFIG=dict(NMS_TYPE="nms_gpu", NMS_THRESH=0.7, NMS_PRE_MAXSIZE=100, NMS_POST_MAXSIZE=20)))


class _Shell(nn.Module):
    """vfe -> (stand-in for backbone_3d + map_to_bev: max over z of the voxel features, scattered to the BEV grid) ->
    backbone_2d -> dense_head, under the reference's attribute names."""

    def __init__(self):
        super().__init__()
        from mssvt_amd.base_bev_backbone import BaseBEVBackbone
        from mssvt_amd.center_head import CenterHead
        from mssvt_amd.config import Config
        from mssvt_amd.dynamic_vfe import DynamicVFE
        self.vfe = DynamicVFE(dict(NUM_FILTERS=[8, 16], WITH_CLUSTER_CENTER=True, WITH_VOXEL_CENTER=True), 5, VOXEL, GRID, PCR)
        self.backbone_2d = BaseBEVBackbone(Config.wrap(dict(LAYER_NUMS=[1], LAYER_STRIDES=[1], NUM_FILTERS=[16], UPSAMPLE_STRIDES=[1],
                                                           NUM_UPSAMPLE_FILTERS=[16])), 16)
        self.dense_head = CenterHead(Config.wrap(HEAD), 16, 3, CLASSES, np.array(GRID), np.array(PCR), VOXEL,
                                     predict_boxes_when_training=False)

    def forward(self, batch_dict):
        bd = self.vfe(batch_dict)
        c, f = bd["voxel_coords"].long(), bd["voxel_features"]
        B = int(batch_dict["batch_size"])
        flat = (c[:, 0] * GRID[1] + c[:, 2]) * GRID[0] + c[:, 3]
        bev = torch.zeros((B * GRID[1] * GRID[0], f.shape[1]), dtype=f.dtype).index_reduce(0, flat, f, "amax", include_self=True)
        bd["spatial_features"] = bev.view(B, GRID[1], GRID[0], -1).permute(0, 3, 1, 2).contiguous()
        bd = self.dense_head(self.backbone_2d(bd))
        loss, tb = self.dense_head.get_loss()
        return loss


def _scene(rank, B=2, P=2000):
    rng = np.random.default_rng(100 + rank)
    pts = np.zeros((P, 6), np.float32)
    pts[:, 0] = rng.integers(0, B, P)
    pts[:, 1:4] = rng.uniform([-15.5, -15.5, -1.9], [15.5, 15.5, 1.9], (P, 3))
    pts[:, 4:6] = rng.uniform(0, 1, (P, 2))
    gt = np.zeros((B, 6, 8), np.float32)
    for b in range(B):
        n = 5 - b
        gt[b, :n, 0:2] = rng.uniform(-12, 12, (n, 2))
        gt[b, :n, 3:6] = rng.uniform([1.5, 0.8, 1.0], [4.0, 2.0, 2.0], (n, 3))
        gt[b, :n, 6] = rng.uniform(-3, 3, n)
        gt[b, :n, 7] = rng.integers(1, 4, n)
    return torch.from_numpy(pts), torch.from_numpy(gt)


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from mssvt_amd import dist as mdist
    torch.set_num_threads(2)
    d = mdist.init("gloo")
    torch.manual_seed(7 + rank)  # different initial weights per rank: DDP broadcasts rank 0's
    model = _Shell().train()
    ddp = torch.nn.parallel.DistributedDataParallel(model)
    opt = torch.optim.SGD(model.parameters(), lr=1e-2)
    pts, gt = _scene(rank)
    before = {k: v.detach().clone() for k, v in model.named_parameters()}
    loss = ddp(dict(points=pts, batch_size=2, gt_boxes=gt))
    loss.backward()
    opt.step()
    moved = sum(int(not torch.equal(before[k], v)) for k, v in model.named_parameters())
    # buffers (BatchNorm running statistics) are broadcast from rank 0 at the NEXT forward: run it, as a second step would
    ddp(dict(points=pts, batch_size=2, gt_boxes=gt))
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    q.put((rank, float(loss), moved, {k: v.numpy() for k, v in state.items() if "num_batches" not in k}))
    d.barrier()
    d.destroy_process_group()


def test_two_ranks_keep_one_set_of_parameters():
    world, ctx = 2, mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, l0, m0, s0), (_, l1, m1, s1) = res
    assert l0 != l1  # different scenes on the two ranks
    assert m0 > 20 and m0 == m1  # the step moved the parameters of every stage
    assert set(s0) == set(s1) and any(k.startswith("vfe.") for k in s0) and any(k.startswith("dense_head.") for k in s0)
    for k in s0:
        if "running_" in k:  # rank 0's statistics broadcast at the second forward, then each rank's own batch on top
            continue
        np.testing.assert_array_equal(s0[k], s1[k], err_msg=k)


def test_sync_batchnorm_conversion_reaches_every_batchnorm_layer():
    model = _Shell()
    n_bn = sum(isinstance(m, nn.modules.batchnorm._BatchNorm) for m in model.modules())
    assert n_bn >= 8 and any(isinstance(m, nn.BatchNorm1d) for m in model.vfe.modules())
    keys = list(model.state_dict().keys())
    conv = nn.SyncBatchNorm.convert_sync_batchnorm(model)
    assert sum(isinstance(m, nn.SyncBatchNorm) for m in conv.modules()) == n_bn
    assert list(conv.state_dict().keys()) == keys  # checkpoints stay loadable by key
