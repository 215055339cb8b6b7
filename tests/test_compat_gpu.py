"""The pybind-level stand-ins (mssvt_amd/mssvt_ops_compat.py, pointnet2_compat.py): called the way the reference's
own Python calls its extension modules (pcdet/ops/mssvt/mssvt_ops.py:10-190, pointnet2_utils.py:10-197: caller
allocates and pre-fills every output, wrappers fill in place and return 1) and compared with the maintained
bindings, which are pinned to the oracle / goldens in tests/test_ops_gpu.py."""
import numpy as np
import pytest
import torch

from mssvt_amd import synthetic

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _scene(B=2, pts=20000, seed=4):
    vc, _, _ = synthetic.voxelize_numpy(synthetic.make_batch_points(pts, B, seed))
    idx = torch.from_numpy(vc).to(DEV)
    cnt = torch.bincount(idx[:, 0].long(), minlength=B).int()
    return idx, cnt


def test_mssvt_ops_cuda_stand_in_matches_the_maintained_bindings():
    from mssvt_amd import mssvt_ops, mssvt_ops_compat as cuda_mod, query_table
    B, H = 2, 40009
    X, Y, Z = synthetic.GRID_SIZE
    idx, cnt = _scene(B)
    n = idx.shape[0]
    # BuildHashTable.forward (ref :10-20)
    dense_map = torch.zeros((B, H, 2)).int().fill_(-1).to(DEV)
    assert cuda_mod.build_mapping_with_hash_wrapper(X, Y, Z, n, H, idx, cnt, dense_map) == 1
    assert torch.equal(dense_map, mssvt_ops.build_hash_table(B, H, [X, Y, Z], idx, cnt))
    # WindowPartition.forward (ref :31-54)
    win, max_wins = [3, 3, 5], 9000
    wg = [X // win[0], Y // win[1], Z // win[2]]
    win_map = torch.zeros((B, H, 2)).int().fill_(-1).to(DEV)
    win_indices = torch.zeros((B, max_wins, 3)).int().fill_(-1).to(DEV)
    vcount = torch.zeros(B).int().to(DEV)
    assert cuda_mod.window_with_hash_wrapper(wg[0], wg[1], wg[2], win[0], win[1], win[2], n, max_wins, H, idx,
                                             win_indices, win_map, vcount) == 1
    rows = []
    for i in range(B):
        w = win_indices[i]
        w = w[w[:, 0] >= 0]
        rows.append(torch.cat([torch.full((w.shape[0], 1), i, dtype=torch.int32, device=DEV), w], 1))
    win_list = torch.cat(rows, 0)
    want_list, want_map = mssvt_ops.get_non_empty_window_center(win, max_wins, B, H, wg, idx)
    assert torch.equal(win_list, want_list) and torch.equal(win_map, want_map)
    assert int(vcount.sum()) == win_list.shape[0]
    # GatherTwoWindowVoxels / GatherOneWindowVoxels (ref :66-127)
    tabs_, _, _ = query_table.vox_query_table([3, 3, 5], [7, 7, 7])
    t = {k: torch.as_tensor(np.asarray(v), dtype=torch.int32).contiguous().to(DEV) for k, v in tabs_.items()}
    nw = win_list.shape[0]
    maxes = [20, 5, 45, 343]
    tabs = [t['odd'], t['even'], t['win1'], t['win2']]
    inds = [torch.zeros((nw, m)).int().fill_(-1).to(DEV) for m in maxes]
    coords = [torch.zeros((nw, m, 3)).int().to(DEV) for m in maxes]
    assert cuda_mod.gather_two_window_voxels_with_hash_wrapper(X, Y, Z, 3, 3, 5, *maxes, nw, H,
                                                               *[x.shape[0] for x in tabs], *inds, *coords, *tabs,
                                                               win_list, dense_map) == 1
    want = mssvt_ops.gather_two_window_voxels([X, Y, Z], [3, 3, 5], *maxes, *tabs, win_list, dense_map)
    for got, w_ in zip(inds + coords, want):
        assert torch.equal(got, w_)
    ind1, coord1 = torch.zeros((nw, 45)).int().fill_(-1).to(DEV), torch.zeros((nw, 45, 3)).int().to(DEV)
    assert cuda_mod.gather_one_window_voxels_with_hash_wrapper(X, Y, Z, 3, 3, 5, 45, nw, H, t['win1'].shape[0], ind1,
                                                               coord1, t['win1'], win_list, dense_map) == 1
    w1 = mssvt_ops.gather_one_window_voxels([X, Y, Z], [3, 3, 5], 45, t['win1'], win_list, dense_map)
    assert torch.equal(ind1, w1[0]) and torch.equal(coord1, w1[1])
    # GroupingOperation forward / backward (ref :136-190)
    C, ns = 32, 45
    feats = torch.randn(n, C, device=DEV)
    kcnt = torch.bincount(win_list[:, 0].long(), minlength=B).int()
    out = torch.zeros((nw, C, ns), device=DEV)
    assert cuda_mod.group_features_wrapper(B, nw, C, ns, feats, cnt, ind1, kcnt, out) == 1
    f2 = feats.clone().requires_grad_(True)
    want_out = mssvt_ops.grouping_operation(f2, cnt, ind1, kcnt)
    assert torch.equal(out, want_out)
    go = torch.randn_like(out)
    grad = torch.zeros((n, C), device=DEV)
    assert cuda_mod.group_features_grad_wrapper(B, nw, C, n, ns, go, ind1, kcnt, cnt, grad) == 1
    want_out.backward(go)
    torch.testing.assert_close(grad, f2.grad, rtol=1e-5, atol=1e-5)  # float atomics: order differs run to run


def test_pointnet2_batch_cuda_stand_in_matches_the_maintained_bindings():
    from mssvt_amd import pointnet2_compat as pointnet2, pointnet2_utils
    g = torch.Generator().manual_seed(3)
    Bn, N, m, C = 6, 343, 32, 16
    xyz = torch.randint(-3, 4, (Bn, N, 3), generator=g).float().to(DEV)  # integer offsets: plenty of exact ties
    # FurthestPointSampling.forward (ref pointnet2_utils.py:10-30)
    idx = torch.zeros((Bn, m), dtype=torch.int32, device=DEV)
    temp = torch.full((Bn, N), 1e10, device=DEV)
    assert pointnet2.farthest_point_sampling_wrapper(Bn, N, m, xyz, temp, idx) == 1
    assert torch.equal(idx, pointnet2_utils.farthest_point_sample(xyz, m))
    # GatherOperation (ref :39-73)
    feats = torch.randn(Bn, C, N, generator=g).to(DEV)
    out = torch.zeros((Bn, C, m), device=DEV)
    assert pointnet2.gather_points_wrapper(Bn, C, N, m, feats, idx, out) == 1
    f2 = feats.clone().requires_grad_(True)
    want = pointnet2_utils.gather_operation(f2, idx)
    assert torch.equal(out, want)
    go = torch.randn_like(out)
    grad = torch.zeros_like(feats)
    assert pointnet2.gather_points_grad_wrapper(Bn, C, N, m, go, idx, grad) == 1
    want.backward(go)
    torch.testing.assert_close(grad, f2.grad, rtol=1e-5, atol=1e-5)
    # ThreeNN (ref :76-105)
    unknown, known = torch.randn(Bn, 45, 3, generator=g).to(DEV), torch.randn(Bn, 20, 3, generator=g).to(DEV)
    dist2, nn_idx = torch.zeros((Bn, 45, 3), device=DEV), torch.zeros((Bn, 45, 3), dtype=torch.int32, device=DEV)
    pointnet2.three_nn_wrapper(Bn, 45, 20, unknown, known, dist2, nn_idx)
    wd, wi = pointnet2_utils.three_nn(unknown, known)
    assert torch.equal(nn_idx, wi) and torch.equal(torch.sqrt(dist2), wd)
    # GroupingOperation (ref :165-197)
    gidx = torch.randint(0, N, (Bn, 45, 3), generator=g).int().to(DEV)
    gout = torch.zeros((Bn, C, 45, 3), device=DEV)
    assert pointnet2.group_points_wrapper(Bn, C, N, 45, 3, feats, gidx, gout) == 1
    f3 = feats.clone().requires_grad_(True)
    gw = pointnet2_utils.grouping_operation(f3, gidx)
    assert torch.equal(gout, gw)
    ggo = torch.randn_like(gout)
    ggrad = torch.zeros_like(feats)
    assert pointnet2.group_points_grad_wrapper(Bn, C, N, 45, 3, ggo, gidx, ggrad) == 1
    gw.backward(ggo)
    torch.testing.assert_close(ggrad, f3.grad, rtol=1e-5, atol=1e-5)
    with pytest.raises(NotImplementedError):
        pointnet2.ball_query_wrapper()
