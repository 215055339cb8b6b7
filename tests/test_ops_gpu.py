"""Parity of the granular HIP operators (through the C ABI) against the CPU oracle.

Integer / index work is compared BIT-EXACTLY (np.testing.assert_array_equal);
float gathers are copies and also compared exactly; three_nn distances are
compared exactly too (the same fma chain is written in both).
"""
import os

import numpy as np
import pytest
import torch

from mssvt_amd import synthetic
from oracle import block_ref, cref

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _ops():
    from mssvt_amd import mssvt_ops, pointnet2_utils
    return mssvt_ops, pointnet2_utils


def scene(num_points, batch, seed=0, pc_range=None, grid=None):
    pc_range = pc_range or synthetic.POINT_CLOUD_RANGE
    grid = grid or synthetic.GRID_SIZE
    pts = synthetic.make_batch_points(num_points, batch, seed)
    vc, _, _ = synthetic.voxelize_numpy(pts, pc_range, synthetic.VOXEL_SIZE, grid)
    return vc


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


# ---------------------------------------------------------------- K1
@pytest.mark.parametrize("hash_size", [400000, 40009, 0])
def test_build_hash_table(hash_size):
    ops, _ = _ops()
    B = 2
    vc = scene(20000, B, seed=3)
    if hash_size == 0:  # tight table: load factor ~0.97 -> long probe chains / displacement
        hash_size = int(max(cref.bs_cnt(vc, B)) / 0.97) + 1
    cnt = cref.bs_cnt(vc, B)
    want = cref.build_hash_table(B, hash_size, synthetic.GRID_SIZE, vc, cnt)
    got = ops.build_hash_table(B, hash_size, synthetic.GRID_SIZE, t(vc), t(cnt))
    np.testing.assert_array_equal(got.cpu().numpy(), want)


def test_build_hash_table_oob_duplicates_and_overflow():
    ops, _ = _ops()
    rng = np.random.default_rng(0)
    grid = [12, 10, 6]
    B = 3
    rows = []
    for b in range(B):
        n = 300
        xyz = np.stack([rng.integers(-1, 14, n), rng.integers(-1, 12, n), rng.integers(-1, 8, n)], 1)
        rows.append(np.concatenate([np.full((n, 1), b), xyz[:, [2, 1, 0]]], 1))  # [b,z,y,x] incl. OOB + dups
    vc = np.concatenate(rows).astype(np.int32)
    cnt = cref.bs_cnt(vc, B)
    for H in (1024, 97, 211):  # 97 < number of distinct keys -> table overflow (silent drop)
        want = cref.build_hash_table(B, H, grid, vc, cnt)
        got = ops.build_hash_table(B, H, grid, t(vc), t(cnt))
        np.testing.assert_array_equal(got.cpu().numpy(), want, err_msg="H=%d" % H)


def test_build_hash_table_empty_and_single():
    ops, _ = _ops()
    vc = np.array([[0, 1, 2, 3]], np.int32)
    got = ops.build_hash_table(1, 17, [8, 8, 8], t(vc), t(np.array([1], np.int32)))
    want = cref.build_hash_table(1, 17, [8, 8, 8], vc, np.array([1], np.int32))
    np.testing.assert_array_equal(got.cpu().numpy(), want)
    assert (want[0, (3 * 64 + 2 * 8 + 1) % 17] == [3 * 64 + 2 * 8 + 1, 0]).all()  # known answer


# ---------------------------------------------------------------- K2
@pytest.mark.parametrize("win,hash_size", [([3, 3, 5], 400000), ([1, 1, 32], 400000), ([2, 2, 2], 30011),
                                           ([7, 7, 7], 4099)])
def test_window_partition(win, hash_size):
    ops, _ = _ops()
    B = 3
    vc = scene(15000, B, seed=5)
    wgrid = [synthetic.GRID_SIZE[i] // win[i] for i in range(3)]
    want_win, want_tab = cref.get_non_empty_window_center(win, 90000, B, hash_size, wgrid, vc)
    got_win, got_tab = ops.get_non_empty_window_center(win, 90000, B, hash_size, wgrid, t(vc))
    np.testing.assert_array_equal(got_win.cpu().numpy(), want_win)
    np.testing.assert_array_equal(got_tab.cpu().numpy(), want_tab)


def test_window_with_hash_reference_shaped_entry_point():
    """mssvt_window_with_hash fills (B,max,3)/vcount exactly like the reference kernel."""
    import ctypes
    from mssvt_amd import _lib, mssvt_ops
    B, win, H, max_w = 2, [3, 3, 5], 50021, 9000
    vc = scene(8000, B, seed=9)
    wgrid = [synthetic.GRID_SIZE[i] // win[i] for i in range(3)]
    want_win, want_tab = cref.get_non_empty_window_center(win, max_w, B, H, wgrid, vc)
    v = t(vc)
    w_idx = torch.full((B, max_w, 3), -1, dtype=torch.int32, device=DEV)
    tab = torch.full((B, H, 2), -1, dtype=torch.int32, device=DEV)
    vcount = torch.zeros(B, dtype=torch.int32, device=DEV)
    ws = mssvt_ops.hash_workspace(vc.shape[0], B, DEV)
    i = ctypes.c_int
    _lib.call("mssvt_window_with_hash", i(wgrid[0]), i(wgrid[1]), i(wgrid[2]), i(win[0]), i(win[1]),
              i(win[2]), i(vc.shape[0]), i(max_w), i(H), i(B), _lib.ptr(v), _lib.ptr(w_idx), _lib.ptr(tab),
              _lib.ptr(vcount), _lib.ptr(ws), _lib.stream())
    w_idx, vcount = w_idx.cpu().numpy(), vcount.cpu().numpy()
    rows = []
    for b in range(B):
        assert (w_idx[b, vcount[b]:] == -1).all()
        rows.append(np.concatenate([np.full((vcount[b], 1), b, np.int32), w_idx[b, :vcount[b]]], 1))
    np.testing.assert_array_equal(np.concatenate(rows), want_win)
    np.testing.assert_array_equal(tab.cpu().numpy(), want_tab)
    np.testing.assert_array_equal(vcount, cref.bs_cnt(want_win, B))


# ---------------------------------------------------------------- K3 / K4
@pytest.mark.parametrize("ws,maxes", [([[3, 3, 5], [7, 7, 7]], (None, None, 45, 343)),
                                      ([[3, 3, 5], [7, 7, 7]], (3, 2, 6, 20)),
                                      ([[2, 2, 2], [4, 4, 4]], (None, None, 8, 64)),
                                      ([[5, 5, 7], [11, 11, 11]], (None, None, 175, 1331))])
def test_gather_two_window_voxels(ws, maxes):
    ops, _ = _ops()
    B, H = 2, 100003
    vc = scene(30000, B, seed=7)
    tabs, n_odd, n_even = block_ref.vox_query_table(ws[0], ws[1])
    m_odd = maxes[0] or n_odd
    m_even = maxes[1] or n_even
    cnt = cref.bs_cnt(vc, B)
    table = cref.build_hash_table(B, H, synthetic.GRID_SIZE, vc, cnt)
    wgrid = [synthetic.GRID_SIZE[i] // ws[0][i] for i in range(3)]
    win, _ = cref.get_non_empty_window_center(ws[0], 90000, B, H, wgrid, vc)
    want = cref.gather_two_window_voxels(synthetic.GRID_SIZE, ws[0], m_odd, m_even, maxes[2], maxes[3],
                                         tabs["odd"], tabs["even"], tabs["win1"], tabs["win2"], win, table)
    got = ops.gather_two_window_voxels(synthetic.GRID_SIZE, ws[0], m_odd, m_even, maxes[2], maxes[3],
                                       t(tabs["odd"]), t(tabs["even"]), t(tabs["win1"]), t(tabs["win2"]),
                                       t(win), t(table))
    names = ["ind_odd", "ind_even", "ind_win1", "ind_win2", "c_odd", "c_even", "c_win1", "c_win2"]
    for n, g, w in zip(names, got, want):
        np.testing.assert_array_equal(g.cpu().numpy(), w, err_msg=n)


@pytest.mark.parametrize("win,max1", [([1, 1, 32], 32), ([3, 3, 5], 45), ([3, 3, 16], 20), ([2, 2, 2], 8)])
def test_gather_one_window_voxels(win, max1):
    ops, _ = _ops()
    B, H = 2, 100003
    vc = scene(20000, B, seed=8)
    tabs, _, _ = block_ref.vox_query_table(win, None)
    table = cref.build_hash_table(B, H, synthetic.GRID_SIZE, vc, cref.bs_cnt(vc, B))
    wgrid = [synthetic.GRID_SIZE[i] // win[i] for i in range(3)]
    wl, _ = cref.get_non_empty_window_center(win, 90000, B, H, wgrid, vc)
    want = cref.gather_one_window_voxels(synthetic.GRID_SIZE, win, max1, tabs["win1"], wl, table)
    got = ops.gather_one_window_voxels(synthetic.GRID_SIZE, win, max1, t(tabs["win1"]), t(wl), t(table))
    for g, w in zip(got, want):
        np.testing.assert_array_equal(g.cpu().numpy(), w)


# ---------------------------------------------------------------- K5 / K6
@pytest.mark.parametrize("C,ns", [(128, 20), (32, 45), (3, 32), (160, 7), (64, 70)])
def test_group_features_and_grad(C, ns):
    ops, _ = _ops()
    rng = np.random.default_rng(1)
    fcnt = np.array([500, 0, 731], np.int32)
    icnt = np.array([40, 0, 61], np.int32)
    N, M = int(fcnt.sum()), int(icnt.sum())
    feats = rng.standard_normal((N, C)).astype(np.float32)
    idx = np.concatenate([rng.integers(-1, fcnt[0], (icnt[0], ns)), rng.integers(-1, fcnt[2], (icnt[2], ns))]).astype(np.int32)
    idx[rng.random(idx.shape) < 0.5] = -1
    want = cref.grouping_operation(feats, fcnt, idx, icnt)
    f = t(feats).requires_grad_(True)
    got = ops.grouping_operation(f, t(fcnt), t(idx), t(icnt))
    np.testing.assert_array_equal(got.detach().cpu().numpy(), want)
    go = rng.standard_normal(want.shape).astype(np.float32)
    got.backward(t(go))
    want_g = cref.grouping_operation_grad(go, N, idx, icnt, fcnt)
    np.testing.assert_allclose(f.grad.cpu().numpy(), want_g, rtol=1e-5, atol=1e-5)  # atomic sum order


# ---------------------------------------------------------------- K7
def _padded_offsets(rng, b, n, lo, hi, max_valid):
    """integer offsets with zero padding after a random number of valid entries (as K3 produces)."""
    xyz = np.zeros((b, n, 3), np.float32)
    for i in range(b):
        nv = rng.integers(0, min(n, max_valid) + 1)
        pts = rng.integers(lo, hi + 1, (nv, 3))
        xyz[i, :nv] = pts
    return xyz


@pytest.mark.parametrize("n,m", [(45, 32), (343, 32), (27, 8), (8, 8), (64, 32), (100, 32), (1331, 32),
                                 (175, 32), (2, 2), (1, 1), (20, 5), (5000, 16)])
def test_farthest_point_sample_ties(n, m):
    _, pn2 = _ops()
    rng = np.random.default_rng(n)
    xyz = _padded_offsets(rng, 97, n, -3, 3, 40)
    want = cref.farthest_point_sample(xyz, m)
    got = pn2.farthest_point_sample(t(xyz), m)
    np.testing.assert_array_equal(got.cpu().numpy(), want)


def test_farthest_point_sample_float_and_all_ties():
    _, pn2 = _ops()
    rng = np.random.default_rng(2)
    xyz = rng.standard_normal((33, 300, 3)).astype(np.float32)
    np.testing.assert_array_equal(pn2.farthest_point_sample(t(xyz), 32).cpu().numpy(),
                                  cref.farthest_point_sample(xyz, 32))
    z = np.zeros((3, 343, 3), np.float32)  # all points identical: every round is a full tie
    np.testing.assert_array_equal(pn2.farthest_point_sample(t(z), 32).cpu().numpy(),
                                  cref.farthest_point_sample(z, 32))


# ---------------------------------------------------------------- K8 / K9 / K10
def test_gather_points_three_nn_group_points():
    _, pn2 = _ops()
    rng = np.random.default_rng(4)
    B, C, N, M = 50, 16, 45, 20
    feats = rng.standard_normal((B, C, N)).astype(np.float32)
    idx = rng.integers(0, N, (B, 32)).astype(np.int32)
    np.testing.assert_array_equal(pn2.gather_operation(t(feats), t(idx)).cpu().numpy(),
                                  cref.gather_operation(feats, idx))
    # grid-like coordinates -> many exact distance ties; a few known slots at the origin
    unknown = (rng.integers(-3, 4, (B, N, 3)) * np.array([0.32, 0.32, 0.1875])).astype(np.float32) + 10.0
    known = (rng.integers(-3, 4, (B, M, 3)) * np.array([0.32, 0.32, 0.1875])).astype(np.float32) + 10.0
    known[:, 7:] = 0.0
    wd, wi = cref.three_nn(unknown, known)
    gd, gi = pn2.three_nn(t(unknown), t(known))
    np.testing.assert_array_equal(gi.cpu().numpy(), wi)
    np.testing.assert_allclose(gd.cpu().numpy(), wd, rtol=1e-6)
    # fewer than 3 known points: indices default to 0, distance 1e40 -> inf after the cast
    wd2, wi2 = cref.three_nn(unknown[:, :5], known[:, :2])
    gd2, gi2 = pn2.three_nn(t(unknown[:, :5]), t(known[:, :2]))
    np.testing.assert_array_equal(gi2.cpu().numpy(), wi2)
    np.testing.assert_array_equal(np.isinf(gd2.cpu().numpy()), np.isinf(wd2))
    f2 = rng.standard_normal((B, C, M)).astype(np.float32)
    np.testing.assert_array_equal(pn2.grouping_operation(t(f2), t(wi)).cpu().numpy(), cref.group_points(f2, wi))


# ---------------------------------------------------------------- against the reference-run goldens
def test_ops_reproduce_reference_run_intermediates(golden_dir):
    """HIP ops vs the index intermediates recorded while the REFERENCE's Block.forward ran."""
    ops, pn2 = _ops()
    for name in ("block_odd_interp", "block_trunc", "block_evenwin_odd_interp"):
        z = np.load(os.path.join(golden_dir, name + ".npz"))
        B, H = int(z["batch_size"]), int(z["hash_size"])
        vc = z["voxel_coords"]
        grid = z["grid_size"].tolist()
        win1 = z["window_size"][0].tolist()
        cnt = cref.bs_cnt(vc, B)
        table = ops.build_hash_table(B, H, grid, t(vc), t(cnt))
        np.testing.assert_array_equal(table.cpu().numpy(), z["map_table"])
        wgrid = [grid[i] // win1[i] for i in range(3)]
        win, _ = ops.get_non_empty_window_center(win1, 90000, B, H, wgrid, t(vc))
        np.testing.assert_array_equal(win.cpu().numpy(), z["rec.get_non_empty_window_center.0.win_ind"])
        qt = {k: t(z["qt." + k]) for k in ("odd", "even", "win1", "win2")}
        outs = ops.gather_two_window_voxels(grid, win1, qt["odd"].shape[0], qt["even"].shape[0],
                                            int(z["max_num_win1"]), int(z["max_num_win2"]), qt["odd"],
                                            qt["even"], qt["win1"], qt["win2"], win, table)
        keys = ["ind_odd", "ind_even", "ind_win1", "ind_win2", "coord_odd", "coord_even", "coord_win1", "coord_win2"]
        for k, o in zip(keys, outs):
            np.testing.assert_array_equal(o.cpu().numpy(), z["rec.gather_two_window_voxels.0." + k], err_msg=k)
        m = int(z["key_num_sample"])
        f1 = pn2.farthest_point_sample(outs[6].float(), m)
        f2 = pn2.farthest_point_sample(outs[7].float(), m)
        np.testing.assert_array_equal(f1.cpu().numpy(), z["rec.farthest_point_sample.0.fps_ind"])
        np.testing.assert_array_equal(f2.cpu().numpy(), z["rec.farthest_point_sample.1.fps_ind"])
