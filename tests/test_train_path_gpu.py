"""Training path: the compact differentiable forward + deterministic segmented-sum backward
(mssvt_amd/train_path.py, csrc/segment_reduce.hip) against the padded operator-level path (the reference's own
structure, whose gradients are pinned to the reference's backward run in tests/test_module_gpu.py), and run to run."""
import numpy as np
import pytest
import torch

from mssvt_amd import synthetic

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _net(C, params, H, seed=3):
    from mssvt_amd.config import Config
    from mssvt_amd.mssvt_backbone import MixedScaleSparseTransformer
    torch.manual_seed(seed)
    cfg = Config.wrap(dict(NAME="MixedScaleSparseTransformer", HASH_SIZE=H, NUM_OUTPUT_FEATURES=C, PARAMS=params))
    return MixedScaleSparseTransformer(cfg, C, synthetic.GRID_SIZE, synthetic.VOXEL_SIZE,
                                       synthetic.POINT_CLOUD_RANGE).to(DEV).eval()  # eval: DropPath off, grads on


def _params(C, heads, cheads, interp=(True, True)):
    blk = dict(name="MixedScaleSparseTransformerBlock", channels=[C, 2 * C, C], num_heads=heads,
               window_size=[[3, 3, 5], [7, 7, 7]], max_num_win1=45, max_num_win2=343, cbs_mode="odd_even",
               key_num_sample=32)
    return [dict(blk, cbs_pattern=1, use_feature_interpolation=interp[0]),
            dict(blk, cbs_pattern=0, use_feature_interpolation=interp[1]),
            dict(name="MixedScaleSparseTransformerCompressBlock", channels=[C, 2 * C, C], num_heads=cheads,
                 window_size=[[1, 1, 32]], max_num_win1=32)]


def _grads(net, x, coords, B, w):
    for p in net.parameters():
        p.grad = None
    x = x.detach().clone().requires_grad_(True)
    out = net(dict(voxel_features=x, voxel_coords=coords, batch_size=B))["encoded_spconv_tensor"].features
    (out * w).sum().backward()
    return out.detach(), x.grad.detach().clone(), {k: p.grad.detach().clone() for k, p in net.named_parameters()}


@pytest.mark.parametrize("C,heads,cheads,interp,tokens", [(32, [2, 2], [4], (True, False), True), (64, [4, 4], [8], (True, True), True),
                                                         (64, [4, 4], [8], (True, False), False)])
def test_compact_training_path_matches_the_operator_path(C, heads, cheads, interp, tokens, monkeypatch):
    from mssvt_amd import fused, train_path
    monkeypatch.setattr(train_path, "TOKENS", tokens)  # False: the autograd composition of the token sets (other widths' path)
    B, H = 2, 40009
    net = _net(C, _params(C, heads, cheads, interp), H)
    vc, _, _ = synthetic.voxelize_numpy(synthetic.make_batch_points(4000, B, 31))
    coords = torch.from_numpy(vc).to(DEV)
    x = torch.randn(vc.shape[0], C, device=DEV)
    g = torch.Generator(device="cpu").manual_seed(7)
    res = {}
    for compact in (True, False):
        fused.TRAIN_COMPACT = compact
        try:
            out_shape = net(dict(voxel_features=x, voxel_coords=coords, batch_size=B))["encoded_spconv_tensor"].features.shape
            w = torch.randn(out_shape, generator=torch.Generator().manual_seed(7)).to(DEV)
            res[compact] = _grads(net, x, coords, B, w)
        finally:
            fused.TRAIN_COMPACT = True
    (o1, gx1, gp1), (o2, gx2, gp2) = res[True], res[False]

    def close(a, b, what):
        scale = max(1.0, float(b.abs().max()))
        err = float((a - b).abs().max())
        assert err <= 2e-4 * scale, "%s: %.3e vs scale %.3e" % (what, err, scale)

    close(o1, o2, "output")
    close(gx1, gx2, "input gradient")
    for k in gp2:
        close(gp1[k], gp2[k], k)


@pytest.mark.parametrize("which", ["block", "compress"])
def test_compact_training_path_with_overlapping_custom_tables(which):
    """A custom table whose offsets leave the window puts a voxel on the lists of several windows: the inverse of the
    gather is then the general inverted index, not one scatter (every contribution of the backward kept)."""
    from mssvt_amd import fused
    C, B, H = 32, 2, 40009
    net = _net(C, _params(C, [2, 2], [4]), H)
    if which == "block":
        for blk in net.backbone[:2]:
            t = {k: v.clone().cpu() for k, v in blk.vox_query_table.items()}
            far = int((t["win2"][:, 0].abs() == 2).nonzero()[0])  # a cell of the 7x7x7 surround, outside the 3x3x5 window
            t["odd"][-1], t["win2"][far] = t["win2"][far].clone(), t["odd"][-1].clone()
            t["even"][-1, 0] += 2  # stays even, leaves the window
            blk.set_vox_query_table(t)
            assert not fused._lists_disjoint(blk)
    else:
        blk = net.backbone[2]
        t = {k: v.clone().cpu() for k, v in blk.vox_query_table.items()}
        # the pillar's own cells + eight cells of the neighbouring column: a voxel there is on two windows' lists
        extra = torch.tensor([[1, 0, z] for z in range(-4, 4)], dtype=t["win1"].dtype)
        t["win1"] = torch.cat([t["win1"][:12], extra, t["win1"][12:]], 0)
        blk.set_vox_query_table(t)
        assert fused._table_covers_window(blk)
    vc, _, _ = synthetic.voxelize_numpy(synthetic.make_batch_points(4000, B, 11))
    coords = torch.from_numpy(vc).to(DEV)
    x = torch.randn(vc.shape[0], C, device=DEV)
    res = {}
    for compact in (True, False):
        fused.TRAIN_COMPACT = compact
        try:
            shape = net(dict(voxel_features=x, voxel_coords=coords, batch_size=B))["encoded_spconv_tensor"].features.shape
            w = torch.randn(shape, generator=torch.Generator().manual_seed(7)).to(DEV)
            res[compact] = _grads(net, x, coords, B, w)
        finally:
            fused.TRAIN_COMPACT = True
    (o1, gx1, gp1), (o2, gx2, gp2) = res[True], res[False]
    for a, b, what in [(o1, o2, "output"), (gx1, gx2, "input gradient")] + [(gp1[k], gp2[k], k) for k in gp2]:
        scale = max(1.0, float(b.abs().max()))
        assert float((a - b).abs().max()) <= 2e-4 * scale, what


def test_a_table_that_leaves_windows_without_keys_takes_the_operator_path():
    """A custom win1 table that omits cells of the window can leave a non-empty window with an empty key list: the reference
    averages the padded slots there (uniform softmax over -100-masked scores), which only the operator path restates --
    the compact kernels must decline such a block, with and without autograd."""
    from mssvt_amd import fused
    C, B, H = 32, 2, 40009
    net = _net(C, _params(C, [2, 2], [4]), H)
    blk = net.backbone[2]
    t = {k: v.clone().cpu() for k, v in blk.vox_query_table.items()}
    t["win1"][::4, 0] = 1  # every fourth cell of the pillar is looked up in the neighbouring column instead
    blk.set_vox_query_table(t)
    assert not fused._table_covers_window(blk)
    vc, _, _ = synthetic.voxelize_numpy(synthetic.make_batch_points(4000, B, 11))
    coords = torch.from_numpy(vc).to(DEV)
    x = torch.randn(vc.shape[0], C, device=DEV)
    with torch.no_grad():
        got = net(dict(voxel_features=x, voxel_coords=coords, batch_size=B))["encoded_spconv_tensor"].features
    xg = x.clone().requires_grad_(True)
    got_g = net(dict(voxel_features=xg, voxel_coords=coords, batch_size=B))["encoded_spconv_tensor"].features
    net.set_impl("ops")
    with torch.no_grad():
        want = net(dict(voxel_features=x, voxel_coords=coords, batch_size=B))["encoded_spconv_tensor"].features
    scale = max(1.0, float(want.abs().max()))
    assert float((got - want).abs().max()) <= 2e-4 * scale and float((got_g.detach() - want).abs().max()) <= 2e-4 * scale


def test_training_gradients_are_bit_identical_run_to_run():
    """Segmented sums in a fixed order instead of atomics: two backward passes of the same step agree bit for bit
    (input gradient and every parameter gradient), at a size where atomics would reorder (20k points x 2)."""
    C, B, H = 64, 2, 400009
    net = _net(C, _params(C, [4, 4], [8]), H)
    vc, _, _ = synthetic.voxelize_numpy(synthetic.make_batch_points(20000, B, 5))
    coords = torch.from_numpy(vc).to(DEV)
    x = torch.randn(vc.shape[0], C, device=DEV)
    n_out = net(dict(voxel_features=x, voxel_coords=coords, batch_size=B))["encoded_spconv_tensor"].features.shape
    w = torch.randn(n_out, generator=torch.Generator().manual_seed(1)).to(DEV)
    a = _grads(net, x, coords, B, w)
    junk = torch.randn(64 << 20, device=DEV)  # different allocator state / cache contents between the runs
    del junk
    b = _grads(net, x, coords, B, w)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for k in a[2]:
        assert torch.equal(a[2][k], b[2][k]), k


def test_segment_sum_rows_matches_index_add():
    from mssvt_amd import train_path
    torch.manual_seed(0)
    R, N, C = 5000, 700, 48
    src = torch.randn(R, C, device=DEV)
    idx = torch.randint(0, N, (R,), device=DEV)
    csr = train_path.Csr.gather(idx, N)  # dst[i] = table[idx[i]]; its transpose sums rows of src per table row
    got = train_path.segment_sum_rows(src, csr.t_off, csr.t_idx, csr.t_w, N)
    want = torch.zeros(N, C, device=DEV, dtype=torch.float64).index_add_(0, idx, src.double())
    assert float((got.double() - want).abs().max()) < 1e-4
    # weighted form with a row that nothing maps to
    w = torch.rand(R, device=DEV)
    off = torch.arange(R + 1, dtype=torch.int32, device=DEV)
    c2 = train_path.Csr(off, idx.int(), w, N + 3)
    got2 = train_path.segment_sum_rows(src, c2.t_off, c2.t_idx, c2.t_w, N + 3)
    want2 = torch.zeros(N + 3, C, device=DEV, dtype=torch.float64).index_add_(0, idx, src.double() * w.double()[:, None])
    assert float((got2.double() - want2).abs().max()) < 1e-4 and float(got2[N:].abs().max()) == 0.0


@pytest.mark.parametrize("M,cin,cout,bias", [(74270, 128, 256, True), (74270, 256, 128, True), (33001, 64, 64, True),
                                             (185003, 64, 128, False), (1, 32, 16, True), (63, 8, 12, True),
                                             (1000, 36, 20, True), (0, 16, 16, True)])
def test_linear_weight_gradient_kernel(M, cin, cout, bias):
    """mssvt_linear_wgrad (split-K fp32 MFMA + ordered slab sum) against float64 dY^T X; twice: bit-identical."""
    from mssvt_amd import train_path
    g = torch.Generator().manual_seed(M + cin)
    x = torch.randn(M, cin, generator=g).to(DEV)
    w = torch.randn(cout, cin, generator=g).to(DEV).requires_grad_(True)
    b = torch.randn(cout, generator=g).to(DEV).requires_grad_(True) if bias else None
    dy = torch.randn(M, cout, generator=g).to(DEV)
    grads = []
    for _ in range(2):
        w.grad = None
        if b is not None:
            b.grad = None
        xin = x.clone().requires_grad_(True)
        y = train_path.linear((w, b), xin)
        assert y.grad_fn is not None and "Linear" in type(y.grad_fn).__name__
        y.backward(dy)
        grads.append((w.grad.clone(), None if b is None else b.grad.clone(), xin.grad.clone()))
    assert torch.equal(grads[0][0], grads[1][0])
    want = dy.double().t() @ x.double()
    scale = max(1.0, float(want.abs().max()))
    assert float((grads[0][0].double() - want).abs().max()) <= 2e-6 * scale * max(1.0, M ** 0.5 / 16)
    if b is not None:
        assert torch.equal(grads[0][1], grads[1][1])
        wb = dy.double().sum(0)
        assert float((grads[0][1].double() - wb).abs().max()) <= 2e-6 * max(1.0, float(wb.abs().max())) * max(1.0, M ** 0.5 / 16)
    torch.testing.assert_close(grads[0][2], dy @ w.detach(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("cg,hd,max_q,max_k", [(64, 16, 20, 32), (64, 16, 45, 100), (64, 16, 2, 17), (128, 16, 1, 32), (32, 8, 45, 64), (16, 4, 3, 5),
                                               (64, 32, 9, 12), (64, 64, 5, 7)])
def test_pair_attention_kernels_vs_float64_autograd(cg, hd, max_q, max_k):
    """mssvt_pair_attention_fwd / _bwd against a float64 torch restatement per window (windows with no query, with no
    key, with more queries than one register pass holds); backward twice: bit-identical."""
    from mssvt_amd import train_path
    heads = cg // hd
    g = torch.Generator().manual_seed(cg * 100 + hd)
    nw = 300
    q_cnt = torch.randint(0, max_q + 1, (nw,), generator=g)
    k_cnt = torch.randint(0, max_k + 1, (nw,), generator=g)
    q_cnt[:3], k_cnt[:3] = torch.tensor([0, max_q, max_q]), torch.tensor([max_k, 0, max_k])
    q_off, k_off = torch.cumsum(q_cnt, 0) - q_cnt, torch.cumsum(k_cnt, 0) - k_cnt
    R, Kn = int(q_cnt.sum()), int(k_cnt.sum())
    q0, kv0, dO = torch.randn(R, cg, generator=g), torch.randn(Kn, 2 * cg, generator=g), torch.randn(R, cg, generator=g)
    wins = {k: v.int().to(DEV) for k, v in dict(q_off=q_off, q_cnt=q_cnt, k_off=k_off, k_cnt=k_cnt).items()}
    res = []
    for _ in range(2):
        q, kv = q0.to(DEV).requires_grad_(True), kv0.to(DEV).requires_grad_(True)
        O = train_path.pair_attention(q, kv, wins, heads, hd)
        O.backward(dO.to(DEV))
        res.append((O.detach().clone(), q.grad.clone(), kv.grad.clone()))
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b)
    qd, kvd = q0.double().requires_grad_(True), kv0.double().requires_grad_(True)
    outs = []
    for w in range(nw):
        qi = qd[q_off[w]:q_off[w] + q_cnt[w]].view(-1, heads, hd)
        kw = kvd[k_off[w]:k_off[w] + k_cnt[w]]
        if k_cnt[w] == 0:
            outs.append(qi.reshape(-1, cg) * 0)
            continue
        k, v = kw[:, :cg].view(-1, heads, hd), kw[:, cg:].view(-1, heads, hd)
        p = torch.softmax(torch.einsum("ihd,jhd->hij", qi, k), dim=-1)
        outs.append(torch.einsum("hij,jhd->ihd", p, v).reshape(-1, cg))
    want = torch.cat(outs)
    want.backward(dO.double())
    for got, ref, what in ((res[0][0], want.detach(), "O"), (res[0][1], qd.grad, "dq"), (res[0][2], kvd.grad, "dkv")):
        err = float((got.cpu().double() - ref).abs().max())
        assert err <= 2e-5 * max(1.0, float(ref.abs().max())), (what, err)


@pytest.mark.parametrize("nnz,n_src,ragged,heavy", [(5000, 700, False, False), (20000, 300, True, True), (1, 1, False, False),
                                                    (0, 5, False, False), (70000, 100000, True, False)])
def test_csr_transpose_matches_a_stable_sort(nnz, n_src, ragged, heavy):
    """mssvt_csr_transpose against torch's stable sort: per source row the destinations in ascending entry order, the
    dropped source left out, the longest-list word, and the chunked sum of a long list against float64."""
    from mssvt_amd import train_path
    g = torch.Generator().manual_seed(nnz + n_src)
    idx = torch.randint(0, n_src, (nnz,), generator=g)
    if heavy:
        idx[::3] = 7  # a list of ~6.7k entries (voxel 0 of the reference's empty-pick quirk looks like this)
    drop = 2 if n_src > 2 else None
    if ragged:
        cnt = torch.randint(0, 4, (nnz,), generator=g)
        keep = int((torch.cumsum(cnt, 0) <= nnz).sum())
        cnt = cnt[:keep]
        cnt[-1] += nnz - int(cnt.sum())
        off = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(cnt, 0)]).int()
        dst_of = torch.repeat_interleave(torch.arange(cnt.numel()), cnt)
    else:
        off, dst_of = None, torch.arange(nnz)
    w = torch.rand(nnz, generator=g)
    csr = train_path.Csr(None if off is None else off.to(DEV), idx.int().to(DEV), w.to(DEV), n_src, drop_src=drop,
                         fwd_longest=None if off is None else 4)
    keep = torch.ones(nnz, dtype=torch.bool) if drop is None else idx != drop
    order = torch.sort(idx[keep], stable=True).indices
    want_idx, want_w = dst_of[keep][order], w[keep][order]
    per = torch.bincount(idx[keep], minlength=n_src)
    want_off = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(per, 0)])
    n_keep = int(keep.sum())
    assert torch.equal(csr.t_off.cpu().long(), want_off)
    assert torch.equal(csr.t_idx.cpu()[:n_keep].long(), want_idx)
    assert torch.equal(csr.t_w.cpu()[:n_keep], want_w)
    longest = int(per.max()) if nnz else 0
    assert int(csr.bwd.pending.item()) == (longest if longest > train_path.CHUNK else 0)
    if nnz:
        n_dst = dst_of.max().item() + 1 if nnz else 0
        grad = torch.randn(int(csr.n_dst), 16, generator=g)
        got = csr.bwd.sum(grad.to(DEV))
        assert (csr.bwd.heavy is not None) == (longest > train_path.CHUNK)
        ref = torch.zeros(n_src, 16, dtype=torch.float64).index_add_(0, idx[keep], (grad.double()[dst_of] * w.double()[:, None])[keep])
        assert float((got.cpu().double() - ref).abs().max()) < 1e-3
        assert torch.equal(got, csr.bwd.sum(grad.to(DEV)))


@pytest.mark.parametrize("N,C", [(74270, 128), (1000, 64), (7, 16), (33, 256), (5000, 32)])
def test_layer_norm_backward_kernel(N, C):
    """mssvt_layer_norm_backward against float64 autograd; twice: bit-identical."""
    from mssvt_amd import train_path
    g = torch.Generator().manual_seed(N + C)
    x0 = (torch.randn(N, C, generator=g) * 2 + 0.3)
    dy = torch.randn(N, C, generator=g)
    norm = torch.nn.LayerNorm(C).to(DEV)
    with torch.no_grad():
        norm.weight.copy_(torch.rand(C, generator=g) + 0.5)
        norm.bias.copy_(torch.randn(C, generator=g))
    res = []
    for _ in range(2):
        norm.weight.grad = norm.bias.grad = None
        x = x0.to(DEV).requires_grad_(True)
        y = train_path.layer_norm(norm, x)
        assert "LayerNorm" in type(y.grad_fn).__name__ and "Native" not in type(y.grad_fn).__name__
        y.backward(dy.to(DEV))
        res.append((y.detach().clone(), x.grad.clone(), norm.weight.grad.clone(), norm.bias.grad.clone()))
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b)
    xd = x0.double().requires_grad_(True)
    wd, bd = norm.weight.detach().cpu().double().requires_grad_(True), norm.bias.detach().cpu().double().requires_grad_(True)
    yd = torch.nn.functional.layer_norm(xd, (C,), wd, bd, norm.eps)
    yd.backward(dy.double())
    for got, ref, what in zip(res[0], (yd.detach(), xd.grad, wd.grad, bd.grad), ("y", "dx", "dw", "db")):
        err = float((got.cpu().double() - ref).abs().max())
        assert err <= 3e-6 * max(1.0, float(ref.abs().max())) * max(1.0, N ** 0.5 / 16), (what, err)


@pytest.mark.parametrize("C,dims,N,R,K", [(128, (64, 64), 5000, 3000, (9000, 12000)), (32, (16, 16), 300, 100, (700, 1)),
                                          (256, (128, 128), 2000, 1, (0, 5000)), (64, (32, 32), 50, 0, (64, 64))])
def test_token_kernels_match_the_autograd_composition(C, dims, N, R, K):
    """train_path.tokens (csrc/train_tok.hip: gather + positional embedding of the four token sets of a Block as one node)
    against the composition it replaces (gather_sum of a column slice + _pos6) in float64, values and all three gradients;
    two backward passes agree bit for bit."""
    from mssvt_amd import train_path
    g = torch.Generator().manual_seed(C + N)
    conv = torch.nn.Conv1d(6, C, 1).to(DEV)
    xhat = torch.randn(N, C, generator=g).to(DEV).requires_grad_(True)
    parts, c0 = [], 0
    sets = []
    for M in (R, R) + tuple(K):
        rows = torch.randint(0, N, (M,), generator=g).int().to(DEV)
        if M > 600:
            rows[::2] = 3  # a long list in the inverted index (cut into chunks)
        geo = torch.cat([torch.randn(M, 6, generator=g), torch.zeros(M, 2)], 1).to(DEV)
        sets.append((rows, geo, train_path.Csr.gather(rows, N)))
    for k, (rows, geo, csr) in enumerate(sets):
        g_i = k % 2
        c0 = sum(dims[:g_i])
        parts.append(dict(rows=rows, geo=geo, csr=csr, c0=c0, c1=c0 + dims[g_i]))
    ws = [torch.randn(p["rows"].numel(), p["c1"] - p["c0"], generator=g).to(DEV) for p in parts]

    def run():
        for t in (xhat, conv.weight, conv.bias):
            t.grad = None
        toks = train_path.tokens(xhat, conv, parts)
        sum((t * w).sum() for t, w in zip(toks, ws)).backward()
        return [t.detach() for t in toks], xhat.grad.clone(), conv.weight.grad.clone(), conv.bias.grad.clone()

    toks, gx, gw, gb = run()
    toks2, gx2, gw2, gb2 = run()
    assert all(torch.equal(a, b) for a, b in zip(toks, toks2)) and torch.equal(gx, gx2) and torch.equal(gw, gw2) and torch.equal(gb, gb2)
    xd = xhat.detach().double().requires_grad_(True)
    wd = conv.weight.detach().double().squeeze(-1).requires_grad_(True)
    bd = conv.bias.detach().double().requires_grad_(True)
    loss = 0
    for p, w, t in zip(parts, ws, toks):
        ref = xd[p["rows"].long()][:, p["c0"]:p["c1"]] + torch.relu(p["geo"][:, :6].double() @ wd[p["c0"]:p["c1"]].T + bd[p["c0"]:p["c1"]])
        if t.numel():
            assert float((t.double() - ref.detach()).abs().max()) <= 1e-5 * max(1.0, float(ref.detach().abs().max()))
        loss = loss + (ref * w.double()).sum()
    loss.backward()
    for got, want, what in ((gx, xd.grad, "d xhat"), (gw.squeeze(-1), wd.grad, "d weight"), (gb, bd.grad, "d bias")):
        err, scale = float((got.double() - want).abs().max()), max(1.0, float(want.abs().max()))
        assert err <= 2e-5 * scale * max(1.0, (max(K) ** 0.5) / 16), (what, err, scale)


def test_strided_segment_sum_writes_and_accumulates_into_a_column_range():
    from mssvt_amd import train_path
    g = torch.Generator().manual_seed(5)
    R, N, C, c0, cg = 9000, 400, 96, 32, 48
    idx = torch.randint(0, N, (R,), generator=g)
    idx[::2] = 11  # ~4.5k entries on one row: the chunked path
    csr = train_path.Csr.gather(idx.int().to(DEV), N)
    src = torch.randn(R, cg, generator=g).to(DEV)
    want = torch.zeros(N, cg, dtype=torch.float64).index_add_(0, idx, src.cpu().double())
    dst = torch.full((N, C), 7.0, device=DEV)
    train_path._sum_into(csr.bwd, src, dst, c0, False)
    assert csr.bwd.heavy is not None
    assert float((dst[:, c0:c0 + cg].cpu().double() - want).abs().max()) < 1e-3
    assert bool((dst[:, :c0] == 7.0).all()) and bool((dst[:, c0 + cg:] == 7.0).all())
    train_path._sum_into(csr.bwd, src, dst, c0, True)
    assert float((dst[:, c0:c0 + cg].cpu().double() - 2 * want).abs().max()) < 2e-3
    assert bool((dst[:, :c0] == 7.0).all()) and bool((dst[:, c0 + cg:] == 7.0).all())


@pytest.mark.parametrize("relu", [False, True])
def test_linear_with_the_relu_epilogue_matches_the_library_composition(relu):
    from mssvt_amd import train_path
    g = torch.Generator().manual_seed(3)
    lin = torch.nn.Linear(64, 128).to(DEV)
    x = torch.randn(5000, 64, generator=g).to(DEV)
    dy = torch.randn(5000, 128, generator=g).to(DEV)
    xr = x.clone().requires_grad_(True)
    out = train_path.linear(lin, xr, relu=relu)
    (out * dy).sum().backward()
    gw, gb = lin.weight.grad.clone(), lin.bias.grad.clone()
    lin.weight.grad = lin.bias.grad = None
    x2 = x.clone().requires_grad_(True)
    o2 = lin(x2)
    o2 = o2.relu() if relu else o2
    (o2 * dy).sum().backward()
    # (64 -> 128 runs on mssvt_linear_rows_h since round 5: the fp32 instruction's error class, not the library's bits;
    # an output within rounding of 0 may fall on either side of the relu: those elements' gradients are not compared)
    assert float((out.detach() - o2.detach()).abs().max()) <= 1e-5 * max(1.0, float(o2.detach().abs().max()))
    same = ((out.detach() > 0) == (o2.detach() > 0)).all(1)
    assert int((~same).sum()) <= 5
    assert float((xr.grad - x2.grad)[same].abs().max()) <= 1e-5 * max(1.0, float(x2.grad.abs().max()))
    assert float((gw - lin.weight.grad).abs().max()) <= 2e-4 * max(1.0, float(lin.weight.grad.abs().max()))
    assert float((gb - lin.bias.grad).abs().max()) <= 2e-4 * max(1.0, float(lin.bias.grad.abs().max()))


@pytest.mark.parametrize("M,K,N,relu", [(33001, 64, 64, False), (132017, 64, 64, True), (17, 64, 64, False), (1, 64, 64, False),
                                        (4097, 64, 128, True), (5000, 128, 64, False), (300, 128, 128, False)])
def test_linear_rows_kernel_forward_and_input_gradient(M, K, N, relu):
    """mssvt_linear_rows (the small weight matrices of a Block) against float64: y = x W^T + b with / without the relu
    epilogue, dx = dy W through the transposed read of the same weight; and through train_path.linear for the shape that
    runs on it, against the library composition."""
    from mssvt_amd import train_path
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g).to(DEV)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    y = train_path._linear_rows(x, w, False, b, relu, N)
    ref = x.double() @ w.double().T + b.double()
    if relu:
        ref = ref.clamp(min=0)
    assert float((y.double() - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))
    dy = torch.randn(M, N, generator=g).to(DEV)
    dx = train_path._linear_rows(dy, w, True, None, False, K)
    refx = dy.double() @ w.double()
    assert float((dx.double() - refx).abs().max()) <= 2e-5 * max(1.0, float(refx.abs().max()))
    if (K, N) in train_path.LINEAR_ROWS:
        lin = torch.nn.Linear(K, N).to(DEV)
        xr = x.clone().requires_grad_(True)
        out = train_path.linear(lin, xr, relu=relu, scale=0.25)
        (out * dy).sum().backward()
        gw, gb = lin.weight.grad.clone(), lin.bias.grad.clone()
        lin.weight.grad = lin.bias.grad = None
        x2 = x.clone().requires_grad_(True)
        o2 = lin(x2)
        o2 = (o2.relu() if relu else o2) * 0.25
        (o2 * dy).sum().backward()
        assert float((out.detach() - o2.detach()).abs().max()) <= 1e-4 * max(1.0, float(o2.detach().abs().max()))
        assert float((xr.grad - x2.grad).abs().max()) <= 1e-4 * max(1.0, float(x2.grad.abs().max()))
        tol = 2e-4 * max(1.0, M ** 0.5 / 16)
        assert float((gw - lin.weight.grad).abs().max()) <= tol * max(1.0, float(lin.weight.grad.abs().max()))
        assert float((gb - lin.bias.grad).abs().max()) <= tol * max(1.0, float(lin.bias.grad.abs().max()))


@pytest.mark.parametrize("M,K,N,relu", [(74270, 128, 256, True), (74270, 256, 128, False), (300011, 64, 128, False), (5000, 128, 64, False),
                                        (19307, 128, 128, True), (17, 128, 256, False), (1, 64, 64, False)])
def test_split_fp16_linear_rows_kernel_vs_float64_at_any_scale(M, K, N, relu):
    """mssvt_linear_rows_h (the large weight matrices of the training path on split-fp16 operands) against float64: forward
    with bias / relu / output scale and the transposed-weight form (dx = dy W) -- on rows whose scales span 2^-60 .. 2^60
    (gradients do not respect the fp16 range: the kernel normalises every row and the matrix by powers of two) and a weight
    matrix far outside it.  Error bound per ROW: relative to that row's own |x| |W| scale, the fp32 instruction's class."""
    from mssvt_amd import train_path
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g)
    x *= torch.exp2(torch.randint(-60, 61, (M, 1), generator=g).float())  # every row at its own scale
    x[M // 2] = 0.0  # an all-zero row stays zero
    for wscale in (1.0, 3.0e7, 2.0e-9):
        w = torch.randn(N, K, generator=g) / K ** 0.5 * wscale
        b = torch.randn(N, generator=g) * wscale
        xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
        y = train_path._linear_rows(xd, wd, False, bd, relu, N, 0.25, split16=True)
        ref = x.double() @ w.double().T + b.double()
        ref = (ref.clamp(min=0) if relu else ref) * 0.25
        row_scale = (x.double().abs().max(1, keepdim=True).values * w.double().abs().max() * K ** 0.5 + b.double().abs().max()).clamp(min=1e-300)
        assert float(((y.cpu().double() - ref).abs() / row_scale).max()) <= 2e-6, wscale
        assert bool((y[M // 2] == ((bd.clamp(min=0) if relu else bd) * 0.25)).all())
        dy = x[:, :N] if N <= K else torch.cat([x, x], 1)[:, :N]
        dy = dy.contiguous()
        dx = train_path._linear_rows(dy.to(DEV), wd, True, None, False, K, 1.0, split16=True)
        refx = dy.double() @ w.double()
        row_scale = (dy.double().abs().max(1, keepdim=True).values * w.double().abs().max() * N ** 0.5).clamp(min=1e-300)
        assert float(((dx.cpu().double() - refx).abs() / row_scale).max()) <= 2e-6, wscale
    if (K, N) in train_path.LINEAR_ROWS_H:  # through the autograd function, against the library composition
        lin = torch.nn.Linear(K, N).to(DEV)
        xs = torch.randn(M, K, generator=g).to(DEV)
        dys = torch.randn(M, N, generator=g).to(DEV)
        xr = xs.clone().requires_grad_(True)
        out = train_path.linear(lin, xr, relu=relu)
        (out * dys).sum().backward()
        o2 = xs.double() @ lin.weight.detach().double().T + lin.bias.detach().double()
        o2 = o2.relu() if relu else o2
        assert float((out.detach().double() - o2).abs().max()) <= 2e-5 * max(1.0, float(o2.abs().max()))
        # the relu's gradient mask is the KERNEL's output sign (an output within rounding of 0 may fall either side)
        gate = (out.detach() > 0).double() if relu else 1.0
        want = (dys.double() * gate) @ lin.weight.detach().double()
        assert float((xr.grad.double() - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))


def test_block_tail_in_one_launch_matches_the_composition_with_drop_path_active():
    """train() mode (DropPath rate 0.3 on the second Block, dropout as configured): the one-launch Block tail
    (interpolation + select + DropPath + residual, train_path._BlockTail) against the autograd composition under the same
    random stream -- output, input gradient and every parameter gradient."""
    from mssvt_amd import train_path
    C, B, H = 64, 2, 40009
    net = _net(C, _params(C, [4, 4], [8], (True, False)), H).train()
    assert any(getattr(b.drop_path, "drop_prob", 0.0) > 0 for b in net.backbone if hasattr(b, "drop_path"))
    vc, _, _ = synthetic.voxelize_numpy(synthetic.make_batch_points(4000, B, 17))
    coords = torch.from_numpy(vc).to(DEV)
    x = torch.randn(vc.shape[0], C, device=DEV)
    res = {}
    for fused_tail in (True, False):
        train_path.BLOCK_TAIL = fused_tail
        try:
            torch.manual_seed(123)
            out_shape = net(dict(voxel_features=x, voxel_coords=coords, batch_size=B))["encoded_spconv_tensor"].features.shape
            w = torch.randn(out_shape, generator=torch.Generator().manual_seed(7)).to(DEV)
            torch.manual_seed(123)
            res[fused_tail] = _grads(net, x, coords, B, w)
        finally:
            train_path.BLOCK_TAIL = True
    (o1, gx1, gp1), (o2, gx2, gp2) = res[True], res[False]
    for a, b, what in [(o1, o2, "output"), (gx1, gx2, "input gradient")] + [(gp1[k], gp2[k], k) for k in gp2]:
        scale = max(1.0, float(b.abs().max()))
        assert float((a - b).abs().max()) <= 1e-4 * scale, what


@pytest.mark.parametrize("cap,nw,K", [(5000, 3777, 32), (300, 300, 64), (64, 1, 8), (1000, 0, 16)])
def test_key_set_kernels_match_the_framework_composition(cap, nw, K):
    """mssvt_train_key_counts / _compact (the compact key sets of a plan) against mask + nonzero + gathers."""
    import ctypes
    from mssvt_amd import _lib
    g = torch.Generator().manual_seed(cap + K)
    kmeta = torch.randn(cap, K, 4, generator=g)
    rows = torch.randint(-1, 900, (cap, K), generator=g).int()
    rows[torch.rand(cap, K, generator=g) < 0.4] = -1
    kmeta[..., 3] = rows.view(torch.float32)
    wcentre = torch.randn(cap, 4, generator=g)
    kmeta, wcentre = kmeta.to(DEV).contiguous(), wcentre.to(DEV)
    num_wins = torch.tensor([nw], dtype=torch.int32, device=DEV)
    cnt = torch.full((cap,), -7, dtype=torch.int32, device=DEV)
    total = torch.zeros(1, dtype=torch.int32, device=DEV)
    i = ctypes.c_int
    _lib.call("mssvt_train_key_counts", i(cap), i(K), _lib.ptr(num_wins), _lib.ptr(kmeta), _lib.ptr(cnt), _lib.ptr(total), _lib.stream())
    valid = (rows.to(DEV) >= 0) & (torch.arange(cap, device=DEV).unsqueeze(1) < nw)
    assert torch.equal(cnt.long(), valid.sum(1)) and int(total.item()) == int(valid.sum())
    n = int(total.item())
    off = (torch.cumsum(cnt[:nw], 0, dtype=torch.int32) - cnt[:nw]).contiguous()
    k_rows = torch.empty(n, dtype=torch.int32, device=DEV)
    k_win = torch.empty(n, dtype=torch.int32, device=DEV)
    k_geo = torch.empty((n, 8), dtype=torch.float32, device=DEV)
    if nw and n:
        _lib.call("mssvt_train_key_compact", i(nw), i(K), _lib.ptr(kmeta), _lib.ptr(wcentre), _lib.ptr(off), _lib.ptr(k_rows),
                  _lib.ptr(k_win), _lib.ptr(k_geo), _lib.stream())
    flat = torch.nonzero(valid.reshape(-1), as_tuple=True)[0]
    win = flat // K
    assert torch.equal(k_rows.long(), rows.to(DEV).reshape(-1)[flat].long()) and torch.equal(k_win.long(), win)
    want = torch.cat([kmeta[..., :3].reshape(-1, 3)[flat], wcentre[win, :3], torch.zeros(n, 2, device=DEV)], 1)
    assert torch.equal(k_geo, want)


@pytest.mark.parametrize("nw,ns,N", [(3000, 32, 9000), (50, 64, 400), (1, 8, 5)])
def test_pair_set_kernels_match_the_framework_composition(nw, ns, N):
    """mssvt_train_list_counts / _pairs_compact (the CompressBlock's (window, voxel) pairs and their geometry) against
    mask + nonzero + gathers + metric_centres."""
    import ctypes
    from mssvt_amd import _lib, fused
    from mssvt_amd.mssvt_backbone import metric_centres
    g = torch.Generator().manual_seed(nw + ns)
    k_ind = torch.randint(0, 3, (nw, ns), generator=g).int()
    k_ind[torch.rand(nw, ns, generator=g) < 0.5] = -1
    vstart = torch.randint(0, N - 3, (nw,), generator=g).int()
    indices = torch.randint(0, 40, (N, 4), generator=g).int()
    win_ind = torch.randint(0, 12, (nw, 4), generator=g).int()
    vs, rng, wsz = [0.32, 0.32, 0.1875], [-74.88, -74.88, -2.0, 74.88, 74.88, 4.0], [0.32, 0.32, 6.0]
    k_ind, vstart, indices, win_ind = (t.to(DEV).contiguous() for t in (k_ind, vstart, indices, win_ind))
    i = ctypes.c_int
    cnt = torch.empty(nw, dtype=torch.int32, device=DEV)
    total = torch.zeros(1, dtype=torch.int32, device=DEV)
    _lib.call("mssvt_train_list_counts", i(nw), i(ns), _lib.ptr(k_ind), _lib.ptr(cnt), _lib.ptr(total), _lib.stream())
    valid = k_ind >= 0
    assert torch.equal(cnt.long(), valid.sum(1)) and int(total.item()) == int(valid.sum())
    P = int(total.item())
    off = (torch.cumsum(cnt, 0, dtype=torch.int32) - cnt).contiguous()
    pair_vox = torch.empty(P, dtype=torch.int32, device=DEV)
    pair_win = torch.empty(P, dtype=torch.int32, device=DEV)
    geo = torch.empty((P, 8), dtype=torch.float32, device=DEV)
    if P:
        _lib.call("mssvt_train_pairs_compact", i(nw), i(ns), _lib.ptr(k_ind), _lib.ptr(vstart), _lib.ptr(off), _lib.ptr(indices),
                  _lib.ptr(win_ind), fused._f3(vs), fused._f3(rng[:3]), fused._f3(wsz), _lib.ptr(pair_vox), _lib.ptr(pair_win),
                  _lib.ptr(geo), _lib.stream())
    flat = torch.nonzero(valid.reshape(-1), as_tuple=True)[0]
    win = flat // ns
    vox = k_ind.reshape(-1)[flat].long() + vstart.long()[win]
    assert torch.equal(pair_vox.long(), vox) and torch.equal(pair_win.long(), win)
    centre = metric_centres(win_ind, rng, wsz)[win]
    rel = metric_centres(indices, rng, vs)[vox] - centre
    assert torch.equal(geo[:, :3], rel) and torch.equal(geo[:, 3:6], centre) and bool((geo[:, 6:] == 0).all())
