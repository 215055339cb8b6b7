"""CPU oracle for the Python-level orchestration of the MsSVT backbone.

TEST INFRASTRUCTURE ONLY (see oracle/mssvt_oracle.c header).  A functional
restatement -- weights in a flat ``{name: ndarray}`` dict, numpy for index work,
torch-CPU fp32 for the dense math -- of

* ``MixedScaleAttention.forward``            pcdet/models/model_utils/mssvt_utils.py:88-157
* ``get_vox_query_table``                    pcdet/models/backbones_3d/mssvt_backbone.py:73-122
* ``MixedScaleSparseTransformerBlock.forward``          ...mssvt_backbone.py:201-346
* ``MixedScaleSparseTransformerCompressBlock.forward``  ...mssvt_backbone.py:351-398
* ``MixedScaleSparseTransformer.forward``               ...mssvt_backbone.py:450-472
* ``SparseTensor.dense`` / ``scatter_nd``    ...mssvt_utils.py:6-19,50-62

with the C oracle (oracle/cref.py) for the CUDA kernels.  PINNED: every function
here is asserted equal to the outputs of the reference's own Python
(tests/golden/*.npz, produced by oracle/gen_golden.py) in tests/test_oracle_golden.py.
Unlike oracle/ref_import.py this file needs no reference tree, so it is the
checker that travels to the GPU box.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import cref

MAX_NUM_WINS = 90000  # mssvt_backbone.py:56


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


# --------------------------------------------------------------------------
# query tables (R3)
# --------------------------------------------------------------------------

def vox_query_table(win1_size, win2_size=None):
    """mssvt_backbone.py:73-122.  Ties inside one Chebyshev shell are ordered by a
    STABLE sort over the x-major meshgrid enumeration (the reference's device
    ``torch.sort`` is unstable there) -- the tables are therefore an
    explicit input of every parity test."""
    size = win2_size if win2_size is not None else win1_size
    xs, ys, zs = np.meshgrid(np.arange(size[0]), np.arange(size[1]), np.arange(size[2]), indexing="ij")
    xyz = np.stack([xs, ys, zs], axis=-1).reshape(-1, 3) - (np.array(size) // 2)[None, :]
    dist = np.abs(xyz).max(axis=-1)
    xyz = xyz[np.argsort(dist, kind="stable")]
    if win2_size is None:
        return {"win1": xyz.astype(np.int32)}, None, None
    assert all((win2_size[i] - win1_size[i]) % 2 == 0 for i in range(3))  # :75
    off = [1 - win1_size[i] % 2 for i in range(3)]
    m1 = np.ones(xyz.shape[0], bool)
    for i in range(3):
        m1 &= (xyz[:, i] <= win1_size[i] // 2 + off[i]) & (xyz[:, i] >= -(win1_size[i] // 2))
    w1, w2o = xyz[m1], xyz[~m1]
    odd = (w1[:, 0] % 2 == 1) & (w1[:, 1] % 2 == 1)  # python-style modulo: -1 is odd
    even = (w1[:, 0] % 2 == 0) & (w1[:, 1] % 2 == 0)
    tab = {"odd": w1[odd].astype(np.int32), "even": w1[even].astype(np.int32),
           "win1": w1[~(odd | even)].astype(np.int32), "win2": w2o.astype(np.int32)}
    return tab, int(odd.sum()), int(even.sum())


# --------------------------------------------------------------------------
# attention (R10)
# --------------------------------------------------------------------------

def mixed_scale_attention(sd, prefix, embed_dim, num_heads, query, keys, query_mask=None,
                          key_masks=None, batch_first=False):
    """mssvt_utils.py:88-157.  query (b,nq,C) / keys (b,G*nk,C) if batch_first else
    (nq,b,C) / (G*nk,b,C); returns the same layout as the query."""
    query, keys = _t(query).float(), _t(keys).float()
    if not batch_first:
        query, keys = query.transpose(1, 0), keys.transpose(1, 0)
    b, nq, _ = query.shape
    G = len(num_heads)
    nk = keys.shape[1] // G
    hd = embed_dim // sum(num_heads)
    scale = hd ** -0.5
    outs = []
    c0 = 0
    for g in range(G):
        c1 = c0 + hd * num_heads[g]
        W = lambda n: _t(sd["%s%s.%d.weight" % (prefix, n, g)])  # noqa: E731
        Bv = lambda n: _t(sd["%s%s.%d.bias" % (prefix, n, g)])  # noqa: E731
        q = F.linear(query[:, :, c0:c1], W("to_qs"), Bv("to_qs"))
        q = q.reshape(b, nq, num_heads[g], hd).permute(0, 2, 1, 3)
        kv = F.linear(keys[:, g * nk:(g + 1) * nk, c0:c1], W("to_kvs"), Bv("to_kvs"))
        kv = kv.reshape(b, nk, 2, num_heads[g], hd).permute(2, 0, 3, 1, 4)
        k, v = kv[0], kv[1]
        attn = (q * scale) @ k.transpose(-2, -1)
        if key_masks is not None:  # softmax only happens on this branch (:129-134)
            km = _t(key_masks)[:, g * nk:(g + 1) * nk]
            add = torch.where(km != 0, torch.tensor(-100.0), torch.tensor(0.0)).view(b, 1, 1, nk)
            attn = torch.softmax(attn + add, dim=-1)
        x = (attn @ v).transpose(1, 2).reshape(b, nq, -1)
        outs.append(F.linear(x, W("projs"), Bv("projs")))
        c0 = c1
    out = torch.cat(outs, dim=-1)
    if query_mask is not None:
        out = out * (~_t(query_mask)).unsqueeze(-1).float()
    if not batch_first:
        out = out.transpose(1, 0)
    return out.contiguous().numpy()


# --------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------

def with_coords(indices, point_cloud_range, voxel_size):
    """mssvt_backbone.py:132-137: ((idx[:, [3,2,1]] + 0.5) * voxel_size + min_range), fp32,
    one rounding per op."""
    vs = torch.tensor(voxel_size).unsqueeze(0)  # default dtype: fp32 (python floats)
    mn = torch.tensor(point_cloud_range[0:3]).unsqueeze(0)
    return ((_t(indices)[:, [3, 2, 1]].float() + 0.5) * vs + mn).numpy()


def _pos_proj(sd, prefix, x):
    """pos_proj: Conv1d(6,C,1)+ReLU (+Conv1d(C,C,1)+ReLU for single-window blocks),
    mssvt_backbone.py:43-54.  x (nw,6,n)."""
    x = _t(x)
    y = F.relu(F.conv1d(x, _t(sd[prefix + "pos_proj.0.weight"]), _t(sd[prefix + "pos_proj.0.bias"])))
    if (prefix + "pos_proj.2.weight") in sd:
        y = F.relu(F.conv1d(y, _t(sd[prefix + "pos_proj.2.weight"]), _t(sd[prefix + "pos_proj.2.bias"])))
    return y.numpy()


def _ffn_tail(sd, prefix, new_features, residual_inside=True):
    """norm2 -> linear1 -> ReLU -> linear2 -> +x -> optional out_linear
    (mssvt_backbone.py:339-342, :384-387)."""
    x = _t(new_features)
    C = x.shape[1]
    h = F.layer_norm(x, (C,), _t(sd[prefix + "norm2.weight"]), _t(sd[prefix + "norm2.bias"]))
    h = F.linear(F.relu(F.linear(h, _t(sd[prefix + "linear1.weight"]), _t(sd[prefix + "linear1.bias"]))),
                 _t(sd[prefix + "linear2.weight"]), _t(sd[prefix + "linear2.bias"]))
    x = x + h
    if (prefix + "out_linear.weight") in sd:
        x = F.linear(x, _t(sd[prefix + "out_linear.weight"]), _t(sd[prefix + "out_linear.bias"]))
    return x.numpy()


class SparseState:
    """The fields of the reference ``SparseTensor`` the path reads/writes (mssvt_utils.py:21-31)."""

    def __init__(self, features, indices, spatial_shape, voxel_size, point_cloud_range,
                 batch_size, hash_size, map_table=None):
        self.features = np.ascontiguousarray(features, np.float32)
        self.indices = np.ascontiguousarray(indices, np.int32)
        self.spatial_shape = [int(v) for v in spatial_shape]
        self.voxel_size = [float(v) for v in voxel_size]
        self.point_cloud_range = [float(v) for v in point_cloud_range]
        self.batch_size = int(batch_size)
        self.hash_size = int(hash_size)
        if map_table is None:  # build_map_table, mssvt_utils.py:33-48
            map_table = cref.build_hash_table(self.batch_size, self.hash_size, self.spatial_shape,
                                              self.indices, cref.bs_cnt(self.indices, self.batch_size))
        self.map_table = map_table

    def dense(self):
        """mssvt_utils.py:50-62 -> (B, C, Z, Y, X)."""
        B, C = self.batch_size, self.features.shape[1]
        X, Y, Z = self.spatial_shape
        ret = np.zeros((B, Z, Y, X, C), np.float32)
        i = self.indices.astype(np.int64)
        ret[i[:, 0], i[:, 1], i[:, 2], i[:, 3]] = self.features
        return np.ascontiguousarray(ret.transpose(0, 4, 1, 2, 3))


# --------------------------------------------------------------------------
# Block (R2-R12)
# --------------------------------------------------------------------------

def block_forward(sd, prefix, sp, window_size, num_heads, max_num_win1, max_num_win2, cbs_pattern,
                  key_num_sample=32, use_feature_interpolation=True, tables=None, record=None):
    """MixedScaleSparseTransformerBlock.forward (eval mode: dropout/drop-path are identity)."""
    win1, win2 = window_size
    if tables is None:
        tables, n_odd, n_even = vox_query_table(win1, win2)
    else:
        n_odd, n_even = tables["odd"].shape[0], tables["even"].shape[0]
    if max_num_win1 is None:
        max_num_win1 = win1[0] * win1[1] * win1[2]
    if max_num_win2 is None:
        max_num_win2 = win2[0] * win2[1] * win2[2]
    rec = record if record is not None else {}
    B, C = sp.batch_size, sp.features.shape[1]
    x_in = sp.features
    x = F.layer_norm(_t(x_in), (C,), _t(sd[prefix + "norm1.weight"]), _t(sd[prefix + "norm1.bias"])).numpy()

    new_shape = [sp.spatial_shape[i] // win1[i] for i in range(3)]  # :140-143
    win_ind, _ = cref.get_non_empty_window_center(win1, MAX_NUM_WINS, B, sp.hash_size, new_shape, sp.indices)
    win_size_m = [sp.voxel_size[i] * win1[i] for i in range(3)]  # :214-215
    (ind_odd, ind_even, ind_w1, ind_w2, c_odd, c_even, c_w1, c_w2) = cref.gather_two_window_voxels(
        sp.spatial_shape, win1, n_odd, n_even, max_num_win1, max_num_win2,
        tables["odd"], tables["even"], tables["win1"], tables["win2"], win_ind, sp.map_table)
    v_cnt = cref.bs_cnt(sp.indices, B)
    k_cnt = cref.bs_cnt(win_ind, B)
    q_ind = {0: ind_even, 1: ind_odd, 2: ind_w1}[cbs_pattern]  # :220-232
    q_mask = q_ind < 0
    nq = q_ind.shape[1]
    n1 = ind_w1.shape[1]

    # FPS keys + masks, :247-258
    fps1 = cref.farthest_point_sample(c_w1.astype(np.float32), key_num_sample)
    fps2 = cref.farthest_point_sample(c_w2.astype(np.float32), key_num_sample)
    m1 = fps1 == 0
    m1[:, 0] = False
    m2 = fps2 == 0
    m2[:, 0] = False
    # gather_operation on the indices round-tripped through fp32, then ``(x + 0.1).int()``
    # (:253-256).  QUIRK: .int() truncates toward zero, so a picked padding slot (-1) becomes
    # -0.9 -> 0, i.e. voxel 0 of the sample, and ``k_ind < 0`` below never fires.
    def _roundtrip(ind, fps):
        g = np.take_along_axis(ind, fps.astype(np.int64), axis=1).astype(np.float32)
        return (g + np.float32(0.1)).astype(np.int32)  # C-style truncation, like Tensor.int()

    k_ind1 = _roundtrip(ind_w1, fps1)
    k_ind2 = _roundtrip(ind_w2, fps2)
    m1 |= k_ind1 < 0
    m2 |= k_ind2 < 0
    rec.update(win_ind=win_ind, ind_odd=ind_odd, ind_even=ind_even, ind_win1=ind_w1, ind_win2=ind_w2,
               coord_odd=c_odd, coord_even=c_even, coord_win1=c_w1, coord_win2=c_w2,
               fps1=fps1, fps2=fps2, k_ind1=k_ind1, k_ind2=k_ind2, k_mask1=m1, k_mask2=m2)

    q_fea = cref.grouping_operation(x, v_cnt, q_ind, k_cnt)  # (nw,C,nq), :260-262
    k_fea1 = cref.grouping_operation(x, v_cnt, k_ind1, k_cnt)
    k_fea2 = cref.grouping_operation(x, v_cnt, k_ind2, k_cnt)
    vcoord = with_coords(sp.indices, sp.point_cloud_range, sp.voxel_size)  # :264
    q_coord = cref.grouping_operation(vcoord, v_cnt, q_ind, k_cnt)  # (nw,3,nq)
    w1_coord = cref.grouping_operation(vcoord, v_cnt, ind_w1, k_cnt)
    k_coord1 = cref.grouping_operation(vcoord, v_cnt, k_ind1, k_cnt)
    k_coord2 = cref.grouping_operation(vcoord, v_cnt, k_ind2, k_cnt)
    centre = with_coords(win_ind, sp.point_cloud_range, win_size_m)[:, :, None]  # (nw,3,1), :269

    k_rel1 = (k_coord1 - centre) * (~m1)[:, None, :]  # :271-276
    k_rel2 = (k_coord2 - centre) * (~m2)[:, None, :]
    q_rel = (q_coord - centre) * (~q_mask)[:, None, :]
    q_pos = _pos_proj(sd, prefix, np.concatenate([q_rel, np.broadcast_to(centre, q_rel.shape)], axis=1))
    k_rel = np.concatenate([k_rel1, k_rel2], axis=-1)
    k_pos = _pos_proj(sd, prefix, np.concatenate([k_rel, np.broadcast_to(centre, k_rel.shape)], axis=1))
    q_in = np.ascontiguousarray((q_fea + q_pos).transpose(0, 2, 1))  # (nw,nq,C)
    k_in = np.ascontiguousarray((np.concatenate([k_fea1, k_fea2], axis=-1) + k_pos).transpose(0, 2, 1))
    k_mask = np.concatenate([m1, m2], axis=-1)

    attn = mixed_scale_attention(sd, prefix + "ms_attn.", C, num_heads, q_in, k_in,
                                 query_mask=q_mask, key_masks=k_mask, batch_first=True)  # (nw,nq,C)
    rec.update(attn=attn)

    if use_feature_interpolation:  # :300-310
        known = np.ascontiguousarray(q_coord.transpose(0, 2, 1))  # (nw,nq,3), padded slots at 0
        unknown = np.ascontiguousarray(w1_coord.transpose(0, 2, 1))  # (nw,n1,3)
        dist, idx = cref.three_nn(unknown, known)
        dist = np.maximum(dist, np.float32(1e-10))
        w = (np.float32(1.0) / dist).astype(np.float32)
        w = w / w.sum(-1, keepdims=True)
        grouped = cref.group_points(np.ascontiguousarray(attn.transpose(0, 2, 1)), idx)  # (nw,C,n1,3)
        w1_fea = (_t(grouped) * _t(w).unsqueeze(1)).sum(-1).numpy()  # (nw,C,n1)
        w1_fea = np.ascontiguousarray(w1_fea.transpose(0, 2, 1)).reshape(-1, C)
        rec.update(nn_idx=idx, nn_dist=dist)
    attn_flat = attn.reshape(-1, C)

    # per-sample scatter, :313-334.  index -1 addresses the appended padding row.
    feats = x_in.copy()
    vs = ks = 0
    for b in range(B):
        nv, nk = int(v_cnt[b]), int(k_cnt[b])
        sel = np.concatenate([x_in[vs:vs + nv], np.zeros((1, C), np.float32)], axis=0)
        if use_feature_interpolation:
            sel[ind_w1[ks:ks + nk].reshape(-1).astype(np.int64)] = w1_fea[ks * n1:(ks + nk) * n1]
        else:
            sel[q_ind[ks:ks + nk].reshape(-1).astype(np.int64)] = attn_flat[ks * nq:(ks + nk) * nq]
        feats[vs:vs + nv] = sel[:-1]
        vs += nv
        ks += nk
    new = feats + x_in  # :338 (drop_path identity in eval)
    sp.features = _ffn_tail(sd, prefix, new)
    return sp


# --------------------------------------------------------------------------
# CompressBlock (R13)
# --------------------------------------------------------------------------

def compress_forward(sd, prefix, sp, window_size, num_heads, max_num_win1, tables=None, record=None):
    """MixedScaleSparseTransformerCompressBlock.forward, mssvt_backbone.py:351-398."""
    win1 = window_size[0]
    if tables is None:
        tables, _, _ = vox_query_table(win1, None)
    if max_num_win1 is None:
        max_num_win1 = win1[0] * win1[1] * win1[2]
    rec = record if record is not None else {}
    B, C = sp.batch_size, sp.features.shape[1]
    x = F.layer_norm(_t(sp.features), (C,), _t(sd[prefix + "norm1.weight"]), _t(sd[prefix + "norm1.bias"])).numpy()
    new_shape = [sp.spatial_shape[i] // win1[i] for i in range(3)]
    win_ind, new_table = cref.get_non_empty_window_center(win1, MAX_NUM_WINS, B, sp.hash_size, new_shape, sp.indices)
    win_size_m = [sp.voxel_size[i] * win1[i] for i in range(3)]
    k_ind, k_gc = cref.gather_one_window_voxels(sp.spatial_shape, win1, max_num_win1, tables["win1"], win_ind, sp.map_table)
    k_mask = k_ind < 0
    v_cnt = cref.bs_cnt(sp.indices, B)
    k_cnt = cref.bs_cnt(win_ind, B)
    rec.update(win_ind=win_ind, ind_win1=k_ind, coord_win1=k_gc, win_table=new_table)

    k_fea = cref.grouping_operation(x, v_cnt, k_ind, k_cnt)  # (nw,C,ns)
    vcoord = with_coords(sp.indices, sp.point_cloud_range, sp.voxel_size)
    k_coord = cref.grouping_operation(vcoord, v_cnt, k_ind, k_cnt)  # (nw,3,ns); padded slots = 0
    q_coord = with_coords(win_ind, sp.point_cloud_range, win_size_m)[:, :, None]
    q_fea = k_fea.max(axis=-1)[None]  # (1,nw,C): zeros of padded slots take part (:370)
    k_rel = k_coord - q_coord  # NOT masked here (:372)
    k_pos = _pos_proj(sd, prefix, np.concatenate([k_rel, np.broadcast_to(q_coord, k_rel.shape)], axis=1))
    k_in = np.ascontiguousarray((k_fea + k_pos).transpose(2, 0, 1))  # (ns,nw,C)
    attn = mixed_scale_attention(sd, prefix + "ms_attn.", C, num_heads, q_fea, k_in, key_masks=k_mask)
    new = attn[0]
    rec.update(attn=new)
    # FFN without the residual to the block input (:383-387)
    sp.features = _ffn_tail(sd, prefix, new)
    sp.indices = win_ind
    sp.spatial_shape = new_shape
    sp.voxel_size = win_size_m
    sp.map_table = new_table
    return sp


# --------------------------------------------------------------------------
# backbone
# --------------------------------------------------------------------------

def backbone_forward(sd, params, voxel_features, voxel_coords, batch_size, grid_size, voxel_size,
                     point_cloud_range, hash_size, tables_per_block=None):
    """MixedScaleSparseTransformer.forward, mssvt_backbone.py:450-472 (state-dict prefix
    ``backbone.{i}.``)."""
    sp = SparseState(voxel_features, np.asarray(voxel_coords).astype(np.int32), grid_size, voxel_size,
                     point_cloud_range, batch_size, hash_size)
    for i, p in enumerate(params):
        prefix = "backbone.%d." % i
        tabs = None if tables_per_block is None else tables_per_block[i]
        if p["name"] == "MixedScaleSparseTransformerBlock":
            sp = block_forward(sd, prefix, sp, p["window_size"], p["num_heads"], p.get("max_num_win1"),
                               p.get("max_num_win2"), p["cbs_pattern"], p.get("key_num_sample", 32),
                               p.get("use_feature_interpolation", True), tables=tabs)
        elif p["name"] == "MixedScaleSparseTransformerCompressBlock":
            sp = compress_forward(sd, prefix, sp, p["window_size"], p["num_heads"], p.get("max_num_win1"),
                                  tables=tabs)
        else:
            raise NotImplementedError(p["name"])
    return sp
