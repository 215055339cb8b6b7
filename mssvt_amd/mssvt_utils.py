"""``SparseTensor``, ``MixedScaleAttention`` and ``scatter_nd`` for the MI355X MsSVT path.

Interface-compatible with the reference's ``pcdet/models/model_utils/mssvt_utils.py``
(``scatter_nd`` :6, ``SparseTensor`` :21, ``MixedScaleAttention`` :65): same class
names, constructor arguments, attributes and state-dict keys
(``to_qs.{g}``, ``to_kvs.{g}``, ``projs.{g}``), so reference checkpoints load and
``HeightCompression`` can consume the result (``.features``, ``.indices``,
``.spatial_shape``, ``.batch_size``, ``.dense()``).

``MixedScaleAttention.forward`` here is the operator-level path (plain torch ops on
the GPU: rocBLAS batched GEMMs); the module fast path runs the same arithmetic in
the fused HIP window kernel (``mssvt_amd/fused.py``) straight from the weights.
"""
import ctypes

import torch
from torch import nn

from . import mssvt_ops


def scatter_nd(indices, updates, shape):
    """Dense tensor of ``shape`` with ``updates`` written at integer ``indices``
    (last dim of ``indices`` addresses the leading dims).  ref: mssvt_utils.py:6-19."""
    out = torch.zeros(*shape, dtype=updates.dtype, device=updates.device)
    nd = indices.shape[-1]
    flat = indices.reshape(-1, nd)
    out[tuple(flat[:, i] for i in range(nd))] = updates.reshape(-1, *shape[nd:])
    return out


@torch.no_grad()
def batch_counts(indices, batch_size):
    """Rows per sample, (B,) int32 on the device, no host sync (the reference loops over
    samples with ``.item()``: mssvt_utils.py:35-37, mssvt_backbone.py:124-130)."""
    if indices.is_cuda and indices.dtype == torch.int32 and indices.is_contiguous() and indices.dim() == 2 \
            and indices.shape[1] == 4:
        from . import _lib
        out = torch.empty(batch_size, dtype=torch.int32, device=indices.device)
        _lib.call("mssvt_batch_counts", _lib.ptr(indices), ctypes.c_int(indices.shape[0]),
                  ctypes.c_int(int(batch_size)), _lib.ptr(out), _lib.stream())
        return out
    return torch.bincount(indices[:, 0].long(), minlength=batch_size)[:batch_size].to(torch.int32)


class SparseTensor(object):
    """Sparse voxel set: ``features (N,C)``, ``indices (N,4) int32 [b,z,y,x]`` (samples
    contiguous), ``spatial_shape [x,y,z]``, per-sample hash ``map_table (B,H,2)``.
    ref: mssvt_utils.py:21-62."""

    def __init__(self, features, indices, spatial_shape, voxel_size, point_cloud_range, batch_size,
                 hash_size, map_table=None, gather_dict=None, lazy_map_table=False):
        self.features = features
        self.indices = indices
        self.spatial_shape = spatial_shape
        self.batch_size = batch_size
        self.voxel_size = voxel_size
        self.point_cloud_range = point_cloud_range
        self.hash_size = hash_size
        self.gather_dict = gather_dict
        self.v_bs_cnt = None
        self._map_table = map_table
        if map_table is None and not lazy_map_table:
            self._map_table = self.build_map_table()
        self._cnt_of = self.indices if self.v_bs_cnt is not None else None

    @property
    def map_table(self):
        """Per-sample hash table key -> voxel index (ref build_map_table, mssvt_utils.py:31-48).  With
        ``lazy_map_table`` it is built on first use: the fused path of a (b,x,y,z)-sorted voxel list resolves cells
        through occupancy columns (csrc/level_sorted.hip) and never reads it."""
        if self._map_table is None:
            self._map_table = self.build_map_table()
        return self._map_table

    @map_table.setter
    def map_table(self, value):
        self._map_table = value

    @torch.no_grad()
    def build_map_table(self):
        cnt = self.v_bs_cnt
        if cnt is None or getattr(self, "_cnt_of", None) is not self.indices:
            cnt = self.v_bs_cnt = batch_counts(self.indices, self.batch_size)  # kept: the plans need it too
            self._cnt_of = self.indices
        table = mssvt_ops.build_hash_table(self.batch_size, self.hash_size, self.spatial_shape, self.indices, cnt)
        self.map_status = getattr(mssvt_ops.build_hash_table, "last_status", None)  # device status word of this table
        return table

    def dense(self, channels_first=True):
        """(B, C, Z, Y, X) (or (B, Z, Y, X, C)) dense grid.  ref: mssvt_utils.py:50-62."""
        zyx = list(self.spatial_shape[::-1])
        shape = [self.batch_size] + zyx + [self.features.shape[1]]
        f = self.features
        if f.shape[0] == 0:
            out_shape = [shape[0], shape[-1]] + zyx if channels_first else shape
            return torch.zeros(out_shape, dtype=f.dtype, device=f.device)
        if (channels_first and f.is_cuda and f.dtype == torch.float32 and not (torch.is_grad_enabled() and f.requires_grad)
                and self.map_table is not None and self.indices.dtype == torch.int32):
            # one gather pass through the hash table instead of zero fill + scatter + permute copy
            from . import _lib
            ci = ctypes.c_int
            X, Y, Z = (int(v) for v in self.spatial_shape)
            out = torch.empty([self.batch_size, f.shape[1]] + zyx, dtype=torch.float32, device=f.device)
            cnt = getattr(self, "v_bs_cnt", None)
            if cnt is None or getattr(self, "_cnt_of", None) is not self.indices:
                cnt = batch_counts(self.indices.contiguous(), self.batch_size)
            _lib.call("mssvt_dense_bev", _lib.ptr(f.contiguous()), ci(f.shape[1]), _lib.ptr(self.map_table),
                      ci(int(self.hash_size)), _lib.ptr(cnt), ci(int(self.batch_size)), ci(X), ci(Y), ci(Z),
                      _lib.ptr(out), _lib.stream())
            return out
        res = scatter_nd(self.indices.to(self.features.device).long(), self.features, shape)
        if not channels_first:
            return res
        nd = len(zyx)
        return res.permute(0, nd + 1, *range(1, nd + 1)).contiguous()


class MixedScaleAttention(nn.Module):
    """Grouped multi-head attention: head-group ``g`` owns a channel slice and attends
    to the ``g``-th chunk of the keys (group 0 -> win1 keys, group 1 -> win2 keys).
    ref: mssvt_utils.py:65-157."""

    def __init__(self, embed_dim, num_heads, dropout=0.):
        super().__init__()
        self.embed_dim = embed_dim
        self.num_heads = list(num_heads)
        self.num_head_groups = len(self.num_heads)
        self.tot_num_heads = sum(self.num_heads)
        assert embed_dim % self.tot_num_heads == 0
        self.per_head_dim = embed_dim // self.tot_num_heads
        self.scale_dims = [self.per_head_dim * h for h in self.num_heads]
        self.group_c_idx = [sum(self.scale_dims[:i + 1]) for i in range(self.num_head_groups)]
        self.to_qs = nn.ModuleList([nn.Linear(d, d) for d in self.scale_dims])
        self.to_kvs = nn.ModuleList([nn.Linear(d, 2 * d) for d in self.scale_dims])
        self.projs = nn.ModuleList([nn.Linear(d, d) for d in self.scale_dims])
        self.scale = self.per_head_dim ** -0.5
        self.attn_drop = nn.Dropout(dropout)
        self.proj_drop = nn.Dropout(dropout)

    def forward(self, query, keys, batch_first=False, pos_emb_mat=None, relative_coords=None,
                relative_position_bias=None, query_mask=None, key_masks=None, need_weights=False):
        if not batch_first:
            query, keys = query.transpose(0, 1), keys.transpose(0, 1)
        b, nq, _ = query.shape
        nk = keys.shape[1] // self.num_head_groups
        hd = self.per_head_dim
        feats, weights = [], []
        lo = 0
        for g, (heads, hi) in enumerate(zip(self.num_heads, self.group_c_idx)):
            q = self.to_qs[g](query[:, :, lo:hi]).view(b, nq, heads, hd).transpose(1, 2)
            kv = self.to_kvs[g](keys[:, g * nk:(g + 1) * nk, lo:hi]).view(b, nk, 2, heads, hd)
            k, v = kv[:, :, 0].transpose(1, 2), kv[:, :, 1].transpose(1, 2)  # (b, heads, nk, hd)
            lo = hi
            attn = (q * self.scale) @ k.transpose(-2, -1)
            if relative_position_bias is not None:
                attn = attn + relative_position_bias[g]
            if key_masks is not None:  # additive -100 (not -inf); softmax only on this branch
                km = key_masks[:, g * nk:(g + 1) * nk]
                attn = torch.softmax(attn + (km != 0).to(attn.dtype).view(b, 1, 1, nk) * -100.0, dim=-1)
            attn = self.attn_drop(attn)
            x = (attn @ v).transpose(1, 2).reshape(b, nq, heads * hd)
            feats.append(self.proj_drop(self.projs[g](x)))
            if need_weights:
                weights.append(attn)
        out = torch.cat(feats, dim=-1)
        if query_mask is not None:
            out = out * (~query_mask).unsqueeze(-1).to(out.dtype)
        if not batch_first:
            out = out.transpose(0, 1)
        return (out, weights) if need_weights else out
