"""A whole backbone forward behind ONE C call (``mssvt_frame_forward``, csrc/frame.hip).

The module path of ``mssvt_amd/fused.py`` issues ~15 entry-point calls, ~65 tensor allocations and the bookkeeping
around them from Python: ~630 us of interpreter time per frame, as much as the frame's kernels.  For the common network
shape -- one resolution level: L two-scale Blocks that share one window configuration, closed by a CompressBlock over
pillar windows (``mssvt.yaml``) -- the same entry points are issued from C++ out of one persistent workspace instead;
Python allocates the four output tensors, makes one call, waits for the early device-to-host copy of the output row
count and wraps the result.  Same kernels, same arguments: the result is bit-identical to the Python-driven path.

Everything this module cannot prove eligible (training, other block layouts, parameters outside the fp16 range of the
FFN, custom tables whose lists overlap, ...) returns ``None`` from :func:`forward` and the caller runs the Python path.
ref: MixedScaleSparseTransformer.forward, pcdet/models/backbones_3d/mssvt_backbone.py:450-472.
"""
import ctypes
import logging
import os

import torch

from . import _lib, fused, mssvt_ops
from .mssvt_utils import SparseTensor

ENABLED = os.environ.get("MSSVT_FRAME", "1") != "0"
_WORDS = 192
# the first norm1 on a second stream, under the Blocks' plan kernel (the pillar lists come out of the level set-up): "auto" = from
# OVERLAP_MIN_VOXELS up (measured with the frame call: one scene 1 432 -> 1 422 frames/s, batch 4 1 739 -> 1 782, batch 8
# 1 828 -> 1 853: the plan kernel slows down when it shares the chip, large frames gain more than that costs)
OVERLAP = os.environ.get("MSSVT_FRAME_OVERLAP", "auto")
OVERLAP_MIN_VOXELS = 250000


class _Stub(object):
    """What the range checks of fused.py read from a plan / SparseTensor."""

    def __init__(self, point_cloud_range):
        self.point_cloud_range = point_cloud_range
        self.coord_bound = max(abs(float(v)) for v in point_cloud_range)


class _Frame(object):
    def __init__(self):
        h = ctypes.c_void_p()
        _lib.call("mssvt_frame_create", ctypes.byref(h))
        self.handle = h.value
        self.keep = []  # tensors / ctypes arrays whose addresses the frame object holds
        self.workspace = None
        self.words = (ctypes.c_int * _WORDS)()

    def __del__(self):
        try:
            if self.handle:
                _lib.lib().mssvt_frame_destroy(self.handle)
        except Exception:
            pass
        self.handle = None


def _params(net):
    out = []
    for blk in net.backbone:
        out.extend(p for p in blk.parameters())
    return out


_why = [None]  # why the last eligibility check said no (read by `_declined`)


def _no(reason, value=False):
    _why[0] = reason
    return value


def _declined(net, reason):
    """Log ONCE per network (and reason) that the whole-frame call is not taking it: the Python-driven path issues the same
    kernels but needs ~3.5x the host time per frame, which should not go unnoticed in production."""
    seen = net.__dict__.setdefault("_frame_declined", set())
    if reason not in seen:
        seen.add(reason)
        logging.getLogger("mssvt_amd.frame").warning(
            "mssvt_frame_forward declined this network (%s): running the Python-driven fused path "
            "(same kernels, ~3.5x the host time per frame)", reason)
    return None


def _structure_ok(net):
    """The constructor-time half of the eligibility: block layout and shapes (cached per network)."""
    from .mssvt_backbone import MixedScaleSparseTransformerBlock as Block, MixedScaleSparseTransformerCompressBlock as Compress
    blocks = list(net.backbone)
    if len(blocks) < 2 or len(blocks) > 17 or not isinstance(blocks[-1], Compress):
        return _no("block layout (Blocks closed by one CompressBlock)")
    body, cmp_blk = blocks[:-1], blocks[-1]
    if any(isinstance(b, Compress) or not isinstance(b, Block) for b in body):
        return _no("a CompressBlock or foreign module among the Blocks")
    if any(hasattr(b, "out_linear") for b in blocks):
        return _no("out_linear")
    C, FF = body[0].linear1.in_features, body[0].linear1.out_features
    if (C, FF) not in fused.FFN_SHAPES or any((b.linear1.in_features, b.linear1.out_features) != (C, FF) for b in blocks):
        return _no("FFN shape outside fused.FFN_SHAPES or not uniform")
    if any(b.plan_key() != body[0].plan_key() or b.max_num_wins != body[0].max_num_wins for b in body):
        return _no("Blocks with different window configurations")
    if not all(fused._supported_static(b) and fused._lists_disjoint(b) for b in body):
        return _no("a Block the fused kernels do not cover, or custom tables with overlapping lists")
    ma0 = body[0].ms_attn
    if any(tuple(b.ms_attn.num_heads) != tuple(ma0.num_heads) or tuple(b.ms_attn.scale_dims) != tuple(ma0.scale_dims) or
           b.ms_attn.per_head_dim != ma0.per_head_dim for b in body):
        return _no("Blocks with different head layouts")
    if len({(b.cbs_pattern, bool(b.use_feature_interpolation)) for b in body}) > 4:
        return _no("more than four (pattern, interpolation) variants")
    # CompressBlock: pillar windows, one head group, every table offset inside the window's own column
    hd = cmp_blk.ms_attn.per_head_dim
    if not (hd <= 64 and (hd & (hd - 1)) == 0 and max(cmp_blk.ms_attn.scale_dims) <= 128):
        return _no("CompressBlock head dimension")
    if not fused._compress_fused_ok(cmp_blk, None, C) or cmp_blk.win2_size is not None or not fused._table_covers_window(cmp_blk):
        return _no("CompressBlock not covered by the fused kernels")
    w = cmp_blk.win1_size
    t = cmp_blk.vox_query_table['win1'].cpu()
    lo = torch.tensor([-(v // 2) for v in w])
    hi = torch.tensor([v - v // 2 - 1 for v in w])
    if not (bool(((t >= lo) & (t <= hi)).all()) and bool((t[:, :2] == 0).all()) and w[0] == 1 and w[1] == 1 and 1 <= t.shape[0] <= 64):
        return _no("CompressBlock window / table is not a pillar")
    return True


def _build(net, dev, batch_size):
    """The frame object for `net` as its parameters are now, or None."""
    blocks = list(net.backbone)
    body, cmp_blk = blocks[:-1], blocks[-1]
    X, Y, Z = (int(v) for v in net.grid_size)
    if Z > 64:
        return _no("grid taller than 64 cells", None)
    C, FF = body[0].linear1.in_features, body[0].linear1.out_features
    stub = _Stub(net.point_cloud_range)
    fr = _Frame()
    f3 = lambda xs: (ctypes.c_float * len(xs))(*[float(v) for v in xs])  # noqa: E731
    i3 = lambda xs: (ctypes.c_int * len(xs))(*[int(v) for v in xs])  # noqa: E731
    _lib.call("mssvt_frame_set_level", fr.handle, int(batch_size), X, Y, Z, int(net.hash_size), f3(net.voxel_size),
              f3(net.point_cloud_range), C, FF)
    P = fused._P
    for blk in body:
        if getattr(blk, "impl", None) != "fused" or getattr(blk, "ffn_arith", fused.FFN_ARITH) != "f16x3":
            return _no("a Block with impl != 'fused' or ffn_arith != 'f16x3'", None)
        t = blk._tables_on(dev)
        fp4, packed_offsets = fused._table_footprint(blk, t)
        if fp4[2] * fp4[3] > 1024:
            return _no("table footprint wider than 1024 words", None)
        r = fused._attn_refs(blk, None)
        n = r["n"]
        pa = lambda ts: (ctypes.c_void_p * n)(*[x.data_ptr() for x in ts])  # noqa: E731
        mode, packed = 0, None
        if getattr(blk, "attn_dtype", "f32") == "bf16" and r["bf16_ok"]:
            mode = 2
        elif getattr(blk, "attn_kv16", fused.ATTN_KV16) and fused._attn_kv16_ok(blk, r, stub):
            mode = 1
            packed = r["kv16_packed"] if getattr(blk, "attn_qo16", fused.ATTN_QO16) else None
        ffr = fused._ffn_refs(blk)
        ffn_packed = fused._ffn_f16_weights(ffr)
        if ffn_packed is None:
            return _no("FFN weights outside the fp16 range of the split products", None)
        arrays = (pa(r["Wq"]), pa(r["bq"]), pa(r["Wkv"]), pa(r["bkv"]), pa(r["Wo"]), pa(r["bo"]))
        fr.keep += [t, packed_offsets, fp4, r, ffr, ffn_packed, arrays, packed]
        _lib.call("mssvt_frame_add_block", fr.handle, i3(blk.win1_size), int(blk.max_num_odd), int(blk.max_num_even),
                  int(blk.max_num_win1), int(blk.max_num_win2), int(t['odd'].shape[0]), int(t['even'].shape[0]),
                  int(t['win1'].shape[0]), int(t['win2'].shape[0]), P(t['odd']), P(t['even']), P(t['win1']), P(t['win2']), fp4,
                  P(packed_offsets), int(blk.key_num_sample), int(blk.max_num_wins), int(blk.cbs_pattern),
                  1 if blk.use_feature_interpolation else 0, P(blk.norm1.weight), P(blk.norm1.bias), float(blk.norm1.eps), n,
                  r["c0"], r["cg"], r["heads"], r["hd"], float(r["scale"]), *arrays, packed, P(r["Wp"]), P(r["bp"]), mode,
                  P(ffr["lnw"]), P(ffr["lnb"]), float(ffr["eps"]), P(ffr["W1"]), P(ffr["b1"]), P(ffr["W2"]), P(ffr["b2"]),
                  P(ffn_packed))
    blk = cmp_blk
    if getattr(blk, "impl", None) != "fused" or getattr(blk, "ffn_arith", fused.FFN_ARITH) != "f16x3":
        return _no("the CompressBlock has impl != 'fused' or ffn_arith != 'f16x3'", None)
    ma = blk.ms_attn
    ffr = fused._ffn_refs(blk)
    ffn_packed = fused._ffn_f16_weights(ffr)
    if ffn_packed is None:
        return _no("CompressBlock FFN weights outside the fp16 range", None)
    t = blk._tables_on(dev)
    split = 1 if fused._compress_f16_ok(blk, stub) else 0
    # the one-launch attention of a sorted pillar level (csrc/compress_ws.hip): the table must list every cell of the slab
    tz = t['win1'].cpu()
    full = bool((tz[:, :2] == 0).all()) and len(set(int(z) for z in tz[:, 2])) == int(blk.win1_size[2]) and \
        int(blk.win1_size[2]) <= int(blk.max_num_win1) <= 32
    ws_packed = fused._compress_ws_weights(blk, stub) if fused.CMP_WS and full else None
    fr.keep += [t, ffr, ffn_packed, ws_packed]
    _lib.call("mssvt_frame_add_compress", fr.handle, i3(blk.win1_size), int(blk.max_num_win1), int(t['win1'].shape[0]),
              P(t['win1']), int(blk.max_num_wins), P(blk.norm1.weight), P(blk.norm1.bias), float(blk.norm1.eps),
              P(blk.pos_proj[0].weight), P(blk.pos_proj[0].bias), P(blk.pos_proj[2].weight), P(blk.pos_proj[2].bias),
              P(ma.to_qs[0].weight), P(ma.to_qs[0].bias), P(ma.to_kvs[0].weight), P(ma.to_kvs[0].bias),
              P(ma.projs[0].weight), P(ma.projs[0].bias), int(ma.per_head_dim), float(ma.scale), split, P(ffr["lnw"]),
              P(ffr["lnb"]), float(ffr["eps"]), P(ffr["W1"]), P(ffr["b1"]), P(ffr["W2"]), P(ffr["b2"]), P(ffn_packed),
              P(ws_packed))
    fr.overlap = None
    return fr


def _state(net, feats, batch_size, stream=0):
    """(frame object or None) for this network / device / batch size AND HIP stream, rebuilt when a parameter moved or
    changed.  One frame object -- persistent workspace, pinned status words, events -- per stream: forwards of the same
    network on different streams (several frames in flight, mssvt_amd/pipeline.py) share nothing they write."""
    st = net.__dict__.get("_frame_state")
    skey = tuple((b.__class__, b.cbs_pattern, b.use_feature_interpolation, b.plan_key(), b.max_num_wins, b.key_num_sample)
                 for b in net.backbone)
    if st is None or st["skey"] != skey:
        st = net.__dict__["_frame_state"] = dict(skey=skey, ok=_structure_ok(net), params=_params(net), key=None, frames={})
        st["why"] = None if st["ok"] else _why[0]
    if not st["ok"]:
        return None
    ps = st["params"]
    key = (feats.device, int(batch_size), OVERLAP, int(net.hash_size), tuple(int(v) for v in net.grid_size),
           tuple(float(v) for v in net.voxel_size), tuple(float(v) for v in net.point_cloud_range), fused.FFN_ARITH, fused.ATTN_KV16, fused.ATTN_QO16, fused.CMP_WS,
           tuple(p._version for p in ps), tuple(p.data_ptr() for p in ps),
           tuple((getattr(b, "impl", None), getattr(b, "attn_dtype", "f32"), getattr(b, "ffn_arith", None),
                  getattr(b, "attn_kv16", None), getattr(b, "attn_qo16", None), b._table_sig,
                  b.vox_query_table['win1'].data_ptr()) for b in net.backbone)) + fused._content_key(ps)
    if st["key"] != key:
        st["frames"] = {}
        st["key"] = key
    if stream not in st["frames"]:
        if len(st["frames"]) >= 16:  # streams come and go: do not keep a workspace for each one ever seen
            st["frames"].pop(next(iter(st["frames"])))
        with torch.no_grad():
            try:
                fr = _build(net, feats.device, batch_size)
            except _lib.MssvtHipError as e:
                if "status -2" not in str(e):  # MSSVT_E_TOOLARGE: a shape the frame call does not cover -> Python path
                    raise
                fr = _no("a shape the frame call does not cover (MSSVT_E_TOOLARGE)", None)
        st["frames"][stream] = fr
        st["why"] = None if fr is not None else _why[0]
    return st["frames"][stream]


def invalidate(net):
    net.__dict__.pop("_frame_state", None)


def forget_stream(net, stream):
    """Drop the frame object (workspace, pinned words) kept for `stream` (a raw handle): the stream is going away."""
    st = net.__dict__.get("_frame_state")
    if st is not None:
        st["frames"].pop(stream or 0, None)


class Pending(object):
    """A frame that has been enqueued but whose host wait (status words + output row count) has not happened yet:
    `finish()` waits for the early device-to-host copy and returns the output SparseTensor (or raises what `forward` raises).
    The frame object's pinned words are overwritten by the NEXT forward on the same stream: finish before that."""

    def __init__(self, net, fr, outs, batch_size, H):
        self.net, self.fr, self.outs, self.batch_size, self.H = net, fr, outs, batch_size, H

    def finish(self):
        sp = _finish(self.net, self.fr, self.outs, self.batch_size, self.H)
        self.outs = None
        return sp


def forward(net, feats, coords, batch_size, defer=False):
    """The output SparseTensor of `net` on (feats, coords), or None: not eligible, run the Python path.  Raises
    fused.UnsortedVoxels when the voxel list is not (b,x,y,z)-sorted (the caller redoes the frame order-agnostically).
    `defer`: return a `Pending` right behind the enqueue instead -- its `finish()` does the frame's one host wait
    (mssvt_amd/pipeline.py keeps several frames in flight that way without the host standing still in between)."""
    if not ENABLED or torch.is_grad_enabled() or fused.FFN_TIMER is not None or net.async_index:
        return None
    if not (fused.SORTED_LEVELS and fused.OCC_COLUMNS and fused.PLAN_TABLES and fused.CMP_FUSED and fused.LEVEL_SETUP):
        return None
    if not (feats.is_cuda and feats.dtype == torch.float32 and feats.dim() == 2 and feats.is_contiguous()):
        return None
    n = feats.shape[0]
    if n <= 0 or coords.shape[0] != n:
        return None
    cur = _lib.stream()
    cur = getattr(cur, "value", cur) or 0
    fr = _state(net, feats, batch_size, cur)
    if fr is None:
        return _declined(net, net.__dict__["_frame_state"].get("why") or "not eligible")
    if feats.shape[1] != net.backbone[0].linear1.in_features:
        return _declined(net, "feature width differs from the first Block's")
    indices = coords if coords.dtype == torch.int32 and coords.is_contiguous() else coords.int().contiguous()
    dev = feats.device
    C = feats.shape[1]
    B, H = int(batch_size), int(net.hash_size)
    need = int(_lib.lib().mssvt_frame_workspace_bytes(fr.handle, n))
    ws = fr.workspace
    if ws is None or ws.numel() < need or ws.device != dev:
        # persistent: the next frame reuses it (stream order keeps the frames apart); grown with 12 % of slack
        fr.workspace = ws = None
        ws = fr.workspace = torch.empty(need + need // 8, dtype=torch.uint8, device=dev)
    overlap = OVERLAP == "1" or (OVERLAP == "auto" and n >= OVERLAP_MIN_VOXELS)
    if fr.overlap != overlap:
        _lib.call("mssvt_frame_set_overlap", fr.handle, 1 if overlap else 0)
        fr.overlap = overlap
    out_f = torch.empty((n, C), dtype=torch.float32, device=dev)
    out_i = torch.empty((n, 4), dtype=torch.int32, device=dev)
    out_t = torch.empty((B, H, 2), dtype=torch.int32, device=dev)
    out_c = torch.empty((B,), dtype=torch.int32, device=dev)
    _lib.call("mssvt_frame_forward", fr.handle, n, feats.data_ptr(), indices.data_ptr(), ws.data_ptr(), ws.numel(),
              out_f.data_ptr(), out_i.data_ptr(), out_t.data_ptr(), out_c.data_ptr(), _lib.stream())
    if defer:
        return Pending(net, fr, (out_f, out_i, out_t, out_c), batch_size, H)
    return _finish(net, fr, (out_f, out_i, out_t, out_c), batch_size, H)


def _finish(net, fr, outs, batch_size, H):
    out_f, out_i, out_t, out_c = outs
    # the forward's single host wait: the early device-to-host copy (status words + output row count)
    _lib.call("mssvt_frame_wait_words", fr.handle, fr.words, _WORDS)
    w = fr.words
    if w[0] & mssvt_ops.ST_UNSORTED:
        raise fused.UnsortedVoxels()
    status = 0
    for off in (0, 64, 128):
        status |= w[off] & (mssvt_ops.ST_TABLE_OVERFLOW | mssvt_ops.ST_WINDOW_OVERFLOW)
    if status:
        fused._check_plan_status(net.backbone[-1], status, H)
    nw = int(w[129])
    cmp_blk = net.backbone[-1]
    grid, vs = [int(v) for v in net.grid_size], [float(v) for v in net.voxel_size]
    sp = SparseTensor(features=out_f[:nw], indices=out_i[:nw],
                      spatial_shape=[grid[i] // int(cmp_blk.win1_size[i]) for i in range(3)],
                      voxel_size=[vs[i] * cmp_blk.win1_size[i] for i in range(3)], point_cloud_range=net.point_cloud_range,
                      batch_size=batch_size, hash_size=net.hash_size, map_table=out_t, gather_dict=None)
    sp.v_bs_cnt, sp._cnt_of = out_c, sp.indices
    sp.map_status = None
    sp._no_sorted_level = True
    return sp
