"""Module fast path: window plan + fused HIP block kernels (``impl="fused"``).

What the reference does per Block with ~100 launches and >= 5B+2 host syncs
 becomes, per forward (every launch a kernel of libmssvt_hip.so),

    once per voxel set   level set-up: counts, occupancy columns, column bases, window partitions (csrc/level_sorted.hip
                         for the (b,x,y,z)-sorted lists DynamicVFE emits -- verified on the device, speculative in the
                         backbone; csrc/hash_build.hip for any other order), the first norm1
    once per window cfg  ONE fused plan kernel (K3 + 2xK7 + 2xK8 + key masks + resolved metadata), work orders and
                         interpolation tables, shared by consecutive Blocks
    per Block            window attention (k_attn_q / k_attn_kv / k_attn_o, or k_attn_bf16), the FFN tail in one launch
                         (k_ffn_ws: interpolation + scatter + residual + norm2 + linear1 + ReLU + linear2 + next norm1)
    CompressBlock        k_cmp_ws on a sorted pillar level (else its plan, k_cmp_query_keys / k_cmp_kv / k_cmp_out), the FFN tail

with no host synchronisation until the end: window counts stay in device memory, per-window buffers
are sized by their capacity (#voxels) and kernels read the live count.  The CompressBlock's data-dependent
output shape is the one host wait per forward (an early async copy; status words ride along).

torch supplies memory and streams; no framework kernel (LayerNorm, GEMM, bincount) runs on the benchmark configuration.
"""
import ctypes

import os

import torch
import torch.nn.functional as F

from . import _lib, mssvt_ops
from .mssvt_utils import batch_counts

try:
    _lib.lib()
except _lib.MssvtHipError:  # library not built: the first entry-point call raises (there is no CPU fallback)
    pass
# scalars and addresses go to the C entry points as plain Python values when their argtypes are declared (_lib.TYPED)
_i, _f = (int, float) if _lib.TYPED else (ctypes.c_int, ctypes.c_float)
def _no_grad(fn):
    """torch.no_grad() as a decorator without its per-call cost when autograd is already off (the inference path enters ~30
    decorated functions per frame; each torch.no_grad() entry clones the context object and flips the grad mode twice)."""
    def wrapper(*args, **kwargs):
        if torch.is_grad_enabled():
            with torch.no_grad():
                return fn(*args, **kwargs)
        return fn(*args, **kwargs)
    wrapper.__name__, wrapper.__doc__ = fn.__name__, fn.__doc__
    return wrapper


_P = _lib.ptr_raw if _lib.TYPED else _lib.ptr_fast  # every tensor handed over here is a contiguous device buffer (allocated below, or a parameter)


def _f3(xs):
    return (ctypes.c_float * 3)(*[float(v) for v in xs])


class _Plan(object):
    pass


# (channels per head group, head dim) pairs instantiated in csrc/block_attn.hip
ATTN_SHAPES = {(8, 8), (16, 8), (16, 16), (24, 8), (32, 8), (32, 16), (32, 32), (48, 16), (64, 8), (64, 16),
               (64, 32)}


TRAIN_COMPACT = os.environ.get("MSSVT_TRAIN_COMPACT", "1") != "0"  # 0: training through the padded operator path


def _needs_grad(block, sp):
    return torch.is_grad_enabled() and (sp.features.requires_grad or any(p.requires_grad for p in block.parameters()))


def _supported_static(block):
    """The constructor-time half of `supported` (window sizes, list lengths, head shapes): evaluated once per block."""
    attn = block.ms_attn
    if any((cg, attn.per_head_dim) not in ATTN_SHAPES for cg in attn.scale_dims) or block.key_num_sample > 64:
        return False
    if block.win2_size is None or len(attn.num_heads) != 2:
        return False
    nq = {0: block.max_num_even, 1: block.max_num_odd, 2: block.max_num_win1}[block.cbs_pattern]
    # plan kernel limits (csrc/window_plan.hip): offsets packed into bytes (|offset| <= 63), FPS lists < 2048 slots
    if max(block.max_num_win1, block.max_num_win2) >= 2048 or max(block.win2_size) > 120:
        return False
    return nq <= 256 and max(block.win1_size) <= 60


def supported(block, sp):
    """Shapes the v1 fused kernels cover; anything else runs the operator-level path."""
    if torch.is_grad_enabled() and (sp.features.requires_grad or any(p.requires_grad for p in block.parameters())):
        return False  # forward-only kernels: training goes through the differentiable ops path
    f = sp.features
    if f.dtype != torch.float32 or not f.is_cuda:
        return False
    key = (block.cbs_pattern, block.key_num_sample, block.max_num_win1, block.max_num_win2)
    ok = block.__dict__.get("_fused_static_ok")
    if ok is None or ok[0] != key:
        ok = block.__dict__["_fused_static_ok"] = (key, _supported_static(block))
    return ok[1]


@_no_grad
def level_state(sp, blocks=()):
    """Per voxel-set (resolution level) device state shared by all plans on it.  A level the backbone did not set up
    (a Block called on its own SparseTensor) tries the sorted set-up first and reads its verdict with one host sync;
    the backbone's input level is set up speculatively without one (setup_input_level)."""
    st = getattr(sp, "_level", None)
    if st is None or st["indices"] is not sp.indices:
        if SORTED_LEVELS and not getattr(sp, "_no_sorted_level", False):
            st = _sorted_level(list(blocks), sp.indices, sp.batch_size, sp.hash_size, sp.spatial_shape)
            if st is not None and int(st["level_status"].item()) & mssvt_ops.ST_UNSORTED:
                st = None
            if st is not None:
                sp.v_bs_cnt, sp._cnt_of = st["v_bs_cnt"], sp.indices
                sp._level = st
                return st
        cnt = getattr(sp, "v_bs_cnt", None)
        if cnt is None or getattr(sp, "_cnt_of", None) is not sp.indices:
            cnt = batch_counts(sp.indices, sp.batch_size)
        st = {"indices": sp.indices, "v_bs_cnt": cnt, "plans": {}}
        sp._level = st
    return st


# Levels whose voxel list is sorted by (b, x, y, z) -- what DynamicVFE / the device voxelizer emit -- are set up from
# one occupancy bitmap (csrc/level_sorted.hip): no voxel hash table, no insert-min / rank passes, no counting atomics.
# The device verifies the order; a list in any other order takes the order-agnostic kernels (mssvt_level_setup).
SORTED_LEVELS = os.environ.get("MSSVT_SORTED_LEVELS", "1") != "0"


class UnsortedVoxels(Exception):
    """Internal: a speculatively sorted level turned out not to be (the frame is redone on the order-agnostic path)."""


def _level_partitions(blocks):
    """The distinct window partitions of the Blocks up to and including the first CompressBlock (at most 4)."""
    from .mssvt_backbone import MixedScaleSparseTransformerCompressBlock as Compress
    todo, keys = [], set()
    for b in blocks:
        k = _partition_key(b)
        if k not in keys and len(todo) < 4:
            keys.add(k)
            todo.append(b)
        if isinstance(b, Compress):
            break
    return todo


_level_static_no_blocks = {}  # (a level set up for no block at all -- tests, tools: nothing of a module is referenced)
_pinned_no_owner = {}


@_no_grad
def _sorted_level(blocks, indices, B, H, spatial_shape, early_readback=False):
    """Level state from `mssvt_level_setup_sorted` (counts, occupancy columns, column bases, window partitions of
    `blocks`), or None when not applicable.  The caller checks st["level_status"] for ST_UNSORTED."""
    from .mssvt_backbone import MixedScaleSparseTransformerCompressBlock as Compress
    n = indices.shape[0]
    X, Y, Z = (int(v) for v in spatial_shape)
    if not (OCC_COLUMNS and indices.is_cuda and n > 0 and indices.dtype == torch.int32 and indices.is_contiguous()
            and Z <= 64):
        return None
    dev = indices.device
    B, H = int(B), int(H)
    # everything that only depends on the blocks and the grid: built once (the frame's front is host bound)
    skey = (tuple((id(b), b.max_num_wins, b.win1_size[0], b.win1_size[1], b.win1_size[2]) for b in blocks), B, X, Y, Z)
    # (owned by the level's first block, i.e. by the backbone module: no module-global references to Blocks)
    cache = blocks[0].__dict__.setdefault("_sorted_static", {}) if blocks else _level_static_no_blocks
    static = cache.get(skey)
    if static is None or any(a is not b for a, b in zip(static["blocks"], blocks)):
        todo = _level_partitions(blocks)
        k = len(todo)
        al = lambda v: (int(v) + 63) // 64 * 64  # noqa: E731  (256-byte aligned pieces)
        # (sample counts and windows-per-sample live in the zeroed block too: the set-up kernels leave them untouched on a
        # list that is not sorted, and a rejected level must read as EMPTY, not as garbage, until the frame is redone)
        sizes = [64, 64 * max(k, 1), al(B + 1), al(2 * B * X * Y), al(B), al(max(k, 1) * B)]
        ints_ = lambda rows: (ctypes.c_int * max(3 * k, 1))(*[int(v) for r in rows for v in r])  # noqa: E731
        if len(cache) > 16:
            cache.clear()
        static = cache[skey] = dict(
            blocks=list(blocks), todo=todo, k=k, sizes=sizes, offs=[sum(sizes[:i]) for i in range(len(sizes))],
            shapes=ints_([[[X, Y, Z][i] // b.win1_size[i] for i in range(3)] for b in todo]),
            wsizes=ints_([b.win1_size for b in todo]),
            maxw=(ctypes.c_int * max(k, 1))(*[int(b.max_num_wins) for b in todo]),
            scratch=int(_lib.lib().mssvt_level_sorted_scratch_ints(_i(B), _i(X), _i(Y))))
    todo, k, sizes, offs = static["todo"], static["k"], static["sizes"], static["offs"]
    # zeroed together with the frame's -1 arena when there is one (FillArena.take_zero), else cleared by the call itself
    arena = mssvt_ops.FillArena.current
    zero = arena.take_zero(sum(sizes)) if arena is not None and arena.buf.device == dev else None
    precleared = zero is not None
    if zero is None:
        zero = torch.empty(sum(sizes), dtype=torch.int32, device=dev)
    status = zero[0:1]
    hdrs = [zero[offs[1] + 64 * i: offs[1] + 64 * (i + 1)] for i in range(k)]
    start = zero[offs[2]:offs[2] + B + 1]
    occ = zero[offs[3]:offs[3] + 2 * B * X * Y].view(torch.int64)
    cnt = zero[offs[4]:offs[4] + B]
    vbase = torch.empty(B * X * Y, dtype=torch.int32, device=dev)
    scratch = torch.empty(static["scratch"], dtype=torch.int32, device=dev)
    # only a CompressBlock's window table is ever read (it becomes the map_table of the block's output)
    tables = [mssvt_ops.full_neg1((B, H, 2), dev) if isinstance(b, Compress) else None for b in todo]
    wins = [torch.empty((n, 4), dtype=torch.int32, device=dev) for _ in range(k)]
    vcounts = zero[offs[5]:offs[5] + max(k, 1) * B].view(max(k, 1), B)
    ptrs = lambda ts: (ctypes.c_void_p * max(k, 1))(*[0 if t is None else t.data_ptr() for t in ts])  # noqa: E731
    _lib.call("mssvt_level_setup_sorted", _i(n), _i(B), _i(X), _i(Y), _i(Z), _i(H), _P(indices), _P(zero),
              ctypes.c_longlong(-zero.numel() * 4 if precleared else zero.numel() * 4), _P(cnt), _P(start), _P(occ), _P(vbase),
              _P(status), _i(k), static["shapes"], static["wsizes"], static["maxw"], ptrs(wins), ptrs(tables),
              ptrs([vcounts[i] for i in range(k)]), ptrs(hdrs), _P(scratch), _lib.stream())
    st = {"indices": indices, "v_bs_cnt": cnt, "plans": {}, "occ": occ, "vbase": vbase, "level_status": status,
          "sorted": True, "status_words": [status], "_zero": zero,
          "partitions": {_partition_key(b): (wins[i], tables[i], vcounts[i], hdrs[i]) for i, b in enumerate(todo)}}
    if early_readback:
        # Every word the host will want from this level -- the level's status, each partition's status and window count
        # (the output shape of the CompressBlock that ends it) -- is final once these three launches are done, i.e. at the
        # very start of the frame: copy them out NOW.  When the host reaches the CompressBlock a frame's worth of launches
        # later the copy has long landed, the forward never blocks on the GPU, and the host runs ahead of it (the frame's
        # front -- a dozen short launches -- was host bound: the GPU idled ~80 us per frame waiting for them).
        n = 64 * (k + 1)
        host = _pinned_words(blocks[0] if blocks else None, n, dev)
        host.copy_(zero[:n], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        st["early"] = (host, ev, {_partition_key(b): 64 * (i + 1) for i, b in enumerate(todo)})
    return st


def _pinned_words(owner, n, dev):
    """A pinned int32 buffer of >= n words, owned by the backbone (its first block) and specific to the device and the
    stream of the forward: every forward reads its words before it returns, so one per (module, device, stream) is enough
    -- two backbones, or one backbone driven on two streams, never read each other's words (allocating pinned memory per
    frame costs more than the frame's front)."""
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), n, torch.cuda.current_stream(dev).cuda_stream)
    store = owner.__dict__.setdefault("_pinned_words", {}) if owner is not None else _pinned_no_owner
    t = store.get(key)
    if t is None:
        if len(store) > 8:
            store.clear()
        t = store[key] = torch.empty(n, dtype=torch.int32, pin_memory=True)
    return t


PARTITION_GROUPS = True  # a Block's window partition and the next CompressBlock's share their launches


def _partition_key(block):
    return (tuple(int(v) for v in block.win1_size), int(block.max_num_wins))


def window_partition(block, sp, st, need_table=False):
    """(win_ind, window table, windows per sample, workspace) of `block`'s windows on this level, cached per
    (window size, max_num_wins).  The partition of the CompressBlock that ends the level only depends on the
    same voxel indices, so it is computed in the same launches (`mssvt_window_partition_multi`).  On a sorted level
    the partitions of Blocks come without a window table (None: nobody reads it); need_table builds it."""
    parts = st.setdefault("partitions", {})
    key = _partition_key(block)
    if key not in parts or (need_table and parts[key][1] is None):
        todo = [block]
        nxt = getattr(sp, "_next_compress", None)
        if (PARTITION_GROUPS and nxt is not None and nxt is not block and _partition_key(nxt) not in parts
                and _partition_key(nxt) != key):
            todo.append(nxt)
        shapes = [[sp.spatial_shape[i] // b.win1_size[i] for i in range(3)] for b in todo]
        res = mssvt_ops.window_partitions_device([b.win1_size for b in todo], [b.max_num_wins for b in todo],
                                                 sp.batch_size, sp.hash_size, shapes, sp.indices)
        for b, r in zip(todo, res):
            parts[_partition_key(b)] = r
    return parts[key]


LEVEL_SETUP = True  # counts + voxel table + occupancy columns + window partitions of the input level in one call


@_no_grad
def setup_input_level(blocks, sp_kwargs, assume_sorted=True):
    """SparseTensor of the backbone input with everything its first resolution level needs (`mssvt_level_setup`:
    per-sample counts, voxel hash table, occupancy columns, the window partitions of the Blocks up to and
    including the first CompressBlock) produced behind ONE fill instead of five.  None when not applicable."""
    from .mssvt_utils import SparseTensor
    from .mssvt_backbone import MixedScaleSparseTransformerCompressBlock as Compress
    indices = sp_kwargs["indices"]
    n = indices.shape[0]
    if not (LEVEL_SETUP and indices.is_cuda and n > 0 and indices.dtype == torch.int32 and indices.is_contiguous()):
        return None
    dev = indices.device
    B, H = int(sp_kwargs["batch_size"]), int(sp_kwargs["hash_size"])
    X, Y, Z = (int(v) for v in sp_kwargs["spatial_shape"])
    if assume_sorted and SORTED_LEVELS:
        # speculative: no host sync here, the verdict (ST_UNSORTED) is read with the frame's other status words
        st = _sorted_level(blocks, indices, B, H, sp_kwargs["spatial_shape"], early_readback=True)
        if st is not None:
            sp = SparseTensor(lazy_map_table=True, **sp_kwargs)
            sp.v_bs_cnt, sp._cnt_of, sp.map_status = st["v_bs_cnt"], sp.indices, None
            st["speculative"] = True
            sp._level = st
            return sp
    todo = _level_partitions(blocks)
    k = len(todo)
    use_occ = OCC_COLUMNS and Z <= 64
    al = lambda v: (int(v) + 63) // 64 * 64  # noqa: E731  (256-byte aligned pieces)
    stride = al(_lib.lib().mssvt_hash_workspace_ints(_i(n), _i(B)))
    sizes = [al(B), stride, k * stride, al(2 * B * X * Y) if use_occ else 0]
    zero = torch.empty(sum(sizes), dtype=torch.int32, device=dev)  # cleared by the call itself
    offs = [sum(sizes[:i]) for i in range(len(sizes))]
    cnt = zero[offs[0]:offs[0] + B]
    map_ws = zero[offs[1]:offs[1] + stride]
    part_ws = zero[offs[2]:offs[2] + k * stride].view(k, stride) if k else None
    occ = zero[offs[3]:offs[3] + 2 * B * X * Y].view(torch.int64) if use_occ else None
    table = mssvt_ops.full_neg1((B, H, 2), dev)
    tables = [mssvt_ops.full_neg1((B, H, 2), dev) for _ in range(k)]
    scratch = [mssvt_ops.full_neg1((B, H, 2), dev) for _ in range(k)]
    wins = [torch.empty((n, 4), dtype=torch.int32, device=dev) for _ in range(k)]
    vcounts = torch.empty((max(k, 1), B), dtype=torch.int32, device=dev)
    shapes = [[[X, Y, Z][i] // b.win1_size[i] for i in range(3)] for b in todo]
    ints = lambda rows: (ctypes.c_int * max(3 * k, 1))(*[int(v) for r in rows for v in r])  # noqa: E731
    ptrs = lambda ts: (ctypes.c_void_p * max(k, 1))(*[t.data_ptr() for t in ts])  # noqa: E731
    _lib.call("mssvt_level_setup", _i(n), _i(B), _i(X), _i(Y), _i(Z), _i(H), _P(indices), _P(zero),
              ctypes.c_longlong(zero.numel() * 4), _P(cnt), _P(table), _P(map_ws), _P(occ),
              _i(k), ints(shapes), ints([b.win1_size for b in todo]),
              (ctypes.c_int * max(k, 1))(*[int(b.max_num_wins) for b in todo]), ptrs(wins), ptrs(tables),
              ptrs(scratch), ptrs([vcounts[i] for i in range(k)]), _P(part_ws), ctypes.c_longlong(stride),
              _lib.stream())
    sp = SparseTensor(map_table=table, **sp_kwargs)
    sp.v_bs_cnt, sp._cnt_of, sp.map_status = cnt, sp.indices, map_ws[0:1]
    sp._level = {"indices": sp.indices, "v_bs_cnt": cnt, "plans": {}, "occ": occ,
                 "partitions": {_partition_key(b): (wins[i], tables[i], vcounts[i], part_ws[i])
                                for i, b in enumerate(todo)}}
    return sp


def occupancy_columns(sp, st):
    """One 64-bit word per (b, x, y) column of the level (bit z = occupied), or None when z > 64."""
    if "occ" not in st:
        X, Y, Z = (int(v) for v in sp.spatial_shape)
        occ = None
        if OCC_COLUMNS and Z <= 64:
            occ = torch.empty(sp.batch_size * X * Y, dtype=torch.int64, device=sp.indices.device)
            _lib.call("mssvt_occupancy_columns", _P(sp.indices), _i(sp.indices.shape[0]), _i(sp.batch_size),
                      _i(X), _i(Y), _i(Z), _P(occ), _lib.stream())
        st["occ"] = occ
    return st["occ"]


PLAN_TABLES = os.environ.get("MSSVT_PLAN_TABLES", "1") != "0"


def _lists_disjoint(block):
    """True when the win1 lists (hence the odd / even lists) of different windows cannot share a voxel: every offset of
    the three tables inside the window's own cells.  Then each listed voxel is owned by its window and the plan kernel
    can write the interpolation tables itself (host check on the tables, cached; custom tables stay correct)."""
    t = block.vox_query_table
    c = block.__dict__.get("_own_cache")
    if c is None or c[0] is not t['win1']:
        lo = torch.tensor([-(w // 2) for w in block.win1_size])
        hi = torch.tensor([w - w // 2 - 1 for w in block.win1_size])
        allt = torch.cat([t[k].reshape(-1, 3).cpu() for k in ('odd', 'even', 'win1')], 0)
        c = block.__dict__["_own_cache"] = (t['win1'], bool(((allt >= lo) & (allt <= hi)).all()))
    return c[1]


def _plan_tables(block, sp, p, key, dev, N):
    """ctypes arguments (num_tabs, lists, interps, zero rows, tab_row / tab_w pointers) of mssvt_window_plan_two for the
    (query list, interpolation) variants of the Blocks that share the plan, and their keys in p.tables."""
    none = (_i(0), None, None, None, None, None)
    p.tables = {}
    group = [b for b in (getattr(sp, "_plan_group", None) or [block]) if b.plan_key() == key and supported(b, sp)]
    if not PLAN_TABLES or not group or not _lists_disjoint(block):
        return none, []
    todo, seen = [], set()
    for b in group:
        C, FF = b.linear1.in_features, b.linear1.out_features
        k = (b.cbs_pattern, 1 if b.use_feature_interpolation else 0, _query(b, p)[1], C)
        if k not in seen and (C, FF) in FFN_SHAPES and len(todo) < 4:
            seen.add(k)
            todo.append((b, k))
    if not todo:
        return none, []
    _attn_buffers_alloc(p, [(k[2], k[3]) for _, k in todo], dev)
    n = len(todo)
    p._tab_rows = mssvt_ops.full_neg1((n, max(N, 1), 4), dev)
    p._tab_ws = torch.empty((n, max(N, 1), 4), dtype=torch.float32, device=dev)  # written with tab_row
    ia = lambda v: (ctypes.c_int * n)(*[int(x) for x in v])  # noqa: E731
    pa = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])  # noqa: E731
    lists = [{1: 0, 0: 1, 2: 2}[k[0]] for _, k in todo]  # cbs_pattern -> list: 1 odd, 0 even, 2 win1
    args = (_i(n), ia(lists), ia([k[1] for _, k in todo]), ia([p.attn_zero[(k[2], k[3])] for _, k in todo]),
            pa([p._tab_rows[i] for i in range(n)]), pa([p._tab_ws[i] for i in range(n)]))
    return args, [k for _, k in todo]


def _voxel_table(sp, st, occ):
    """The voxel hash table for a plan kernel -- or None on a sorted level (column bases instead): the table of such a
    level is only built when somebody asks for `sp.map_table`."""
    if st.get("sorted") and occ is not None and st.get("vbase") is not None:
        return sp._map_table  # whatever exists; never forces the build
    _speculative_verdict(sp)
    return sp.map_table


def _table_footprint(block, t):
    """(min x offset, min y offset, x extent, y extent) over the four query tables (host, cached)."""
    fp = getattr(block, "_footprint_cache", None)
    if fp is None or fp[0] is not t['win2']:
        allt = torch.cat([t[k].reshape(-1, 3).cpu() for k in ('odd', 'even', 'win1', 'win2')], 0)
        lo, hi = allt.min(0).values, allt.max(0).values
        arr = (ctypes.c_int * 4)(int(lo[0]), int(lo[1]), int(hi[0] - lo[0] + 1), int(hi[1] - lo[1] + 1))
        a64 = allt.to(torch.int64) + 64
        # one word per offset: (x+64) | (y+64)<<7 | (z+64)<<14 | column<<21, column = the offset's (x, y) cell of the footprint
        col = (allt[:, 0].to(torch.int64) - int(lo[0])) * int(hi[1] - lo[1] + 1) + (allt[:, 1].to(torch.int64) - int(lo[1]))
        fits = int(arr[2]) * int(arr[3]) <= 1024 and int(a64.max()) < 128 and int(a64.min()) >= 0
        packed = (a64[:, 0] | (a64[:, 1] << 7) | (a64[:, 2] << 14) | ((col if fits else col * 0) << 21)).to(torch.int32)
        fp = block._footprint_cache = (t['win2'], arr, packed.to(t['win2'].device))
    return fp[1], fp[2]


@_no_grad
def two_scale_plan(block, sp, all_lists=False):
    st = level_state(sp, [block] + ([sp._next_compress] if getattr(sp, "_next_compress", None) is not None else []))
    key = block.plan_key()
    if key in st["plans"]:
        p = st["plans"][key]
        if _qmeta(block, p) is not None:
            return p
        # a Block with a query pattern the plan was not built for (Blocks called one by one): once more, every list
        del st["plans"][key]
        return two_scale_plan(block, sp, all_lists=True)
    dev = sp.indices.device
    N = sp.indices.shape[0]
    B, H = sp.batch_size, sp.hash_size
    p = _Plan()
    p.new_spatial_shape = [sp.spatial_shape[i] // block.win1_size[i] for i in range(3)]
    p.win_size_m = [sp.voxel_size[i] * block.win1_size[i] for i in range(3)]
    p.win_ind, p.win_table, p.k_bs_cnt, ws = window_partition(block, sp, st)
    p.num_wins = ws[1:2]  # device scalar
    p.status = ws[0:1]
    st.setdefault("status_words", []).append(p.status)
    p.cap = cap = max(N, 1)
    n_o, n_e, n1, n2, K = (block.max_num_odd, block.max_num_even, block.max_num_win1, block.max_num_win2,
                           block.key_num_sample)
    p.ind_odd = torch.empty((cap, n_o), dtype=torch.int32, device=dev)
    p.ind_even = torch.empty((cap, n_e), dtype=torch.int32, device=dev)
    p.ind_win1 = torch.empty((cap, n1), dtype=torch.int32, device=dev)
    p.k_ind = [torch.empty((cap, K), dtype=torch.int32, device=dev) for _ in range(2)]
    p.k_mask = [torch.empty((cap, K), dtype=torch.uint8, device=dev) for _ in range(2)]
    p.win_vstart = torch.empty(cap, dtype=torch.int32, device=dev)
    p.qbuf = None
    f4 = lambda n: torch.empty((cap, n, 4), dtype=torch.float32, device=dev)  # noqa: E731
    # resolved metadata of the query lists: only for the query patterns of the Blocks that share this plan (each list
    # costs 16 bytes per slot and window; k_query_rows is their one consumer)
    pats = {block.cbs_pattern} | {b.cbs_pattern for b in (getattr(sp, "_plan_group", None) or ()) if b.plan_key() == key}
    if all_lists:
        pats = {0, 1, 2}
    p.qmeta_odd = f4(n_o) if 1 in pats else None
    p.qmeta_even = f4(n_e) if 0 in pats else None
    p.qmeta_win1 = f4(n1) if 2 in pats else None
    p.kmeta = [f4(K), f4(K)]
    p.wcentre = torch.empty((cap, 4), dtype=torch.float32, device=dev)
    p.coord_bound = max(abs(float(v)) for v in sp.point_cloud_range)  # |metric coordinate| of any voxel / window centre
    p.nq_valid = torch.empty((3, cap), dtype=torch.int32, device=dev)
    p.orders = {}
    owners = mssvt_ops.full_neg1((3, cap), dev)
    p.owner_win1, p.owner_odd, p.owner_even = owners[0], owners[1], owners[2]
    t = block._tables_on(dev)
    fp4, packed = _table_footprint(block, t)
    occ = occupancy_columns(sp, st)
    if occ is not None and fp4[2] * fp4[3] > 1024:
        occ = None  # footprint beyond the plan kernel's column tile: it probes the hash instead
    p._plan_args = (  # raw pointers / sizes only (tools/time_plan.py launches the kernel alone with them)
              *[_i(int(v)) for v in sp.spatial_shape],
              *[_i(int(v)) for v in block.win1_size], _i(n_o), _i(n_e), _i(n1), _i(n2), _i(H), _i(B),
              _i(t['odd'].shape[0]), _i(t['even'].shape[0]), _i(t['win1'].shape[0]), _i(t['win2'].shape[0]),
              _P(t['odd']), _P(t['even']), _P(t['win1']), _P(t['win2']), _i(K),
              _P(p.win_ind), _P(p.num_wins), _i(cap), _P(_voxel_table(sp, st, occ)),
              _P(st["v_bs_cnt"]), _P(p.ind_odd), _P(p.ind_even), _P(p.ind_win1),
              _P(p.k_ind[0]), _P(p.k_ind[1]), _P(p.k_mask[0]), _P(p.k_mask[1]),
              _P(p.win_vstart), _P(p.owner_win1), _P(p.owner_odd), _P(p.owner_even),
              _P(sp.indices), _f3(sp.voxel_size), _f3(sp.point_cloud_range[0:3]), _f3(p.win_size_m),
              _P(p.qmeta_odd), _P(p.qmeta_even), _P(p.qmeta_win1), _P(p.kmeta[0]),
              _P(p.kmeta[1]), _P(p.wcentre), _P(p.nq_valid), _P(occ),
              fp4, _P(packed), _P(st.get("vbase") if occ is not None else None),
              _P(st.get("level_status") if occ is not None else None), _P(p.k_bs_cnt))
    # the interpolation tables of the Blocks that share this plan: in the same launch when every voxel has ONE owner
    tab_args, tab_keys = _plan_tables(block, sp, p, key, dev, N)
    _lib.call("mssvt_window_plan_two", *p._plan_args, *tab_args, _lib.stream())
    for i, k in enumerate(tab_keys):
        p.tables[k] = (p._tab_rows[i], p._tab_ws[i])
    st["plans"][key] = p
    return p


def _attn_buffers_alloc(p, specs, dev):
    """Attention output rows for the (nq, C) combinations of `specs` in ONE allocation per C that ends in a
    single zero row: buffer i is the view from its first row to the end, so the shared zero row is row
    `total - 1 - offset_i` of it (one fill launch per plan instead of one per buffer)."""
    bufs = getattr(p, "attn_bufs", None)
    if bufs is None:
        bufs, p.attn_zero = {}, {}
        p.attn_bufs = bufs
    by_c = {}
    for nq, C in specs:
        if (nq, C) not in bufs and (nq, C) not in by_c.setdefault(C, []):
            by_c[C].append((nq, C))
    for C, todo in by_c.items():
        if not todo:
            continue
        total = sum(p.cap * nq for nq, _ in todo) + 1
        big = torch.empty((total, C), dtype=torch.float32, device=dev)
        big[-1].zero_()
        off = 0
        for nq, _ in todo:
            bufs[(nq, C)] = big[off:]
            p.attn_zero[(nq, C)] = total - 1 - off
            off += p.cap * nq


def _attn_buffer(p, nq, C, dev):
    """(>= cap*nq + 1, C) rows of attention output, one per (window, query slot), shared by the blocks of a
    plan; only rows of valid slots are ever written, row `_attn_zero_row` stays zero."""
    if (nq, C) not in getattr(p, "attn_bufs", ()):
        _attn_buffers_alloc(p, [(nq, C)], dev)
    return p.attn_bufs[(nq, C)]


def _attn_zero_row(p, nq, C, dev):
    _attn_buffer(p, nq, C, dev)
    return p.attn_zero[(nq, C)]


def _query(block, p):
    if block.cbs_pattern == 0:
        return p.ind_even, block.max_num_even, p.owner_even
    if block.cbs_pattern == 1:
        return p.ind_odd, block.max_num_odd, p.owner_odd
    return p.ind_win1, block.max_num_win1, p.owner_win1


def _qmeta(block, p):
    return {0: p.qmeta_even, 1: p.qmeta_odd, 2: p.qmeta_win1}[block.cbs_pattern]


def _row_capacity(block, p, nq, num_voxels):
    """Upper bound of the number of valid query slots over all windows: a voxel sits in one window's lists per
    axis with an odd window size and in up to two per axis with an even one (the lists then cover w + 1 cells:
    mssvt_backbone.py:94-97), and never more often than there are slots."""
    overlap = 1
    for w in block.win1_size:
        overlap *= 2 if int(w) % 2 == 0 else 1
    return max(min(int(num_voxels) * overlap, int(p.cap) * int(nq)), 1)


@_no_grad
def _work_order(block, p, nq, num_voxels):
    """Work order + compact query rows of this cbs_pattern's query list (mssvt_plan_order): dict of
    perm, n_act, q_off, nq_valid, row_meta, row_src, n_rows, row_cap."""
    pat = block.cbs_pattern
    if pat not in p.orders:
        dev = p.win_ind.device
        cap_rows = _row_capacity(block, p, nq, num_voxels)
        o = dict(perm=torch.empty(p.cap, dtype=torch.int32, device=dev),
                 n_act=torch.empty(1, dtype=torch.int32, device=dev),
                 q_off=torch.empty(p.cap, dtype=torch.int32, device=dev),
                 nq_valid=p.nq_valid[{1: 0, 0: 1, 2: 2}[pat]],  # rows: odd, even, win1
                 row_meta=torch.empty((cap_rows, 4), dtype=torch.float32, device=dev),
                 row_src=torch.empty((cap_rows, 2), dtype=torch.int32, device=dev),
                 n_rows=torch.empty(1, dtype=torch.int32, device=dev), row_cap=cap_rows)
        _lib.call("mssvt_plan_order", _P(p.num_wins), _P(o["nq_valid"]), _i(nq),
                  _P(_qmeta(block, p)), _i(p.cap), _i(cap_rows), _P(o["perm"]), _P(o["n_act"]),
                  _P(o["q_off"]), _P(o["row_meta"]), _P(o["row_src"]), _P(o["n_rows"]),
                  _lib.stream())
        p.orders[pat] = o
    return p.orders[pat]


def _query_scratch(p, rows, ma, dev):
    """qbuf of mssvt_block_attention: one row per valid query (`rows` = the work order's row capacity), one
    region per head group."""
    width = sum(4 * ((h + 3) // 4) * cg for h, cg in zip(ma.num_heads, ma.scale_dims))
    if p.qbuf is None or p.qbuf.shape[0] < rows or p.qbuf.shape[1] < width:
        p.qbuf = torch.empty((max(rows, 1), width), dtype=torch.float32, device=dev)
    return p.qbuf


# (channels per head group, head dim) pairs instantiated in csrc/block_attn_bf16.hip
ATTN_BF16_SHAPES = {(16, 8), (16, 16), (32, 8), (32, 16), (32, 32), (48, 16), (64, 8), (64, 16), (64, 32)}


def attn_uses_bf16(block):
    """The bf16-operand kernel runs when the module asks for it (`attn_dtype == "bf16"`) and the shape is
    instantiated; anything else runs the fp32 kernels."""
    ma = block.ms_attn
    return (getattr(block, "attn_dtype", "f32") == "bf16" and block.key_num_sample <= 64
            and all((cg, ma.per_head_dim) in ATTN_BF16_SHAPES for cg in ma.scale_dims))


def _attn_refs(block, groups):
    """Constant pieces of the attention call of `block` (channel offsets, widths, head counts as ctypes arrays; the
    parameter tensors of the chosen head groups), built once: indexing nn.ModuleLists and nn.Module attributes per
    frame costs more host time than the launch itself."""
    ma = block.ms_attn
    gs = tuple(range(len(ma.num_heads))) if groups is None else tuple(groups)
    cache = block.__dict__.setdefault("_attn_ref_cache", {})
    r = cache.get(gs)
    if r is None or r["Wq"][0] is not ma.to_qs[gs[0]].weight or r["Wp"] is not block.pos_proj[0].weight:
        ia = lambda v: (ctypes.c_int * len(v))(*[int(x) for x in v])  # noqa: E731
        r = cache[gs] = dict(
            gs=gs, n=len(gs), c0=ia([sum(ma.scale_dims[:g]) for g in gs]), cg=ia([ma.scale_dims[g] for g in gs]),
            heads=ia([ma.num_heads[g] for g in gs]), hd=int(ma.per_head_dim), scale=float(ma.scale),
            Wq=[ma.to_qs[g].weight for g in gs], bq=[ma.to_qs[g].bias for g in gs],
            Wkv=[ma.to_kvs[g].weight for g in gs], bkv=[ma.to_kvs[g].bias for g in gs],
            Wo=[ma.projs[g].weight for g in gs], bo=[ma.projs[g].bias for g in gs],
            Wp=block.pos_proj[0].weight, bp=block.pos_proj[0].bias, K=int(block.key_num_sample),
            bf16_ok=all((cg, ma.per_head_dim) in ATTN_BF16_SHAPES for cg in ma.scale_dims) and block.key_num_sample <= 64)
    return r


# launch B of the fp32 attention (scores, softmax, weighted key sum per window) with split-fp16 matrix operands
# (csrc/block_attn.hip, k_attn_kvh: hi + 2^-11 lo halves, 3 x v_mfma_f32_16x16x32_f16 per product sum, fp32 accumulation --
# the FFN's arithmetic); "0" keeps the fp32 matrix instruction.  Operands outside the fp16 range always take the fp32 form.
ATTN_KV16 = os.environ.get("MSSVT_ATTN_KV16", "1") != "0"
# ... and the two row-tiled launches on pre-split weight fragments (k_attn_q16 / k_attn_o16); "0": fp32 matrix instruction
ATTN_QO16 = os.environ.get("MSSVT_ATTN_QO16", "1") != "0"


@_no_grad
def _attn_kv16_ok(block, r, p):
    """True when the matrix operands of the split-fp16 attention launches stay inside the fp16 range whatever the input
    is: key / query tokens |xhat| + positional term (|xhat| <= sqrt(C) max|w| + max|b|; positional term <= |Wp_c|_1 max|coordinate| + |bp_c|), Q' by |Wq_o|_1 tmax + |bq_o|, Qt = scale Wk_h^T q'_h
    by scale sum_o |Wk_oc| |q'_o|, Xbar (a convex combination of key tokens) by tmax, V by |Wv_o|_1 tmax + |bv_o|, and the
    weights themselves.  Once per parameter version (one small host sync); the same pass packs the projections into
    MFMA fragments (mssvt_attn_pack_weights -> r["kv16_packed"], a ctypes pointer array, or None: shape not instantiated)."""
    ts = [block.norm1.weight, block.norm1.bias, r["Wp"], r["bp"]] + list(r["Wq"]) + list(r["bq"]) + list(r["Wkv"]) + \
        list(r["bkv"]) + list(r["Wo"])
    ver = tuple(t._version for t in ts) + tuple(t.data_ptr() for t in ts) + (float(p.coord_bound),) + _content_key(ts)
    if r.get("kv16_ver") != ver:
        g1, b1, Wp, bp = [t.detach().float() for t in ts[:4]]
        C = g1.numel()
        xmax = (C ** 0.5) * g1.abs().max() + b1.abs().max()
        tmax = xmax + (Wp.reshape(C, -1).abs().sum(1) * p.coord_bound + bp.abs()).max()
        worst = [tmax]
        for Wq, bq, Wkv, bkv, Wo in zip(r["Wq"], r["bq"], r["Wkv"], r["bkv"], r["Wo"]):
            cg = Wq.shape[0]
            Wq, bq, Wkv, bkv, Wo = [t.detach().float() for t in (Wq, bq, Wkv, bkv, Wo)]
            qmax = Wq.abs().sum(1) * tmax + bq.abs()  # (cg) bound of |q'_o|
            # (x log2 e: the Q' hand-off form folds it into the Wk fragments)
            worst += [qmax.max(), (Wkv[:cg].abs() * qmax[:, None]).sum(0).max() * abs(r["scale"]) * 1.4426950408889634,
                      (Wkv[cg:].abs().sum(1) * tmax + bkv[cg:].abs()).max(), Wq.abs().max(),
                      Wkv.abs().max() * max(1.0, abs(r["scale"]) * 1.4426950408889634), Wo.abs().max()]
        worst = torch.stack([w.float() for w in worst]).max()
        r["kv16_ok"] = bool(torch.isfinite(worst).item() and float(worst) < FFN_F16_LIMIT)
        r["kv16_packed"] = None
        if r["kv16_ok"]:
            sizes = [int(_lib.lib().mssvt_attn_packed_bytes(_i(int(cg)), _i(r["hd"]))) for cg in r["cg"]]
            if all(n > 0 for n in sizes):
                blobs = [torch.empty((n,), dtype=torch.uint8, device=g1.device) for n in sizes]
                for cg, Wq, Wkv, Wo, blob in zip(r["cg"], r["Wq"], r["Wkv"], r["Wo"], blobs):
                    _lib.call("mssvt_attn_pack_weights", _i(int(cg)), _i(r["hd"]), _f(r["scale"]),
                              _lib.ptr(Wq.detach().contiguous()), _lib.ptr(Wkv.detach().contiguous()),
                              _lib.ptr(Wo.detach().contiguous()), _lib.ptr(blob), _lib.stream())
                r["kv16_blobs"] = blobs  # keeps the buffers alive
                r["kv16_packed"] = (ctypes.c_void_p * len(blobs))(*[b.data_ptr() for b in blobs])
        r["kv16_ver"] = ver
    return r["kv16_ok"]


def _attn_weight_args(r, pa):
    """The eight weight arguments of an attention entry point (six pointer arrays + the positional layer): built once per
    parameter set -- the tensors of `r` are replaced together with `r` itself (_attn_refs, refresh_weights)."""
    w = r.get("warg")
    ptrs = tuple(t.data_ptr() for k in ("Wq", "bq", "Wkv", "bkv", "Wo", "bo") for t in r[k]) + (r["Wp"].data_ptr(), r["bp"].data_ptr())
    if w is None or w[0] != ptrs:
        w = r["warg"] = (ptrs, (pa(r["Wq"]), pa(r["bq"]), pa(r["Wkv"]), pa(r["bkv"]), pa(r["Wo"]), pa(r["bo"]),
                                _P(r["Wp"]), _P(r["bp"])))
    return w[1]


def _attention_call(block, p, od, C, nq, xhat, qbuf, attn, groups=None):
    """mssvt_block_attention (or its bf16-operand form) for the given head groups (default: all)."""
    r = _attn_refs(block, groups)
    n = r["n"]
    pa = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])  # noqa: E731
    head = (_i(C), _i(n), r["c0"], r["cg"], r["heads"], _i(r["hd"]), _f(r["scale"]), _i(nq), _i(r["K"]), _P(xhat),
            _P(od["n_act"]), _P(od["perm"]), _P(od["q_off"]), _P(od["nq_valid"]), _P(od["n_rows"]), _i(od["row_cap"]),
            _P(od["row_meta"]), _P(od["row_src"]), pa([p.kmeta[g] for g in r["gs"]]), _P(p.wcentre),
            *_attn_weight_args(r, pa))
    if getattr(block, "attn_dtype", "f32") == "bf16" and r["bf16_ok"]:
        _lib.call("mssvt_block_attention_bf16", *head, _P(attn), _lib.stream())
    elif getattr(block, "attn_kv16", ATTN_KV16) and _attn_kv16_ok(block, r, p):
        _lib.call("mssvt_block_attention_kv16", *head, _P(qbuf), _P(attn),
                  r["kv16_packed"] if getattr(block, "attn_qo16", ATTN_QO16) else None, _lib.stream())
    else:
        _lib.call("mssvt_block_attention", *head, _P(qbuf), _P(attn), _lib.stream())


FFN_SHAPES = {(128, 256), (64, 128), (32, 64)}  # instantiated in csrc/ffn.hip
CMP_FUSED = os.environ.get("MSSVT_CMP_FUSED", "1") != "0"
FFN_TIMER = None  # bench.py sets this to a list to time k_ffn_up live (see _ffn_tail)
# arithmetic of the FFN's matrix products: "f16x3" = every fp32 operand split into two fp16 halves (hi + 2^-11 lo: 22 of 24 mantissa bits), three
# 16-bit MFMAs per product sum, fp32 accumulation (k_ffn_ws: same error against float64 as the fp32 instruction, 3/16 of
# its cycles, one launch); "f32" = v_mfma_f32_16x16x4_f32 (k_ffn_up + k_ffn_down).  A module attribute `ffn_arith`
# overrides it; operands outside the fp16 range (checked from the parameters) always take "f32".
FFN_ARITH = os.environ.get("MSSVT_FFN_ARITH", "f16x3")
FFN_F16_LIMIT = 6.0e4  # fp16 max = 65504; the conversions round toward zero (never to inf)
OCC_COLUMNS = os.environ.get("MSSVT_OCC_COLUMNS", "1") != "0"


def _ffn_refs(block):
    """The FFN tail's parameter tensors and sizes of `block`, looked up once (see _attn_refs)."""
    r = block.__dict__.get("_ffn_ref_cache")
    if r is None or r["W1"] is not block.linear1.weight:
        r = block.__dict__["_ffn_ref_cache"] = dict(
            C=int(block.linear1.in_features), FF=int(block.linear1.out_features), W1=block.linear1.weight,
            b1=block.linear1.bias, W2=block.linear2.weight, b2=block.linear2.bias, lnw=block.norm2.weight,
            lnb=block.norm2.bias, eps=float(block.norm2.eps), has_out=hasattr(block, 'out_linear'))
    return r


VERIFY_WEIGHTS = os.environ.get("MSSVT_VERIFY_WEIGHTS", "0") == "1"


def _content_key(ts):
    """Debug mode (MSSVT_VERIFY_WEIGHTS=1): a checksum of the parameter CONTENTS joins the cache keys, so that even a write
    through `.data` (which leaves `_version` alone) is noticed -- at one host sync per call."""
    if not VERIFY_WEIGHTS:
        return ()
    return tuple(float(t.detach().double().sum().item()) + float(t.detach().double().abs().sum().item()) * 1e-3 for t in ts)


@_no_grad
def _ffn_f16_weights(fr):
    """The split-fp16 fragments of W1 / W2 (mssvt_ffn_pack_weights) when the operands of the split-fp16 FFN stay inside
    the fp16 range whatever the input rows are, else None: a LayerNorm output is bounded by sqrt(C) max|w| + max|b|, a
    hidden activation by max_h(|W1_h|_1 xmax + |b1_h|).  Evaluated once per parameter version (one small host sync)."""
    ts = (fr["W1"], fr["b1"], fr["W2"], fr["b2"], fr["lnw"], fr["lnb"])
    ver = tuple(t._version for t in ts) + (fr["W1"].data_ptr(), fr["W2"].data_ptr()) + _content_key(ts)
    if fr.get("f16_ver") != ver:
        W1, b1, W2, b2, lnw, lnb = [t.detach().float() for t in ts]
        xmax = (fr["C"] ** 0.5) * lnw.abs().max() + lnb.abs().max()
        # |W1_h . x| <= |W1_h|_1 max|x|  and  <= |W1_h|_2 |x|_2 with |x|_2 <= sqrt(C) max|w| + |b|_2 (a normalised row has
        # length sqrt(C)): the smaller of the two
        x2 = (fr["C"] ** 0.5) * lnw.abs().max() + lnb.norm()
        hmax = (torch.minimum(W1.abs().sum(1) * xmax, W1.norm(dim=1) * x2) + b1.abs()).max()
        worst = torch.stack([xmax, hmax, W1.abs().max(), W2.abs().max()]).max()
        ok = bool(torch.isfinite(worst).item() and float(worst) < FFN_F16_LIMIT)
        packed = None
        if ok:
            nbytes = int(_lib.lib().mssvt_ffn_packed_bytes(_i(fr["C"]), _i(fr["FF"])))
            packed = torch.empty((nbytes,), dtype=torch.uint8, device=fr["W1"].device)
            _lib.call("mssvt_ffn_pack_weights", _i(fr["C"]), _i(fr["FF"]), _lib.ptr(fr["W1"].detach().contiguous()),
                      _lib.ptr(fr["W2"].detach().contiguous()), _lib.ptr(packed), _lib.stream())
        fr["f16_packed"] = packed
        fr["f16_ver"] = ver
    return fr["f16_packed"]


def _ffn_tail(block, sp, x_new, x_in=None, owner=None, table=None, n_rows_dev=None, apply_out=True, phases=3):
    """y = x + linear2(relu(linear1(norm2(x)))) (+ out_linear) with x = x_new, or 2*x_in on rows
    no list slot owns.  One fused MFMA kernel when the shape is instantiated; it also emits the
    NEXT block's norm1(y) (sp._xhat) so that LayerNorm never runs as a launch of its own."""
    fr = _ffn_refs(block)
    C, FF = fr["C"], fr["FF"]
    if table is not None:
        x_new = x_in  # shapes / dtype template only
    if (C, FF) not in FFN_SHAPES:
        x = x_new if owner is None else torch.where((owner >= 0).unsqueeze(1), x_new, x_in * 2.0)
        y = x + block.linear2(F.relu(block.linear1(block.norm2(x))))
        sp._xhat = None
    else:
        n = x_new.shape[0]
        y = torch.empty_like(x_new)
        nxt = getattr(sp, "_next_norm1", None)
        has_out = fr["has_out"]
        y_norm = None
        if nxt is not None and not has_out and nxt.normalized_shape[0] == C:
            y_norm = torch.empty_like(x_new)
        packed = None
        if phases == 3 and getattr(block, "ffn_arith", FFN_ARITH) == "f16x3":
            packed = _ffn_f16_weights(fr)
            if packed is not None:
                phases = 4  # one launch, split fp16 operands, no hidden scratch
        # fp32 MFMA: two launches with LDS-resident weights; the hidden activations go through this scratch
        split = phases != 4
        hidden = torch.empty((n, FF), dtype=torch.float32, device=x_new.device) if split else packed
        tail = (_P(fr["lnw"]), _P(fr["lnb"]), _f(fr["eps"]), _P(fr["W1"]), _P(fr["b1"]), _P(fr["W2"]), _P(fr["b2"]), _P(y),
                _P(nxt.weight if y_norm is not None else None),
                _P(nxt.bias if y_norm is not None else None),
                _f(nxt.eps if y_norm is not None else 0.0), _P(y_norm), _P(hidden),
                _P(n_rows_dev), _i(phases), _lib.stream())
        def launch(tail_):
            if table is not None:
                (tab_row, tab_w), attn = table
                _lib.call("mssvt_ffn_fused_interp", _i(n), _i(C), _i(FF), _P(x_in), _P(tab_row),
                          _P(tab_w), _P(attn), *tail_)
            else:
                _lib.call("mssvt_ffn_fused", _i(n), _i(C), _i(FF), _P(x_new), _P(x_in),
                          _P(owner), *tail_)

        if FFN_TIMER is not None and phases == 4:
            # bench.py's live roofline: HIP events around k_ffn_ws inside a repeat of the timed steps
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            launch(tail)
            e1.record()
            FFN_TIMER.append(("ws", e0, e1, n if n_rows_dev is None else n_rows_dev, C, FF, y_norm is not None))
        elif FFN_TIMER is not None and split and phases == 3:
            # fp32 arithmetic: k_ffn_up alone (two C calls instead of one: the same two launches, the same stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            launch(tail[:-2] + (_i(1), tail[-1]))
            e1.record()
            launch(tail[:-2] + (_i(2), tail[-1]))
            FFN_TIMER.append(("up", e0, e1, n if n_rows_dev is None else n_rows_dev, C, FF, y_norm is not None))
        else:
            launch(tail)
        sp._xhat = (y_norm, nxt, y) if y_norm is not None else None
    if apply_out and fr["has_out"]:
        y = block.out_linear(y)
    return y


def _norm1(block, sp, x_in):
    """norm1(x_in): taken from the previous block's fused FFN epilogue when it produced it."""
    pre = getattr(sp, "_xhat", None)
    if pre is not None and pre[1] is block.norm1 and pre[2] is sp.features:
        return pre[0]
    return layer_norm(x_in, block.norm1)


LN_WIDTHS = {16, 32, 64, 128, 256}  # instantiated in csrc/rowops.hip


def layer_norm(x, norm):
    C = x.shape[1]
    if C not in LN_WIDTHS or x.dtype != torch.float32 or not x.is_cuda:
        return F.layer_norm(x, (C,), norm.weight, norm.bias, norm.eps)
    x = x.contiguous()
    y = torch.empty_like(x)
    _lib.call("mssvt_layer_norm", _P(x), _i(x.shape[0]), _i(C), _P(norm.weight), _P(norm.bias),
              _f(norm.eps), _P(y), _lib.stream())
    return y


# ---------------------------------------------------------------------------------------------------------
# Index work of a frame on a second HIP stream (MixedScaleSparseTransformer.async_index)
# ---------------------------------------------------------------------------------------------------------
# Everything a resolution level needs before its first feature kernel -- sample counts, voxel hash table, occupancy
# columns, window partitions, the two-scale plan (K3 + 2 x FPS + masks + metadata), work orders, interpolation tables, the
# CompressBlock's plan -- depends on the voxel INDICES only: integer / VALU / latency bound work, ~25 % of a frame's GPU
# time, while the feature kernels are HBM / matrix bound.  With `async_index` the forward issues it on a side stream and
# the feature kernels on the caller's stream behind one event; the next frame's index work then runs UNDER this frame's
# feature kernels (the host is one frame ahead anyway: the forward's only host wait is an early device-to-host copy of
# the window count).  Buffers allocated on the side stream and read by the feature kernels are handed to the caching
# allocator with `record_stream`, so that a buffer freed by the host is not reused by the next frame's index work
# while this frame's feature kernels still read it.
_side_streams = {}


def side_stream(dev):
    dev = torch.device(dev)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    s = _side_streams.get(idx)
    if s is None:
        s = _side_streams[idx] = torch.cuda.Stream(device=idx)
    return s


# A lighter form of the same idea: only the two launches of a frame that neither need nor feed the index
# chain -- the first Block's LayerNorm (HBM bound, 16 us) and the CompressBlock's pillar plan (one lane per window, 10 us
# + its prefill) -- go to the side stream, under the Blocks' plan kernel (VALU / latency bound, 70 us).  Two events, no
# record_stream walk: the side stream starts every frame behind everything queued on the caller's stream
# (wait_stream), so memory it allocates is never rewritten before its readers are done.
# Measured (bench.py, one box each): one scene 0.73 -> 0.80 ms with it (the cross-stream waits and the two stream switches
# cost more than the 26 us they hide), batch 4 (297k voxels) 2.407 -> 2.420 ms, batch 8 (594k) / bf16 4.15 -> 4.09 ms.  "auto": on from SIDE_OVERLAP_MIN_VOXELS up.
SIDE_OVERLAP = os.environ.get("MSSVT_SIDE_OVERLAP", "auto")
SIDE_OVERLAP_MIN_VOXELS = 400000


def side_overlap_on(sp):
    return SIDE_OVERLAP == "1" or (SIDE_OVERLAP == "auto" and sp.features.shape[0] >= SIDE_OVERLAP_MIN_VOXELS)


@_no_grad
def overlap_front(schedule, sp):
    """Issue norm1 of the first Block and the plan of the level's CompressBlock on the side stream; the consumers wait
    for `sp._xhat_event` / `sp._cmp_plan_event` on their own stream (block_forward, _compress_forward_fused)."""
    from .mssvt_backbone import MixedScaleSparseTransformerCompressBlock as Compress
    blk, _nxt, _group, cmp_blk = schedule[0]
    if isinstance(blk, Compress) or getattr(blk, "impl", None) != "fused" or not supported(blk, sp) or \
            sp.features.shape[1] not in LN_WIDTHS or getattr(sp, "_xhat", None) is not None:
        return
    main = torch.cuda.current_stream(sp.features.device)
    s2 = side_stream(sp.features.device)
    s2.wait_stream(main)
    with torch.cuda.stream(s2):
        xhat = layer_norm(sp.features, blk.norm1)
        ev = torch.cuda.Event()
        ev.record(s2)
        sp._xhat, sp._xhat_event = (xhat, blk.norm1, sp.features), ev
        C = sp.features.shape[1]
        if cmp_blk is not None and getattr(cmp_blk, "impl", None) == "fused" and getattr(sp, "_level", None) is not None and \
                sp._level.get("sorted") and compress_supported(cmp_blk, sp) and CMP_FUSED and \
                cmp_blk.linear1.in_features == C and _compress_fused_ok(cmp_blk, sp, C):
            sp._cmp_plan = (cmp_blk, one_scale_plan(cmp_blk, sp, sync=False))
            ev2 = torch.cuda.Event()
            ev2.record(s2)
            sp._cmp_plan_event = ev2


def _wait_side(sp, name):
    ev = sp.__dict__.pop(name, None)
    if ev is not None:
        torch.cuda.current_stream(sp.features.device).wait_event(ev)


@_no_grad
def prefetch_level(schedule, sp):
    """All index work of the input level, on the CURRENT stream: the plans of its Blocks (with their work orders and
    interpolation tables) and the plan of the CompressBlock that ends it.  The feature phase finds them cached."""
    from .mssvt_backbone import MixedScaleSparseTransformerCompressBlock as Compress
    done = set()
    for blk, _nxt_norm, group, nxt_cmp in schedule:
        if getattr(blk, "impl", None) != "fused":
            break
        if isinstance(blk, Compress):
            C = sp.features.shape[1]
            if compress_supported(blk, sp) and CMP_FUSED and blk.linear1.in_features == C and _compress_fused_ok(blk, sp, C):
                sp._cmp_plan = (blk, one_scale_plan(blk, sp, sync=False))
            break
        if not supported(blk, sp) or blk.linear1.in_features != sp.features.shape[1] or hasattr(blk, "out_linear"):
            break
        k = blk.plan_key()
        if k in done:
            continue
        done.add(k)
        sp._plan_group, sp._next_compress = group, nxt_cmp
        p = two_scale_plan(blk, sp)
        if group and not getattr(p, "group_done", False):
            p.group_done = True
            prepare_group([b for b in group if b.plan_key() == k and supported(b, sp)], sp, p)


def record_streams(roots, stream):
    """`record_stream(stream)` on every device tensor reachable from `roots` (plans, level state, arena)."""
    seen, stack = set(), list(roots)
    while stack:
        o = stack.pop()
        if o is None or id(o) in seen:
            continue
        seen.add(id(o))
        if torch.is_tensor(o):
            if o.is_cuda:
                o.record_stream(stream)
        elif isinstance(o, dict):
            stack.extend(o.values())
        elif isinstance(o, (list, tuple)):
            stack.extend(o)
        elif isinstance(o, _Plan) or type(o).__name__ in ("SparseTensor", "FillArena"):
            stack.extend(o.__dict__.values())


def block_forward(block, sp):
    """Fused forward of a MixedScaleSparseTransformerBlock (eval / no-grad)."""
    if not supported(block, sp):
        if _needs_grad(block, sp) and TRAIN_COMPACT:
            from . import train_path  # differentiable compact path (deterministic segmented-sum backward)
            return train_path.block_forward(block, sp)
        _speculative_verdict(sp)
        return block.forward_ops(sp)
    xhat = _norm1(block, sp, sp.features)
    x_in = sp.features.contiguous()
    N, C = x_in.shape
    p = two_scale_plan(block, sp)
    group = getattr(sp, "_plan_group", None)
    if group and not getattr(p, "group_done", False):
        p.group_done = True
        prepare_group([b for b in group if b.plan_key() == block.plan_key() and supported(b, sp)], sp, p)
    q_ind, nq, owner_q = _query(block, p)
    attn = _attn_buffer(p, nq, C, x_in.device)
    od = _work_order(block, p, nq, N)
    ma = block.ms_attn
    qbuf = _query_scratch(p, od["row_cap"], ma, x_in.device)
    vs3, mn3, ws3 = _f3(sp.voxel_size), _f3(sp.point_cloud_range[0:3]), _f3(p.win_size_m)
    _wait_side(sp, "_xhat_event")  # xhat of the frame's first Block comes from the side stream (overlap_front)
    _attention_call(block, p, od, C, nq, xhat, qbuf, attn)
    interp = 1 if block.use_feature_interpolation else 0
    upd_ind, n_upd, owner = (p.ind_win1, block.max_num_win1, p.owner_win1) if interp else (q_ind, nq, owner_q)
    FF = block.linear1.out_features
    if (C, FF) in FFN_SHAPES:
        # interpolation + scatter + residual are folded into the FFN's input stage through a
        # per-voxel table (3 attention rows + weights) that only depends on the plan
        tab = _interp_table(block, sp, p, q_ind, nq, upd_ind, n_upd, owner, interp, vs3, mn3)
        sp.features = _ffn_tail(block, sp, None, x_in, None, table=(tab, attn))
    else:
        # rows no list slot owns are never written here: the FFN reads them as 2 * x_in
        # (features + shortcut, ref quirk R12) through the owner array
        new = torch.empty_like(x_in)
        _lib.call("mssvt_block_interp_scatter", _i(C), _i(nq), _i(n_upd), _i(interp), _P(attn),
                  _P(x_in), _P(new), _P(sp.indices), _P(p.win_ind), _P(p.num_wins),
                  _i(p.cap), _P(p.win_vstart), _P(q_ind), _P(upd_ind), _P(owner), vs3,
                  mn3, _lib.stream())
        sp.features = _ffn_tail(block, sp, new, x_in, owner)
    sp.gather_dict = None
    return sp


@_no_grad
def prepare_group(blocks, sp, p):
    """Work orders + interpolation tables of ALL blocks that share plan `p`, one launch (pair) per kind
    instead of one per block: they depend on the plan only, and a single-workgroup ordering kernel per
    query list is pure latency."""
    dev = sp.indices.device
    N = sp.indices.shape[0]
    ia = lambda v: (ctypes.c_int * len(v))(*[int(x) for x in v])  # noqa: E731
    pa = lambda ts: (ctypes.c_void_p * len(ts))(*[0 if t is None else t.data_ptr() for t in ts])  # noqa: E731
    _attn_buffers_alloc(p, [(_query(b, p)[1], b.linear1.in_features) for b in blocks], dev)
    # --- work orders, one per cbs_pattern
    todo, seen = [], set(p.orders)
    for b in blocks:
        if b.cbs_pattern not in seen:
            seen.add(b.cbs_pattern)
            todo.append(b)
    if 1 < len(todo) <= 4:
        cap_rows = max(_row_capacity(b, p, _query(b, p)[1], N) for b in todo)
        outs = []
        for b in todo:
            _, nq, _ = _query(b, p)
            outs.append(dict(perm=torch.empty(p.cap, dtype=torch.int32, device=dev),
                             n_act=torch.empty(1, dtype=torch.int32, device=dev),
                             q_off=torch.empty(p.cap, dtype=torch.int32, device=dev),
                             nq_valid=p.nq_valid[{1: 0, 0: 1, 2: 2}[b.cbs_pattern]],
                             row_meta=torch.empty((cap_rows, 4), dtype=torch.float32, device=dev),
                             row_src=torch.empty((cap_rows, 2), dtype=torch.int32, device=dev),
                             n_rows=torch.empty(1, dtype=torch.int32, device=dev), row_cap=cap_rows, nq=nq))
        _lib.call("mssvt_plan_order_multi", _i(len(todo)), _P(p.num_wins), pa([o["nq_valid"] for o in outs]),
                  ia([o["nq"] for o in outs]), pa([_qmeta(b, p) for b in todo]), _i(p.cap), _i(cap_rows),
                  pa([o["perm"] for o in outs]), pa([o["n_act"] for o in outs]), pa([o["q_off"] for o in outs]),
                  pa([o["row_meta"] for o in outs]), pa([o["row_src"] for o in outs]),
                  pa([o["n_rows"] for o in outs]), _lib.stream())
        for b, o in zip(todo, outs):
            p.orders[b.cbs_pattern] = o
    # --- interpolation tables, one per (cbs_pattern, interpolation)
    tabs = getattr(p, "tables", None)
    if tabs is None:
        tabs = p.tables = {}
    todo, seen = [], set(tabs)
    for b in blocks:
        C, FF = b.linear1.in_features, b.linear1.out_features
        key = (b.cbs_pattern, 1 if b.use_feature_interpolation else 0, _query(b, p)[1], C)
        if key not in seen and (C, FF) in FFN_SHAPES:
            seen.add(key)
            todo.append((b, key))
    if 1 < len(todo) <= 4:
        vs3, mn3 = _f3(sp.voxel_size), _f3(sp.point_cloud_range[0:3])
        # one fill for all tab_row arrays
        rows = mssvt_ops.full_neg1((len(todo), max(N, 1), 4), dev)
        ws = torch.empty((len(todo), max(N, 1), 4), dtype=torch.float32, device=dev)  # written with tab_row
        nqs, nus, its, qis, uis, ows, zrs = [], [], [], [], [], [], []
        for b, (pat, interp, _, _) in todo:
            q_ind, nq, owner_q = _query(b, p)
            upd_ind, n_upd, owner = (p.ind_win1, b.max_num_win1, p.owner_win1) if interp else (q_ind, nq, owner_q)
            nqs.append(nq); nus.append(n_upd); its.append(interp); qis.append(q_ind); uis.append(upd_ind)
            ows.append(owner); zrs.append(_attn_zero_row(p, nq, b.linear1.in_features, dev))
        _lib.call("mssvt_block_interp_table_multi", _i(len(todo)), ia(nqs), ia(nus), ia(its), _P(sp.indices),
                  _P(p.win_ind), _P(p.num_wins), _i(p.cap), _P(p.win_vstart), pa(qis), pa(uis),
                  pa(ows), vs3, mn3, ia(zrs), pa([rows[i] for i in range(len(todo))]),
                  pa([ws[i] for i in range(len(todo))]), _lib.stream())
        for i, (b, key) in enumerate(todo):
            tabs[key] = (rows[i], ws[i])


def _interp_table(block, sp, p, q_ind, nq, upd_ind, n_upd, owner, interp, vs3, mn3):
    """(tab_row (N,4) int32, tab_w (N,4) f32): where each voxel's update comes from.  Geometry
    only -> computed once per plan and (cbs_pattern, interpolation) and reused by later blocks."""
    # the table embeds the zero row of this (nq, C) attention buffer: part of the key
    key = (block.cbs_pattern, interp, nq, block.linear1.in_features)
    tabs = getattr(p, "tables", None)
    if tabs is None:
        tabs = p.tables = {}
    if key not in tabs:
        dev = sp.indices.device
        N = sp.indices.shape[0]
        tab_row = mssvt_ops.full_neg1((max(N, 1), 4), dev)
        tab_w = torch.empty((max(N, 1), 4), dtype=torch.float32, device=dev)  # written with tab_row
        _lib.call("mssvt_block_interp_table", _i(nq), _i(n_upd), _i(interp), _P(sp.indices),
                  _P(p.win_ind), _P(p.num_wins), _i(p.cap), _P(p.win_vstart), _P(q_ind),
                  _P(upd_ind), _P(owner), vs3, mn3,
                  _i(_attn_zero_row(p, nq, block.linear1.in_features, dev)), _P(tab_row),
                  _P(tab_w), _lib.stream())
        tabs[key] = (tab_row, tab_w)
    return tabs[key]


def _table_covers_window(block):
    """True when the win1 table lists every cell of the window itself (what the generated tables do).  A custom table that
    omits cells can leave a non-empty window with an EMPTY key list; the reference then averages the 32 padded slots
    (uniform softmax over -100-masked scores, ref mssvt_utils.py:129-134) -- the compact kernels here do not restate that
    corner: such a block runs the operator path (which does)."""
    t = block.vox_query_table['win1']
    c = block.__dict__.get("_cover_cache")
    if c is None or c[0] is not t:
        have = set(map(tuple, t.cpu().tolist()))
        w = [int(v) for v in block.win1_size]
        need = ((x, y, z) for x in range(-(w[0] // 2), w[0] - w[0] // 2) for y in range(-(w[1] // 2), w[1] - w[1] // 2)
                for z in range(-(w[2] // 2), w[2] - w[2] // 2))
        c = block.__dict__["_cover_cache"] = (t, all(cell in have for cell in need))
    return c[1]


def compress_supported(block, sp):
    if not _table_covers_window(block):
        return False
    if torch.is_grad_enabled() and (sp.features.requires_grad or any(p.requires_grad for p in block.parameters())):
        return False
    if sp.features.dtype != torch.float32 or not sp.features.is_cuda:
        return False
    hd = block.ms_attn.per_head_dim
    return hd <= 64 and (hd & (hd - 1)) == 0 and max(block.ms_attn.scale_dims) <= 128


@_no_grad
def one_scale_plan(block, sp, sync=True):
    """K2 + K4 + pair-row allocation for a CompressBlock; sync=True: one host sync here (the ragged
    kernels size their buffers with the window count), sync=False: the caller reads p.ws later."""
    st = level_state(sp, [block])
    dev = sp.indices.device
    N, B, H = sp.indices.shape[0], sp.batch_size, sp.hash_size
    p = _Plan()
    p.new_spatial_shape = [sp.spatial_shape[i] // block.win1_size[i] for i in range(3)]
    p.win_size_m = [sp.voxel_size[i] * block.win1_size[i] for i in range(3)]
    p.win_ind, p.win_table, p.k_bs_cnt, ws = window_partition(block, sp, st)
    if p.win_table is None:  # a sorted level built this partition for a Block: no table yet
        p.win_ind, p.win_table, p.k_bs_cnt, ws = window_partition(block, sp, st, need_table=True)
    p.num_wins = ws[1:2]
    cap = max(N, 1)
    ns = block.max_num_win1
    p.with_pad = 1 if block.ms_attn.num_head_groups > 1 else 0
    # the lists of different windows are disjoint iff every table offset stays inside the window
    # (true for the tables this package generates; checked so that custom tables stay correct)
    tw = block.vox_query_table['win1']
    cached = getattr(block, "_disjoint_cache", None)
    if cached is None or cached[0] is not tw:
        lo = torch.tensor([-(w // 2) for w in block.win1_size])
        hi = torch.tensor([w - w // 2 - 1 for w in block.win1_size])
        tcpu = tw.cpu()
        dis = 1 if bool(((tcpu >= lo) & (tcpu <= hi)).all()) else 0
        full = False
        if dis and bool((tcpu[:, :2] == 0).all()):
            dis = 2  # ... and every offset stays in the window's own (x, y) column (pillar windows: a lane per window)
            # ... and the table lists EVERY cell of the slab: a window of a sorted level is then one run of voxel rows
            full = block.win1_size[0] == 1 and block.win1_size[1] == 1 and \
                len(set(int(z) for z in tcpu[:, 2])) == block.win1_size[2]
        cached = block._disjoint_cache = (tw, dis, full)
    p.disjoint = cached[1]
    p.runs = bool(cached[2]) and block.win1_size[2] <= ns <= 32 and bool(st.get("sorted"))
    overlap = 1 if p.disjoint else 8
    row_cap = cap * overlap + (cap if p.with_pad else 0)
    p.k_ind = torch.empty((cap, ns), dtype=torch.int32, device=dev)
    p.win_vstart = torch.empty(cap, dtype=torch.int32, device=dev)
    p.win_cnt = torch.empty(cap, dtype=torch.int32, device=dev)
    p.pair_base = torch.empty(cap, dtype=torch.int32, device=dev)
    p.pair_win = mssvt_ops.full_neg1((row_cap,), dev)
    # pair_vox is only read by the ragged kernels (sync=True); the plan kernel writes every live entry
    p.pair_vox = (torch.full if sync else torch.empty)(*(((row_cap,), -1) if sync else ((row_cap,),)),
                                                       dtype=torch.int32, device=dev)
    p.num_rows = ws[2:3]
    t = block._tables_on(dev)
    occ = st.get("occ") if st.get("sorted") else None
    _lib.call("mssvt_window_plan_one", *[_i(int(v)) for v in sp.spatial_shape],
              *[_i(int(v)) for v in block.win1_size], _i(ns), _i(H), _i(t['win1'].shape[0]), _P(t['win1']),
              _P(p.win_ind), _P(p.num_wins), _i(cap), _P(_voxel_table(sp, st, occ)),
              _P(st["v_bs_cnt"]), _i(p.with_pad), _i(p.disjoint), _i(N), _P(p.k_ind),
              _P(p.win_vstart),
              _P(p.win_cnt), _P(p.pair_base), _P(p.pair_win), _P(p.pair_vox),
              _P(p.num_rows), _P(occ), _P(st.get("vbase") if occ is not None else None),
              _P(st.get("level_status") if occ is not None else None), _lib.stream())
    p.ws, p.N = ws, N
    if not sync:
        # the window count is final here, ~250 us of GPU work before the block ends: copy it to pinned
        # host memory now and wait for THAT copy later (not for the stream) -- the forward returns while
        # the GPU still runs the block's tail and the next frame's launches queue up behind it
        # ... together with the status words of the voxel table and of the Block plans of this level
        early = st.get("early")
        off = early[2].get(_partition_key(block)) if early is not None else None
        if off is not None and p.disjoint and getattr(sp, "map_status", None) is None and \
                all(w.data_ptr() == st["_zero"].data_ptr() or any(w.data_ptr() == st["_zero"].data_ptr() + 4 * o for o in early[2].values())
                    for w in st.get("status_words", [])):
            # (every status word of the level sits in the block copied out at the start of the frame)
            host_t, p.host_ev = early[0], early[1]
            p.host_words = lambda: (lambda h: [h[off], h[off + 1], h[off + 2], h[0]] + [h[o] for o in early[2].values()])(host_t.tolist())
            return p
        words = [ws[:3]] + list(st.get("status_words", []))
        if getattr(sp, "map_status", None) is not None:
            words.append(sp.map_status)
        host_t = torch.empty(3 + len(words) - 1, dtype=torch.int32, pin_memory=True)
        host_t.copy_(torch.cat(words) if len(words) > 1 else ws[:3], non_blocking=True)
        p.host_ev = torch.cuda.Event()
        p.host_ev.record()
        p.host_words = host_t.tolist
        return p
    # the forward's single host sync; the status words of the voxel table and of this level's Block plans
    # ride along (an overflow there must not pass silently either)
    words = [ws[:3]] + list(st.get("status_words", []))
    if getattr(sp, "map_status", None) is not None:
        words.append(sp.map_status)
    host = (torch.cat(words) if len(words) > 1 else ws[:3]).tolist()
    status, p.nw, p.R = host[:3]
    _raise_if_unsorted(st, host[3:])
    for extra in host[3:]:
        status |= extra & (mssvt_ops.ST_TABLE_OVERFLOW | mssvt_ops.ST_WINDOW_OVERFLOW)
    if p.disjoint:
        p.R = N + (p.nw if p.with_pad else 0)  # rows = voxel rows (+ one pad row per window)
        p.num_rows = torch.full((1,), p.R, dtype=torch.int32, device=dev)
    _check_plan_status(block, status, H)
    return p


def check_level_status(sp, max_num_wins=None):
    """Overflow words of the level's voxel table and Block plans when no fused CompressBlock read them with
    its output shape (operator-path CompressBlock, or a backbone that ends on a Block): one host sync."""
    st = getattr(sp, "_level", None)
    words = list(st.get("status_words", [])) if st else []
    if getattr(sp, "map_status", None) is not None:
        words.append(sp.map_status)
    if not words or (st is not None and st.get("status_checked")):
        return
    status = 0
    host = torch.cat(words).tolist()
    _raise_if_unsorted(st, host)
    for w in host:
        status |= w & (mssvt_ops.ST_TABLE_OVERFLOW | mssvt_ops.ST_WINDOW_OVERFLOW)
    if st is not None:
        st["status_checked"] = True
    if status & mssvt_ops.ST_WINDOW_OVERFLOW:
        raise _lib.MssvtHipError("a sample has more windows than max_num_wins")
    if status & mssvt_ops.ST_TABLE_OVERFLOW:
        raise _lib.MssvtHipError("hash table overflow (hash_size=%d)" % sp.hash_size)


def _speculative_verdict(sp):
    """Before anything outside the fused kernels (operator path, lazily built hash table) reads a level that was set
    up speculatively as sorted: fetch its status (one host sync, once per level) and raise UnsortedVoxels first."""
    st = getattr(sp, "_level", None)
    if st is not None and st.get("speculative") and not st.get("verdict_read"):
        st["verdict_read"] = True
        _raise_if_unsorted(st, [int(st["level_status"].item())])


def _raise_if_unsorted(st, words):
    """A level that was set up speculatively as sorted (no host sync) and is not: every partition reported 0 windows,
    the frame's output is empty -- the caller (MixedScaleSparseTransformer.forward) redoes it on the order-agnostic path."""
    if st is not None and st.get("speculative") and any(w & mssvt_ops.ST_UNSORTED for w in words):
        raise UnsortedVoxels()


def _check_plan_status(block, status, H):
    if status & mssvt_ops.ST_WINDOW_OVERFLOW:
        raise _lib.MssvtHipError("a sample has more than max_num_wins=%d windows" % block.max_num_wins)
    if status & mssvt_ops.ST_TABLE_OVERFLOW:
        raise _lib.MssvtHipError("window hash table overflow (hash_size=%d)" % H)


CMP_SHAPES = {(128, 16), (128, 32), (64, 8), (64, 16), (64, 32), (32, 8), (32, 16), (32, 32)}  # csrc/compress_fused.hip


def _compress_fused_ok(block, sp, C):
    """One head group, C / head_dim instantiated, FFN shape instantiated, window lists that cannot overlap."""
    ma = block.ms_attn
    if ma.num_head_groups != 1 or (C, ma.per_head_dim) not in CMP_SHAPES:
        return False
    if (C, block.linear1.out_features) not in FFN_SHAPES or block.linear1.in_features != C:
        return False
    if len(block.pos_proj) < 3 or tuple(block.pos_proj[0].weight.shape[:2]) != (C, 6):
        return False
    return True


@_no_grad
def _compress_f16_ok(block, sp):
    """True when every matrix operand of the CompressBlock attention stays inside the fp16 range whatever the input is
    (the split-fp16 products of csrc/compress_fused.hip): |xhat| <= sqrt(C) max|w1| + max|b1|; the positional hidden
    layer <= |Wp1_h|_1 max|coordinate| + |bp1_h|; key tokens, V rows and their weighted means by the same rule.  Once
    per parameter version (one small host sync)."""
    ma = block.ms_attn
    ts = (block.norm1.weight, block.norm1.bias, block.pos_proj[0].weight, block.pos_proj[0].bias, block.pos_proj[2].weight,
          block.pos_proj[2].bias, ma.to_qs[0].weight, ma.to_kvs[0].weight, ma.to_kvs[0].bias, ma.projs[0].weight)
    ver = tuple(t._version for t in ts) + tuple(t.data_ptr() for t in ts) + (tuple(float(v) for v in sp.point_cloud_range),) + \
        _content_key(ts)
    cache = block.__dict__.setdefault("_cmp_f16_cache", {})
    if cache.get("ver") != ver:
        g1, b1, Wp1, bp1, Wp2, bp2, Wq, Wkv, bkv, Wo = [t.detach().float() for t in ts]
        C = g1.numel()
        coord = max(abs(float(v)) for v in sp.point_cloud_range)
        # element-wise and Euclidean bounds side by side; a product row is bounded by the smaller of |W_o|_1 max|x| and
        # |W_o|_2 |x|_2
        xmax = (C ** 0.5) * g1.abs().max() + b1.abs().max()
        x2 = (C ** 0.5) * g1.abs().max() + b1.norm()
        hvec = Wp1.reshape(C, -1).abs().sum(1) * coord + bp1.abs()  # positional hidden layer, per channel
        W2 = Wp2.reshape(C, -1)
        pvec = torch.minimum(W2.abs().sum(1) * hvec.max(), W2.norm(dim=1) * hvec.norm()) + bp2.abs()
        kmax, k2 = xmax + pvec.max(), x2 + pvec.norm()  # key tokens = xhat + positional term
        vmax = (torch.minimum(Wkv.abs().sum(1) * kmax, Wkv.norm(dim=1) * k2) + bkv.abs()).max()
        worst = torch.stack([xmax, hvec.max(), kmax, vmax, Wq.abs().max(), Wkv.abs().max(), Wo.abs().max(), Wp2.abs().max()]).max()
        cache["ok"] = bool(torch.isfinite(worst).item() and float(worst) < FFN_F16_LIMIT)
        cache["ver"] = ver
    return cache["ok"]


CMP_WS = os.environ.get("MSSVT_CMP_WS", "1") != "0"  # csrc/compress_ws.hip (sorted pillar levels); 0: compress_fused.hip


@_no_grad
def arith_report(net):
    """Which arithmetic every block of `net` RUNS on with its parameters as they are now: the outcome of the fp16-range
    guards per block (not the policy) -- bench.py prints it as config.arith_per_block, so a checkpoint whose weights send a
    block to the fp32-instruction kernels shows in the line.  One small host sync per block and parameter version."""
    from .frame import _Stub
    from .mssvt_backbone import MixedScaleSparseTransformerCompressBlock as Cmp
    stub = _Stub(net.point_cloud_range)
    out = []
    for i, blk in enumerate(net.backbone):
        if getattr(blk, "impl", None) != "fused":
            out.append({"block": i, "attn": "operator path (fp32)", "ffn": "library GEMM (fp32)"})
            continue
        want16 = getattr(blk, "ffn_arith", FFN_ARITH) == "f16x3"
        ffn = "split16" if want16 and _ffn_f16_weights(_ffn_refs(blk)) is not None else "f32"
        if isinstance(blk, Cmp):
            C = blk.linear1.in_features
            if not _compress_fused_ok(blk, None, C):
                attn = "operator path (fp32)"
            elif CMP_WS and _compress_ws_weights(blk, stub) is not None:
                attn = "split16 (k_cmp_ws, one launch)"
            elif want16 and _compress_f16_ok(blk, stub):
                attn = "split16 (three launches)"
            else:
                attn = "f32"
        elif not _supported_static(blk):
            attn = "operator path (fp32)"
        elif attn_uses_bf16(blk):
            attn = "bf16"
        else:
            r = _attn_refs(blk, None)
            ok = getattr(blk, "attn_kv16", ATTN_KV16) and _attn_kv16_ok(blk, r, stub)
            attn = ("split16" if getattr(blk, "attn_qo16", ATTN_QO16) and r.get("kv16_packed") is not None else
                    "split16 (window launch only)") if ok else "f32"
        out.append({"block": i, "attn": attn, "ffn": ffn})
    return out


def _compress_ws_weights(block, sp):
    """The split-fp16 fragments of pos_proj.2 / to_q / to_kv / proj for mssvt_compress_ws (once per parameter version), or
    None: shape not covered (C = 128, one head group of 16-channel heads), fp32 arithmetic asked for, or operands that may
    leave the fp16 range."""
    ma = block.ms_attn
    C = block.linear1.in_features
    if C != 128 or ma.num_head_groups != 1 or ma.per_head_dim != 16 or getattr(block, "ffn_arith", FFN_ARITH) != "f16x3":
        return None
    if not _compress_f16_ok(block, sp):
        return None
    ts = (block.pos_proj[2].weight, ma.to_qs[0].weight, ma.to_kvs[0].weight, ma.projs[0].weight)
    ver = tuple(t._version for t in ts) + tuple(t.data_ptr() for t in ts) + _content_key(ts)
    cache = block.__dict__.setdefault("_cmp_ws_cache", {})
    if cache.get("ver") != ver:
        nbytes = int(_lib.lib().mssvt_compress_ws_packed_bytes(_i(C)))
        packed = torch.empty((nbytes,), dtype=torch.uint8, device=ts[0].device)
        _lib.call("mssvt_compress_ws_pack", _i(C), *[_lib.ptr(t.detach().contiguous()) for t in ts], _lib.ptr(packed), _lib.stream())
        cache["packed"], cache["ver"] = packed, ver
    return cache["packed"]


@_no_grad
def _compress_forward_fused(block, sp, xhat, x_in):
    """Four MFMA launches + the two FFN launches, all counts on the device; ONE host sync at the end
    (the output shape)."""
    C = x_in.shape[1]
    dev = x_in.device
    pre = sp.__dict__.pop("_cmp_plan", None)  # built ahead on the index stream (prefetch_level / overlap_front)
    _wait_side(sp, "_cmp_plan_event")
    p = pre[1] if pre is not None and pre[0] is block else one_scale_plan(block, sp, sync=False)
    if not p.disjoint or p.with_pad:
        return None, p
    N, cap_w, ns = p.N, max(p.N, 1), block.max_num_win1
    ma = block.ms_attn
    vs3, mn3, ws3 = _f3(sp.voxel_size), _f3(sp.point_cloud_range[0:3]), _f3(p.win_size_m)
    f32 = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)  # noqa: E731
    packed = _compress_ws_weights(block, sp) if CMP_WS and getattr(p, "runs", False) else None
    if packed is not None:
        # a sorted pillar level: every window is a run of consecutive rows -> ONE launch, nothing handed through memory
        new = f32(cap_w, C)
        _lib.call("mssvt_compress_ws", _i(C), _i(ma.per_head_dim), _f(ma.scale), _i(block.win1_size[2]), _i(ns), _i(N),
                  _P(p.num_wins), _i(cap_w), _P(sp.indices), _P(p.win_cnt), _P(p.pair_win),
                  vs3, mn3, ws3, _P(xhat), _P(block.pos_proj[0].weight), _P(block.pos_proj[0].bias),
                  _P(block.pos_proj[2].bias), _P(ma.to_qs[0].bias), _P(ma.to_kvs[0].bias), _P(ma.projs[0].bias),
                  _P(packed), _P(new), _lib.stream())
        return _compress_fused_tail(block, sp, p, new)
    qp, ktok, score, vp, new = f32(cap_w, C), f32(max(N, 1), C), f32(max(N, 1), C // ma.per_head_dim), \
        f32(max(N, 1), C), f32(cap_w, C)
    _lib.call("mssvt_compress_fused", _i(C), _i(ma.per_head_dim), _f(ma.scale), _i(ns), _i(N), _P(p.num_wins),
              _i(cap_w), _P(p.win_ind), _P(sp.indices), _P(p.k_ind), _P(p.win_vstart),
              _P(p.win_cnt), _P(p.pair_win), vs3, mn3, ws3, _P(xhat),
              _P(block.pos_proj[0].weight), _P(block.pos_proj[0].bias),
              _P(block.pos_proj[2].weight), _P(block.pos_proj[2].bias),
              _P(ma.to_qs[0].weight), _P(ma.to_qs[0].bias), _P(ma.to_kvs[0].weight),
              _P(ma.to_kvs[0].bias), _P(ma.projs[0].weight), _P(ma.projs[0].bias),
              _P(qp), _P(ktok), _P(score), _P(vp), _P(new),
              _i(1 if getattr(block, "ffn_arith", FFN_ARITH) == "f16x3" and _compress_f16_ok(block, sp) else 0), _lib.stream())
    return _compress_fused_tail(block, sp, p, new)


def _compress_fused_tail(block, sp, p, new):
    """FFN tail over the live windows + the forward's single host wait (the output shape)."""
    y = _ffn_tail(block, sp, new, n_rows_dev=p.num_wins, apply_out=False)  # no residual to the block input (ref :383-385)
    p.host_ev.synchronize()  # the forward's single host wait: the output shape (copied out long ago)
    host = p.host_words()
    status, nw = host[0], host[1]
    _raise_if_unsorted(getattr(sp, "_level", None), host[3:])
    for extra in host[3:]:  # voxel hash table / Block plans: overflow must not pass silently
        status |= extra & (mssvt_ops.ST_TABLE_OVERFLOW | mssvt_ops.ST_WINDOW_OVERFLOW)
    _check_plan_status(block, status, sp.hash_size)
    p.nw = nw
    pre = getattr(sp, "_xhat", None)
    y = y[:nw]
    if hasattr(block, 'out_linear'):
        y = block.out_linear(y)
        sp._xhat = None
    elif pre is not None:
        sp._xhat = (pre[0][:nw], pre[1], y)
    return y, p


def compress_forward(block, sp):
    """Fused forward of a MixedScaleSparseTransformerCompressBlock (eval / no-grad)."""
    if not compress_supported(block, sp):
        if _needs_grad(block, sp) and TRAIN_COMPACT:
            from . import train_path
            return train_path.compress_forward(block, sp)
        check_level_status(sp)
        return block.forward_ops(sp)
    xhat = _norm1(block, sp, sp.features)
    x_in = sp.features.contiguous()
    C = x_in.shape[1]
    dev = x_in.device
    if CMP_FUSED and _compress_fused_ok(block, sp, C):
        y, p = _compress_forward_fused(block, sp, xhat, x_in)
        if y is not None:
            return _compress_finish(sp, p, y)
    p = one_scale_plan(block, sp)
    nw, R, ns = p.nw, p.R, block.max_num_win1
    ma = block.ms_attn
    vs3, mn3, ws3 = _f3(sp.voxel_size), _f3(sp.point_cloud_range[0:3]), _f3(p.win_size_m)
    # key tokens, one row per valid (window, slot) pair: feature + 2-layer positional MLP
    rows = torch.empty((max(R, 1), C), dtype=torch.float32, device=dev)
    _lib.call("mssvt_compress_pos1", _i(C), _P(p.num_rows), _i(R), _P(p.pair_win),
              _P(p.pair_vox), _P(sp.indices), _P(p.win_ind), vs3, mn3, ws3,
              _P(block.pos_proj[0].weight), _P(block.pos_proj[0].bias), _P(rows), _lib.stream())
    k_tok = F.relu(F.linear(rows, block.pos_proj[2].weight.view(C, C), block.pos_proj[2].bias))
    _lib.call("mssvt_compress_add_features", _i(C), _P(p.num_rows), _i(R), _P(p.pair_vox),
              _P(xhat), _P(k_tok), _lib.stream())
    q_tok = torch.empty((max(nw, 1), C), dtype=torch.float32, device=dev)
    _lib.call("mssvt_compress_pool", _i(C), _i(ns), _P(p.num_wins), _i(nw), _P(p.k_ind),
              _P(p.win_vstart), _P(p.win_cnt), _P(xhat), _P(q_tok), _lib.stream())
    G = ma.num_head_groups
    nk = ns // G
    pre = torch.empty_like(q_tok)
    qp = torch.empty_like(q_tok)
    outs = []
    c0 = 0
    for g in range(G):
        cg = ma.scale_dims[g]
        qp[:, c0:c0 + cg] = ma.to_qs[g](q_tok[:, c0:c0 + cg])
        kv = ma.to_kvs[g](k_tok[:, c0:c0 + cg]).contiguous()  # (R, 2*cg) = [K | V]
        _lib.call("mssvt_compress_attention_group", _i(C), _i(c0), _i(cg), _i(ma.per_head_dim), _f(ma.scale),
                  _i(nk), _i(g), _i(p.with_pad), _i(ns), _i(p.N), _P(p.num_wins), _i(nw),
                  _P(p.win_cnt), _P(p.pair_base), _P(p.k_ind), _P(p.win_vstart),
                  _P(qp), _P(kv), _P(pre), _lib.stream())
        c0 += cg
    c0 = 0
    for g in range(G):
        cg = ma.scale_dims[g]
        outs.append(ma.projs[g](pre[:, c0:c0 + cg]))
        c0 += cg
    new = (outs[0] if G == 1 else torch.cat(outs, dim=-1))[:nw].contiguous()
    return _compress_finish(sp, p, _ffn_tail(block, sp, new))  # no residual to the block input (ref :383-385)


def _compress_finish(sp, p, features):
    sp.features = features
    pre = getattr(sp, "_xhat", None)
    if pre is not None:
        sp._xhat = (pre[0], pre[1], features)
    sp.indices = p.win_ind[:p.nw].contiguous()
    sp.v_bs_cnt, sp._cnt_of = p.k_bs_cnt, sp.indices  # windows per sample = rows per sample of the output
    sp.spatial_shape = p.new_spatial_shape
    sp.voxel_size = p.win_size_m
    sp.map_table = p.win_table
    sp.gather_dict = None
    sp._level = None
    sp._ops_plans = None
    # the output rows are in window order (first occurrence), which is (b,x,y,z)-sorted for pillar windows only:
    # later levels take the order-agnostic set-up instead of a sorted attempt + host sync per level
    sp._no_sorted_level = True
    return sp


MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 matrix-core peak (256 CUs x 256 FLOP/clk x 2.4 GHz)


MFMA_F16_PEAK_TFLOPS = 2516.6  # MI355X_MICROARCH.md: dense 16-bit matrix-core peak (256 CUs x 4096 FLOP/clk x 2.4 GHz)


def ffn_timer_summary(samples):
    """{"ws" | "up": (launches, total voxel rows, rows with a second LayerNorm output, total seconds, C, FF)} of the FFN
    launches recorded through FFN_TIMER."""
    out = {}
    for kind, e0, e1, rows, C, FF, norm2 in samples:
        rows = int(rows.item()) if torch.is_tensor(rows) else int(rows)
        cnt, tot, tot2, sec, _, _ = out.get(kind, (0, 0, 0, 0.0, C, FF))
        out[kind] = (cnt + 1, tot + rows, tot2 + (rows if norm2 else 0), sec + e0.elapsed_time(e1) * 1e-3, C, FF)
    return out


def kv16_early(blk, p):
    return getattr(blk, "attn_dtype", "f32") != "bf16" and getattr(blk, "attn_kv16", ATTN_KV16) and \
        _attn_kv16_ok(blk, _attn_refs(blk, None), p)


def _attention_roofline(net, blk, sp, xhat, event_time_ms, peak_gbs, pmc):
    """One Block's window attention on the bench frame against both roofs: algorithmic bytes (window metadata, valid key
    rows Cg wide, query rows in + attention rows out C wide -- the Q~ / Xbar hand-off of the fp32 launches is this
    implementation's own traffic, not counted) / time vs 8 TB/s, and the reference's FLOP (keys projected) / time vs the
    matrix peak of the operand type."""
    p = two_scale_plan(blk, sp)
    x_in = sp.features.contiguous()
    C = x_in.shape[1]
    q_ind, nq, _ = _query(blk, p)
    attn = torch.empty((p.cap, nq, C), dtype=torch.float32, device=x_in.device)
    od = _work_order(blk, p, nq, x_in.shape[0])
    ma = blk.ms_attn
    qbuf = _query_scratch(p, od["row_cap"], ma, x_in.device)
    ms = event_time_ms(lambda: _attention_call(blk, p, od, C, nq, xhat, qbuf, attn), 20)
    ms_ceil = ms_ceil_wv = None
    r_ = _attn_refs(blk, None)
    if r_["n"] == 2 and tuple(ma.scale_dims) == (64, 64) and blk.key_num_sample == 32 and ma.per_head_dim == 16:
        # csrc/ceiling.hip: k_attn_kvh's bytes through the real work order and key metadata, its instruction counts per
        # window and pass, no dependency between them (the two row-tiled launches either side are not part of it)
        ms_ceil = event_time_ms(lambda: _lib.call(
            "mssvt_ceiling_attn_kvh", _i(C), _i(0), _i(64), _i(32), _P(xhat), _P(p.kmeta[0]), _P(p.kmeta[1]), _P(od["perm"]),
            _P(od["n_act"]), _P(od["q_off"]), _P(od["nq_valid"]), _i(od["row_cap"]), _i(p.cap), _P(qbuf), _lib.stream()), 20)
        # ... and the mix of "Wv applied at the end of the window launch" (round 6: measured instead of argued)
        ms_ceil_wv = event_time_ms(lambda: _lib.call(
            "mssvt_ceiling_attn_kvh_variant", _i(1), _i(C), _i(0), _i(64), _i(32), _P(xhat), _P(p.kmeta[0]), _P(p.kmeta[1]),
            _P(od["perm"]), _P(od["n_act"]), _P(od["q_off"]), _P(od["nq_valid"]), _i(od["row_cap"]), _i(p.cap), _P(qbuf),
            _lib.stream()), 20)
    nw = int(p.num_wins.item())
    K = blk.key_num_sample
    keys = [int((p.k_mask[g][:nw] == 0).sum()) for g in range(2)]
    n_q = int((q_ind[:nw] >= 0).sum())
    alg = nw * (16 + 16 * (nq + 2 * K)) + sum(k * 4 * cg for k, cg in zip(keys, ma.scale_dims)) + 2 * n_q * 4 * C
    # reference arithmetic per window and group: Wq, Wo on the valid queries, Wkv on the valid keys, QK^T and PV
    nqv = (q_ind[:nw] >= 0).sum(1).double()
    flop = 0.0
    for g, cg in enumerate(ma.scale_dims):
        kg = (p.k_mask[g][:nw] == 0).sum(1).double()
        flop += float((4.0 * nqv * cg * cg + 4.0 * kg * cg * cg + 4.0 * nqv * kg * cg).sum()) + 12.0 * cg * float((nqv + kg).sum())
    bf16 = getattr(blk, "attn_dtype", "f32") == "bf16" and _attn_refs(blk, None)["bf16_ok"]
    kv16 = not bf16 and getattr(blk, "attn_kv16", ATTN_KV16) and _attn_kv16_ok(blk, _attn_refs(blk, None), p)
    names = ["k_attn_bf16"] if bf16 else ["k_attn_q", "k_attn_kv", "k_attn_o"]  # prefixes: k_attn_q16 / k_attn_kvh / k_attn_o16 too
    # priced against the pipe the products RUN on: bf16 operands -> one 16-bit MFMA per product; split-fp16 operands ->
    # three 16-bit MFMAs per fp32 product (3 x FLOP against the 16-bit peak); fp32 instruction -> the fp32 matrix peak
    peak_tf = MFMA_F16_PEAK_TFLOPS if (bf16 or kv16_early(blk, p)) else MFMA_F32_PEAK_TFLOPS
    gbs = alg / (ms * 1e-3) / 1e9
    counters = {k: {kk: v.get(kk) for kk in ("hbm_bytes_per_launch", "mfma_busy_frac", "valu_issue_frac", "cycles_per_launch")}
                for k, v in (pmc or {}).items() if isinstance(v, dict) and any(k.startswith(n) for n in names)}
    return {"bound": "hbm", "cbs_pattern": int(blk.cbs_pattern),
            "kernel": ("mssvt_block_attention_bf16 (k_attn_bf16: one launch, keys projected in the kernel, bf16 operands)" if bf16
                       else "mssvt_block_attention_kv16, both head groups (k_attn_q16 + k_attn_kvh + k_attn_o16, grid.y = group: "
                            "split-fp16 operands, fp32 accumulation, Q' hand-off)" if kv16
                       else "mssvt_block_attention, both head groups (k_attn_q + k_attn_kv + k_attn_o, grid.y = group)"),
            "achieved": gbs, "peak": peak_gbs, "unit": "GB/s", "frac": gbs / peak_gbs, "algorithmic_bytes_per_launch": alg,
            "avg_launch_us": ms * 1e3, "units_per_launch": {"windows": nw, "valid_key_rows": sum(keys), "valid_query_rows": n_q},
            "matrix": {"algorithmic_flop_per_launch": flop, "tflops": flop / (ms * 1e-3) / 1e12, "peak_tflops": peak_tf,
                       "issued_over_algorithmic": 3.0 if kv16 else 1.0,
                       "frac": (3.0 if kv16 else 1.0) * flop / (ms * 1e-3) / 1e12 / peak_tf,
                       "frac_vs_f32_matrix_peak": flop / (ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS,
                       "operands": "bf16" if bf16 else "f32 as hi + 2^-11 lo fp16 halves, 3 x v_mfma_f32_16x16x32_f16: frac = 3 x FLOP "
                                   "against the 16-bit matrix peak (the pipe it runs on); frac_vs_f32_matrix_peak = the same FLOP "
                                   "against the fp32 instruction's peak (what rounds 3-4 reported as frac)" if kv16 else "f32"},
            "ceiling_us_window_launch": None if ms_ceil is None else ms_ceil * 1e3,
            "ceiling_us_window_launch_wv_fused": None if ms_ceil_wv is None else ms_ceil_wv * 1e3,
            "pmc": counters or None,
            "note": "ceiling_us_window_launch = k_ceiling_attn_kvh (csrc/ceiling.hip, timing only): the bytes and instruction "
                    "counts of the per-window launch k_attn_kvh (its own time: profiles/r05_*_kernel_stats.csv) with no dependency "
                    "between tokens, scores, softmax and the second product; algorithmic bytes exclude the Q' / Xbar hand-off between the launches; pmc = per-launch means of "
                    "profiles/pmc_frame.json (HBM bytes = 2 FETCH + WRITE, MFMA-pipe and VALU-issue busy fractions)"}


def roofline(net, vc, feats, batch, event_time_ms, peak_gbs, live=None):
    """Roofline of the dominant kernel of the frame on the bench inputs.

    Split-fp16 arithmetic (default): k_ffn_ws<128,256>, the whole FFN tail in one launch (5 per frame, the largest share
    of GPU time).  HBM bound: algorithmic bytes per voxel row = 4 C read (x_in) + 4 C written (y) + 4 C written (the next
    block's LayerNorm of y, when the launch emits it -- it replaces that block's own read + write pass); the 3 attention
    rows interpolated into the input are NOT counted (17 MB per launch, L2 / Infinity-Cache resident), so the figure is
    conservative.  fp32 arithmetic (ffn_arith = "f32"): k_ffn_up, fp32 MFMA bound, 2 C FF FLOP per row.
    The attention call of a Block is reported beside it for BOTH query patterns (the heavy odd one and the light even one),
    with the bf16-operand kernel in place of the fp32 launches when the module runs it (configs[2])."""
    import json
    from .mssvt_utils import SparseTensor
    blk = net.backbone[0]
    pmc_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_frame.json")
    counters = {}
    if os.path.exists(pmc_path):
        with open(pmc_path) as f:
            counters = json.load(f)
    with torch.no_grad():
        sp = SparseTensor(features=feats, indices=vc.int().contiguous(), spatial_shape=net.grid_size,
                          voxel_size=net.voxel_size, point_cloud_range=net.point_cloud_range,
                          batch_size=batch, hash_size=net.hash_size)
        sp._plan_group = [b for b in net.backbone if b.plan_key() == blk.plan_key()] if hasattr(blk, "plan_key") else None
        p = two_scale_plan(blk, sp)
        x_in = sp.features.contiguous()
        C = x_in.shape[1]
        xhat = layer_norm(x_in, blk.norm1)
        attn_lines, seen = [], set()
        for b2 in net.backbone:
            if hasattr(b2, "plan_key") and b2.plan_key() == blk.plan_key() and supported(b2, sp) and b2.cbs_pattern not in seen:
                seen.add(b2.cbs_pattern)
                attn_lines.append(_attention_roofline(net, b2, sp, xhat, event_time_ms, peak_gbs, counters))
        q_ind, nq, _ = _query(blk, p)
        vs3, mn3 = _f3(sp.voxel_size), _f3(sp.point_cloud_range[0:3])
        interp = 1 if blk.use_feature_interpolation else 0
        upd_ind, n_upd, owner = (p.ind_win1, blk.max_num_win1, p.owner_win1) if interp else (q_ind, nq, _query(blk, p)[2])
        tab = _interp_table(blk, sp, p, q_ind, nq, upd_ind, n_upd, owner, interp, vs3, mn3)
        abuf = _attn_buffer(p, nq, C, x_in.device)
        abuf.zero_()
        sp._next_norm1 = net.backbone[1].norm1
        split16 = getattr(blk, "ffn_arith", FFN_ARITH) == "f16x3" and _ffn_f16_weights(_ffn_refs(blk)) is not None
        ms_ceil = None
        if split16:
            ms_ws = event_time_ms(lambda: _ffn_tail(blk, sp, None, x_in, None, table=(tab, abuf)), 20)
            ms_up = ms_down = None
            if C == 128 and blk.linear1.out_features == 256:
                # csrc/ceiling.hip: the same bytes from the same tables, the same matrix / vector instruction counts in
                # workgroups of the same shape, NO dependency between the phases: what the structure could reach
                frag = _ffn_f16_weights(_ffn_refs(blk))
                yc, ync = torch.empty_like(x_in), torch.empty_like(x_in)
                ms_ceil = event_time_ms(lambda: _lib.call(
                    "mssvt_ceiling_ffn_ws", _i(x_in.shape[0]), _P(x_in), _P(tab[0]), _P(tab[1]), _P(abuf), _P(frag), _P(yc), _P(ync),
                    _lib.stream()), 20)
        else:
            ms_ws = None
            ms_up = event_time_ms(lambda: _ffn_tail(blk, sp, None, x_in, None, table=(tab, abuf), phases=1), 20)
            ms_down = event_time_ms(lambda: _ffn_tail(blk, sp, None, x_in, None, table=(tab, abuf), phases=2), 20)
    N, FF = x_in.shape[0], blk.linear1.out_features
    flop = 2.0 * C * FF * N
    # HBM bytes per launch of the FFN kernel from the PMC passes committed under profiles/ (tools/pmc_frame.sh: rocprofv3
    # --pmc FETCH_SIZE / WRITE_SIZE in separate runs over whole forwards of this frame, FETCH doubled as MI355X_MICROARCH.md
    # prescribes for 16-B-per-lane reads on gfx950); bench.py cannot collect counters itself
    traffic = (counters.get("k_ffn_ws<%d, %d, true, true>" % (C, FF) if split16 else "k_ffn_up<%d, %d>" % (C, FF)) or {}).get(
        "hbm_bytes_per_launch")
    pmc = pmc_path
    # what the box delivers on bare loops (tools/peaks/run.py; `peak` below stays the guide's number)
    ceilings = None
    cpath = os.path.join(os.path.dirname(pmc), "r01_measured_ceilings.json")
    if os.path.exists(cpath):
        with open(cpath) as f:
            mc = json.load(f)
        ceilings = {"mfma_f32_tflops": max(mc.get("mfma_f32_tflops", {}).values(), default=None),
                    "hbm_copy_gbs": max((v for k, v in mc.get("hbm_gbs", {}).items() if k.startswith("copy")), default=None),
                    "hbm_read_gbs": max((v for k, v in mc.get("hbm_gbs", {}).items() if k.startswith("read")), default=None)}
    other = []
    if split16:
        # bytes per row: x_in + y + the next block's LayerNorm output (every full-size launch of the frame emits it)
        row_bytes = 12.0 * C
        alg_b = row_bytes * N
        head = {"achieved": alg_b / (ms_ws * 1e-3) / 1e9, "alg": alg_b, "avg_launch_us": ms_ws * 1e3,
                "units_per_launch": {"voxel_rows": N, "bytes_per_row": row_bytes, "matrix_flop_per_row": 4 * C * FF},
                "timing": "HIP events around 20 isolated launches on the bench frame"}
        s_ = (live or {}).get("ws")
        if s_ and s_[0] > 0:
            cnt, rows, rows2, sec, _, _ = s_
            lb = 8.0 * C * rows + 4.0 * C * rows2
            head = {"achieved": lb / sec / 1e9, "alg": lb / cnt, "avg_launch_us": sec / cnt * 1e6,
                    "units_per_launch": {"voxel_rows_mean": rows / cnt,
                                         "bytes_per_row": "8 C, + 4 C when the next LayerNorm is emitted",
                                         "matrix_flop_per_row": 4 * C * FF, "launches_timed": cnt},
                    "timing": "HIP events around every k_ffn_ws launch of a repeat of the timed steps",
                    "isolated_launch_us_full_frame_rows": ms_ws * 1e3}
            if traffic is not None:
                traffic = int(traffic * (rows / cnt) / N)  # the PMC passes ran full-size launches (N rows)
        res = {"bound": "hbm",
               "kernel": "k_ffn_ws<128,256> (FFN tail in one launch: input from x_in + 3 attention rows, norm2, linear1 + ReLU + "
                         "linear2 with split-fp16 operands, residual, next norm1)",
               "achieved": head["achieved"], "peak": peak_gbs, "unit": "GB/s", "frac": head["achieved"] / peak_gbs,
               "traffic": traffic, "measured_ceilings": ceilings, "algorithmic_bytes_per_launch": head["alg"],
               "avg_launch_us": head["avg_launch_us"], "units_per_launch": head["units_per_launch"], "timing": head["timing"],
               "matrix": {"flop_per_launch_f32_equivalent": 2.0 * flop,
                          "mfma_issued": "3 x v_mfma_f32_16x16x32_f16 per 16x16x32 product, fp32 accumulate",
                          "tflops_f32_equivalent_isolated": 2.0 * flop / (ms_ws * 1e-3) / 1e12,
                          "frac_of_f16_peak_incl_3x": 3 * 2.0 * flop / (ms_ws * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS}}
        if "isolated_launch_us_full_frame_rows" in head:
            res["isolated_launch_us_full_frame_rows"] = head["isolated_launch_us_full_frame_rows"]
        if ms_ceil is not None:
            res["ceiling_us"] = ms_ceil * 1e3
            res["ceiling"] = {"us": ms_ceil * 1e3, "kernel_us_same_rows": ms_ws * 1e3, "frac_of_ceiling": ms_ceil / ms_ws,
                              "hbm_frac_at_ceiling": alg_b / (ms_ceil * 1e-3) / 1e9 / peak_gbs,
                              "what": "k_ceiling_ffn_ws (csrc/ceiling.hip, timing only): the launch's byte traffic on the frame's "
                                      "real tables, 48 v_mfma_f32_16x16x32_f16 and ~260 vector instructions per wave and 16-row "
                                      "tile on 128 resident fragment registers, 8-wave workgroups, two waves per SIMD -- with no LDS "
                                      "hand-off, no barrier and no dependency between the phases; kernel_us - us = the price of "
                                      "the chain rows -> LDS -> product -> LDS -> product -> rows"}
    else:
        tf_up = flop / (ms_up * 1e-3) / 1e12
        tf_down = flop / (ms_down * 1e-3) / 1e12
        head = {"achieved": tf_up, "alg": flop, "avg_launch_us": ms_up * 1e3,
                "units_per_launch": {"voxel_rows": N, "flop_per_row": 2 * C * FF},
                "timing": "HIP events around 20 isolated launches on the bench frame"}
        s_ = (live or {}).get("up")
        if s_ and s_[0] > 0:
            cnt, rows, _, sec, _, _ = s_
            head = {"achieved": 2.0 * C * FF * rows / sec / 1e12, "alg": 2.0 * C * FF * rows / cnt, "avg_launch_us": sec / cnt * 1e6,
                    "units_per_launch": {"voxel_rows_mean": rows / cnt, "flop_per_row": 2 * C * FF, "launches_timed": cnt},
                    "timing": "HIP events around every k_ffn_up launch of a repeat of the timed steps"}
            if traffic is not None:
                traffic = int(traffic * (rows / cnt) / N)
        res = {"bound": "mfma",
               "kernel": "k_ffn_up<128,256> (norm2 + GEMM1 + ReLU of the FFN tail, fp32 MFMA; input built from x_in + 3 attention rows)",
               "achieved": head["achieved"], "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
               "frac": head["achieved"] / MFMA_F32_PEAK_TFLOPS, "traffic": traffic, "measured_ceilings": ceilings,
               "algorithmic_flop_per_launch": head["alg"], "avg_launch_us": head["avg_launch_us"],
               "units_per_launch": head["units_per_launch"], "timing": head["timing"]}
        other.append({"bound": "mfma", "kernel": "k_ffn_down<128,256> (GEMM2 + residual + next norm1)", "achieved": tf_down,
                      "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf_down / MFMA_F32_PEAK_TFLOPS,
                      "algorithmic_flop_per_launch": flop, "avg_launch_us": ms_down * 1e3})
    other += attn_lines
    res["other_kernels"] = other
    return res
