"""Differentiable COMPACT path of the MsSVT blocks (training).

With autograd on, the reference -- and this package's operator-level ``forward_ops`` -- materialise padded
``(windows, C, slots)`` tensors for every gather (``grouping_operation`` x7 per Block, ref mssvt_backbone.py:260-268) and
accumulate their gradients with atomicAdd (K6 / K11: ref group_features_gpu.cu:15-47, sampling_gpu.cu:53-90,
group_points_gpu.cu:14-50): ~15 GB and ~200 ms per 160k-point training step here, and gradients whose low bits change
from run to run.  This path computes the same function on COMPACT rows:

* index work = the fused path's device-resident plan (``fused.two_scale_plan`` / ``one_scale_plan``: hand-written HIP,
  no padded lists), shared by the Blocks of a level;
* valid query rows ``R``, valid key rows per scale, (query, key) pairs of a window as flat index arrays -- nothing padded;
* every gather and its gradient through ``mssvt_segment_sum_rows`` (csrc/segment_reduce.hip): a segmented sum over an
  inverted index built once per index set, fixed summation order -> **bit-identical gradients run to run**;
* dense math (LayerNorm, the projections, FFN) as library GEMMs through torch autograd; softmax over the pairs of a
  query with ``torch.segment_reduce`` (contiguous segments, deterministic).

Same arithmetic as the reference up to re-association: masked key slots (additive -100 in the reference, weight <= e^-100)
and empty query slots are dropped instead of padded; the interpolation uses the plan's 3-NN table (weights are geometry only).
"""
import ctypes
import os

import torch
import torch.nn.functional as F

from . import _lib, fused, mssvt_ops

_i = ctypes.c_int


# ---------------------------------------------------------------------------------------------------------
# segmented sums
# ---------------------------------------------------------------------------------------------------------

CHUNK = 256  # entries per partial sum of a long contribution list


def _ranges_sum(src, start, end, idx, w, n_dst):
    src = src.contiguous()
    C = src.shape[1]
    dst = torch.empty((n_dst, C), dtype=torch.float32, device=src.device)
    if n_dst:
        _lib.call("mssvt_segment_sum_rows_ranges", _i(C), _i(n_dst), _lib.ptr(start), _lib.ptr(end), _lib.ptr(idx),
                  _lib.ptr(w), _lib.ptr(src), _lib.ptr(dst), _lib.stream())
    return dst


class Segments(object):
    """Entry ranges of a segmented sum ``dst[d] = sum_{e in [off[d], off[d+1])} w[e] src[idx[e]]`` (ascending e).
    Destinations with more than CHUNK entries are cut into fixed chunks (summed by separate lane groups, then added in
    chunk order): the order is fixed either way, and no lane group serialises a whole launch.

    `longest`: None = look at the offsets now (one host sync); an int = the caller's bound (no sync); a 1-element device
    tensor = the word mssvt_csr_transpose left (0 when no list is longer than CHUNK), read later with the next host read of
    sizes (`_read_sizes`), or on first use."""

    def __init__(self, off, idx, w, longest=None):
        self.idx, self.w, self.n_dst = idx, w, off.numel() - 1
        self.off = off
        self.start, self.end = off[:-1], off[1:]
        self.heavy = None
        self.pending = None
        if longest is None:
            counts = off[1:] - off[:-1]
            self._cut(int(counts.max().item()) if self.n_dst else 0)
        elif torch.is_tensor(longest):
            self.pending = longest
        else:
            self._cut(int(longest))

    def _cut(self, longest):
        """Long lists -> fixed chunks.  No host wait: the number of long lists and of their chunks is bounded by the entry
        count (a long list has more than CHUNK entries), the tables are built at that size and padded with empty chunks."""
        self.pending = None
        if longest <= CHUNK:
            return
        off = self.off
        dev = off.device
        nnz = int(self.idx.numel())
        H, M2 = nnz // (CHUNK + 1) + 1, 2 * (nnz // CHUNK) + 2
        counts = (off[1:] - off[:-1]).long()
        heavy = counts > CHUNK
        self.end = torch.where(heavy, self.start, off[1:]).contiguous()  # heavy rows: empty here, filled below
        hd = _nonzero_known(heavy, H, fill=-1)
        real = hd >= 0
        # (padding entries name distinct rows: their exact zeros are added without piling onto one address)
        hd = torch.where(real, hd, torch.arange(H, device=dev) % max(self.n_dst, 1))
        hs = self.start[hd].long()
        hc = torch.where(real, counts[hd], torch.zeros_like(hd))
        nch = (hc + CHUNK - 1) // CHUNK  # 0 for the padding
        last = torch.cumsum(nch, 0)
        first = last - nch
        m = torch.arange(M2, device=dev)
        which = torch.searchsorted(last, m, right=True)  # the long list chunk m belongs to; H past the last chunk
        used = which < H
        which = which.clamp(max=H - 1)
        c_start = hs[which] + (m - first[which]) * CHUNK
        c_end = torch.minimum(c_start + CHUNK, hs[which] + hc[which])
        zero = torch.zeros_like(c_start)
        self.heavy = dict(rows=hd, c_start=torch.where(used, c_start, zero).int().contiguous(),
                          c_end=torch.where(used, c_end, zero).int().contiguous(),
                          p_start=first.int().contiguous(), p_end=last.int().contiguous(),
                          p_idx=torch.arange(M2, dtype=torch.int32, device=dev), n_chunks=M2)

    def sum(self, src):
        if self.idx is None or self.idx.numel() == 0:  # an empty index set: nothing to add (the C entry rejects a NULL idx)
            return torch.zeros((self.n_dst,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        if self.pending is not None:
            self._cut(int(self.pending.item()))
        dst = _ranges_sum(src, self.start, self.end, self.idx, self.w, self.n_dst)
        h = self.heavy
        if h is not None:
            part = _ranges_sum(src, h["c_start"], h["c_end"], self.idx, self.w, h["n_chunks"])
            # (the kernel above left the long lists' rows zero; padding rows add an exact 0 to row 0)
            dst.index_add_(0, h["rows"], _ranges_sum(part, h["p_start"], h["p_end"], h["p_idx"], None, h["rows"].numel()))
        return dst


def _sum_into(seg, src, dst, c0, accumulate):
    """Segments.sum of `src` (M, cg) written / added to columns [c0, c0 + cg) of `dst` (n_dst, C): no temporary."""
    cg = src.shape[1]
    view = dst[:, c0:c0 + cg]
    if seg.idx is None or seg.idx.numel() == 0 or seg.n_dst == 0:
        if not accumulate:
            view.zero_()
        return
    if seg.pending is not None:
        seg._cut(int(seg.pending.item()))
    src = src.contiguous()
    _lib.call("mssvt_segment_sum_rows_strided", _i(cg), _i(seg.n_dst), _lib.ptr(seg.start), _lib.ptr(seg.end), _lib.ptr(seg.idx),
              _lib.ptr(seg.w), _lib.ptr(src), _i(cg), ctypes.c_void_p(view.data_ptr()), _i(dst.stride(0)),
              _i(1 if accumulate else 0), _lib.stream())
    h = seg.heavy
    if h is not None:  # the long lists (empty ranges above: 0 written / added), chunk sums added in chunk order
        part = _ranges_sum(src, h["c_start"], h["c_end"], seg.idx, seg.w, h["n_chunks"])
        tot = _ranges_sum(part, h["p_start"], h["p_end"], h["p_idx"], None, h["rows"].numel())
        view.index_add_(0, h["rows"], tot)  # 0 (written) or the earlier sum (accumulated) + the long lists' sums


# Segments whose longest-list word has not been read yet.  Process-wide like the workspaces above: one process drives one
# GPU from one thread (bench.py --gpus N starts N processes); the list only ever SAVES a host read -- a Segments that is
# used before the next read fetches its own word (`Segments.sum`, `_sum_into`).
_deferred = []


def _read_sizes(words):
    """int32 device words -> host ints, in one host wait shared with every pending longest-list word."""
    segs = [g for g in _deferred if g.pending is not None]
    del _deferred[:]
    n = words.numel()
    host = torch.cat([words.reshape(-1).int()] + [g.pending for g in segs]).tolist()
    for g, longest in zip(segs, host[n:]):
        g._cut(longest)
    return host[:n]


def _nonzero_known(mask, count, fill=None):
    """Positions of the true entries of a 1-D mask, ascending, without the host wait torch.nonzero needs for its output
    shape: `count` is their number, or with `fill` an upper bound (the tail is padded with `fill`)."""
    if hasattr(torch, "nonzero_static"):
        try:
            return torch.nonzero_static(mask, size=count, fill_value=-1 if fill is None else fill)[:, 0]
        except (RuntimeError, NotImplementedError):
            pass
    pos = torch.nonzero(mask, as_tuple=True)[0]
    if fill is not None:
        pos = torch.cat([pos, pos.new_full((count - pos.numel(),), fill)])
    return pos


def segment_sum_rows(src, off, idx, w, n_dst):
    """dst[d] = sum_{e in [off[d], off[d+1])} w[e] * src[idx[e]] in ascending e (HIP, deterministic)."""
    return _ranges_sum(src, off[:-1], off[1:], idx, w, n_dst)


_csr_ws = {}


def _csr_workspace(dev, nnz, n_src):
    need = int(_lib.lib().mssvt_csr_transpose_workspace_bytes(_i(nnz), _i(n_src)))
    ws = _csr_ws.get(dev)
    if ws is None or ws.numel() < need:
        ws = _csr_ws[dev] = torch.empty((need,), dtype=torch.uint8, device=dev)
    return ws


class Csr(object):
    """A (weighted) gather ``dst[d] = sum_e w[e] src[idx[e]]`` with its transpose (the inverted index that turns the
    gradient scatter-add into a segmented sum; mssvt_csr_transpose, csrc/csr_transpose.hip: no host sync).  Built once
    per index set and reused by every block sharing the plan.  `drop_src`: a source row whose gradient is not needed (the
    constant zero row): its list is left empty.  `off` None: a plain row gather.  `fwd_longest`: the caller's bound on
    the entries per destination (<= CHUNK for everything this file builds); `bwd_longest`: the same for the inverted
    lists (None: the word the transpose leaves on the device, read later)."""

    def __init__(self, off, idx, w, n_src, drop_src=None, fwd_longest=None, bwd_longest=None):
        dev = idx.device
        idx = idx.int().contiguous()
        nnz = idx.numel()
        self.n_src = int(n_src)
        self.n_dst = nnz if off is None else off.numel() - 1
        t_off = torch.empty(self.n_src + 1, dtype=torch.int32, device=dev)
        t_idx = torch.empty(nnz, dtype=torch.int32, device=dev)
        t_w = None if w is None else torch.empty(nnz, dtype=torch.float32, device=dev)
        longest = torch.empty(1, dtype=torch.int32, device=dev)
        _lib.call("mssvt_csr_transpose", _i(nnz), _i(self.n_dst), _i(self.n_src), _lib.ptr(off), _lib.ptr(idx) if nnz else None,
                  _lib.ptr(w) if nnz else None, _i(-1 if drop_src is None else int(drop_src)), _i(CHUNK), _lib.ptr(t_off),
                  _lib.ptr(t_idx) if nnz else None, _lib.ptr(t_w) if (nnz and w is not None) else None, _lib.ptr(longest),
                  _lib.ptr(_csr_workspace(dev, nnz, self.n_src)) if nnz else None, _lib.stream())
        if off is None:
            off = torch.arange(nnz + 1, dtype=torch.int32, device=dev)
            fwd_longest = 1
        self.off, self.idx, self.w = off, idx, w
        self.fwd = Segments(off, idx, w, longest=fwd_longest)
        # (the cut of long lists is a matter of speed only: a caller that knows its lists are short says so and saves the read)
        self.bwd = Segments(t_off, t_idx, t_w, longest=longest if bwd_longest is None else bwd_longest)
        self.t_off, self.t_idx, self.t_w = t_off, t_idx, t_w

    @staticmethod
    def gather(idx, n_src, bwd_longest=None):
        """Plain row gather dst[i] = src[idx[i]]."""
        return Csr(None, idx, None, n_src, bwd_longest=bwd_longest)


def _gather_unique(idx, n_src):
    """Csr of a plain row gather whose rows are DISTINCT (a window's query rows when every window size is odd, the pillar
    lists of a CompressBlock): the inverse is a partial permutation -- one scatter instead of the sort of
    mssvt_csr_transpose; source row v sums the one entry inv[v] or nothing."""
    c = Csr.__new__(Csr)
    dev = idx.device
    idx = idx.int().contiguous()
    nnz = idx.numel()
    c.n_src, c.n_dst = int(n_src), nnz
    inv = torch.empty(c.n_src, dtype=torch.int32, device=dev)
    bwd_idx = torch.empty(c.n_src, dtype=torch.int32, device=dev)
    bwd_end = torch.empty(c.n_src, dtype=torch.int32, device=dev)
    _lib.call("mssvt_train_unique_inverse", _i(nnz), _i(c.n_src), _lib.ptr(idx) if nnz else None, _lib.ptr(inv), _lib.ptr(bwd_idx),
              _lib.ptr(bwd_end), _lib.stream())
    c.off = torch.arange(nnz + 1, dtype=torch.int32, device=dev)
    c.idx, c.w = idx, None
    c.fwd = Segments(c.off, idx, None, longest=1)
    c.bwd = Segments(torch.arange(c.n_src + 1, dtype=torch.int32, device=dev), bwd_idx, None, longest=1)
    c.bwd.end = bwd_end  # an empty range where no entry reads the row
    c.t_off = c.t_idx = c.t_w = None
    return c


class _SegmentSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, csr):
        ctx.csr = csr
        return csr.fwd.sum(src)

    @staticmethod
    def backward(ctx, grad):
        return ctx.csr.bwd.sum(grad), None


def gather_sum(src, csr):
    """Differentiable ``dst[d] = sum_e w[e] src[idx[e]]``; forward and backward are both segmented sums."""
    return _SegmentSum.apply(src, csr)


class _RepeatRows(torch.autograd.Function):
    """x (R, ...) -> rows repeated `lengths[r]` times (contiguous segments); the gradient is a segmented sum in row
    order (torch's own repeat_interleave backward is an atomic index_add)."""

    @staticmethod
    def forward(ctx, x, lengths, total):
        ctx.lengths = lengths
        return torch.repeat_interleave(x, lengths, dim=0, output_size=total)

    @staticmethod
    def backward(ctx, grad):
        return torch.segment_reduce(grad.contiguous(), "sum", lengths=ctx.lengths, unsafe=True), None, None


def repeat_rows(x, lengths, total):
    return _RepeatRows.apply(x, lengths, total)


class _PairAttention(torch.autograd.Function):
    """softmax(q k^T) v of every window's queries against that window's keys (csrc/pair_attn.hip): one launch forward,
    one backward, nothing of size (pairs, ...) in memory."""

    @staticmethod
    def forward(ctx, q, kv, wins, heads, hd):
        q, kv = q.contiguous(), kv.contiguous()
        R, cg = q.shape
        O = torch.empty_like(q)
        lse = torch.empty((R, heads), dtype=torch.float32, device=q.device)
        nw = wins["q_off"].numel()
        if R > 0:
            _lib.call("mssvt_pair_attention_fwd", _i(nw), _i(cg), _i(heads), _i(hd), _lib.ptr(wins["q_off"]),
                      _lib.ptr(wins["q_cnt"]), _lib.ptr(wins["k_off"]), _lib.ptr(wins["k_cnt"]), _lib.ptr(q), _lib.ptr(kv),
                      _lib.ptr(O), _lib.ptr(lse), _lib.stream())
        ctx.save_for_backward(q, kv, O, lse)
        ctx.wins, ctx.shape = wins, (nw, cg, heads, hd)
        return O

    @staticmethod
    def backward(ctx, dO):
        q, kv, O, lse = ctx.saved_tensors
        nw, cg, heads, hd = ctx.shape
        wins = ctx.wins
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        if kv.shape[0] > 0 and nw > 0:
            _lib.call("mssvt_pair_attention_bwd", _i(nw), _i(cg), _i(heads), _i(hd), _lib.ptr(wins["q_off"]),
                      _lib.ptr(wins["q_cnt"]), _lib.ptr(wins["k_off"]), _lib.ptr(wins["k_cnt"]), _lib.ptr(q), _lib.ptr(kv),
                      _lib.ptr(O), _lib.ptr(lse), _lib.ptr(dO.contiguous()), _lib.ptr(dq) if q.shape[0] else None,
                      _lib.ptr(dkv), _lib.stream())
        else:
            dq.zero_()
        return dq, dkv, None, None, None


def pair_attention(q, kv, wins, heads, hd):
    """q (R, cg) scaled queries, kv (Kn, 2 cg) = [K | V]; `wins` = int32 q_off / q_cnt / k_off / k_cnt per window
    (every key row belongs to exactly one window).  Returns O (R, cg)."""
    return _PairAttention.apply(q, kv, wins, heads, hd)


# ---------------------------------------------------------------------------------------------------------
# nn.Linear with a deterministic split-K weight gradient (csrc/linear_wgrad.hip)
# ---------------------------------------------------------------------------------------------------------

_wgrad_ws = {}


def _wgrad_workspace(dev, M, cin, cout):
    need = int(_lib.lib().mssvt_linear_wgrad_workspace_floats(_i(M), _i(cin), _i(cout)))
    ws = _wgrad_ws.get(dev)
    if ws is None or ws.numel() < need:
        ws = _wgrad_ws[dev] = torch.empty((need,), dtype=torch.float32, device=dev)
    return ws


def wgrad_supported(cin, cout):
    return cin % 4 == 0 and cout % 4 == 0 and ((cout + 15) // 16) * ((cin + 15) // 16 + 1) <= 160


LINEAR_ROWS = {(64, 64)}  # (cin, cout) shapes that run on mssvt_linear_rows instead of the library GEMM (measured: the only win)


# ... and the (cin, cout) shapes that run on mssvt_linear_rows_h (csrc/linear_rows_h.hip: split-fp16 operands, rows and weight
# matrix normalised by powers of two inside the kernel): the Blocks' 128 <-> 256 FFN, 64 <-> 128 to_kvs, 128 x 128 pos_proj.2 --
# the library GEMMs that were 24 % of the training step's GPU time (profiles/r04_f_train_step_kernel_stats.csv)
LINEAR_ROWS_H = {(128, 256), (256, 128), (64, 128), (128, 64), (128, 128)} if os.environ.get("MSSVT_LINEAR_ROWS_H", "1") != "0" else set()


def _linear_rows(x, w, transpose_w, b, relu, n_out, scale=1.0, split16=False):
    M, K = x.shape
    y = torch.empty((M, n_out), dtype=torch.float32, device=x.device)
    if M:
        _lib.call("mssvt_linear_rows_h" if split16 else "mssvt_linear_rows", _i(M), _i(K), _i(n_out), _lib.ptr(x), _i(K), _lib.ptr(w),
                  _i(1 if transpose_w else 0), _lib.ptr(b), _i(1 if relu else 0), ctypes.c_float(scale), _lib.ptr(y), _i(n_out),
                  _lib.stream())
    return y


class _Linear(torch.autograd.Function):
    """y = x W^T + b (optionally relu'd in place) over compact rows.  y and dx are library GEMMs on the large shapes (a
    hand-written fp32-MFMA kernel with the weight matrix in LDS runs at 0.4 - 0.7 of the library's rate there) and
    mssvt_linear_rows (csrc/linear_rows.hip) on 64 x 64, where the library picks poor tiles (13 vs 8 us at 33k rows, 154 vs
    34 at 132k); dW = dY^T X and db reduce over the ROWS into a <= 256 x 128 result, which library GEMMs do at 7 % of the
    matrix pipe: one split-K MFMA launch here."""

    @staticmethod
    def forward(ctx, x, w, b, relu=False, scale=1.0):
        cout, cin = w.shape
        if ((cin, cout) in LINEAR_ROWS or (cin, cout) in LINEAR_ROWS_H) and x.dtype == torch.float32:
            x = x.contiguous()
            y = _linear_rows(x, w.detach().contiguous(), False, None if b is None else b.detach(), relu, cout, scale,
                             split16=(cin, cout) in LINEAR_ROWS_H)
        else:
            y = F.linear(x, w, b)
            if relu:
                y = y.clamp_(min=0)
            if scale != 1.0:
                y = y.mul_(scale)
        ctx.save_for_backward(x, w, y if relu else None)
        ctx.has_bias, ctx.scale = b is not None, float(scale)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        if y is not None:
            dy = torch.ops.aten.threshold_backward(dy, y, 0.0)  # relu: the gradient where the output is positive
        dy = dy.contiguous()
        cout, cin = w.shape
        sc = ctx.scale  # y = sc * act(...): dx and dW / db carry the factor (a relu'd output is saved scaled: same sign)
        dx = None
        if ctx.needs_input_grad[0]:
            if (cout, cin) in LINEAR_ROWS or (cout, cin) in LINEAR_ROWS_H:
                dx = _linear_rows(dy, w.detach().contiguous(), True, None, False, cin, sc, split16=(cout, cin) in LINEAR_ROWS_H)
            else:
                dx = dy @ w
                if sc != 1.0:
                    dx = dx.mul_(sc)
        dw = db = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            x = x.contiguous()
            M = x.shape[0]
            alloc = torch.empty if M > 0 else torch.zeros
            dw = alloc(w.shape, dtype=w.dtype, device=w.device)
            db = alloc((cout,), dtype=w.dtype, device=w.device) if ctx.has_bias else None
            if M > 0:
                _lib.call("mssvt_linear_wgrad", _i(M), _i(cin), _i(cout), _lib.ptr(x), _lib.ptr(dy), _lib.ptr(dw),
                          _lib.ptr(db), _lib.ptr(_wgrad_workspace(x.device, M, cin, cout)), _lib.stream())
                if sc != 1.0:  # (cout x cin and cout elements: two small launches instead of two over the rows)
                    dw.mul_(sc)
                    if db is not None:
                        db.mul_(sc)
        return dx, dw, db, None, None


def linear(mod, x, relu=False, scale=1.0):
    """``scale * mod(x)`` (``scale * relu(mod(x))``) for an nn.Linear (or a weight / bias pair) with the deterministic weight
    gradient."""
    w, b = (mod.weight, mod.bias) if hasattr(mod, "weight") else mod
    if w.dim() == 3:
        w = w.squeeze(-1)  # Conv1d(k = 1)
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and wgrad_supported(w.shape[1], w.shape[0])
            and torch.is_grad_enabled() and (w.requires_grad or (b is not None and b.requires_grad))):
        y = F.relu(F.linear(x, w, b)) if relu else F.linear(x, w, b)
        return y * scale if scale != 1.0 else y
    return _Linear.apply(x, w, b, relu, float(scale))


class _LayerNorm(torch.autograd.Function):
    """nn.LayerNorm over the channels of compact rows: forward k_layer_norm, backward k_layer_norm_bwd (csrc/rowops.hip;
    x and dy read once, column sums in a fixed order).  Returns (x, y): the caller routes the OTHER uses of x (the
    residual connection) through the returned x, so that their gradient arrives here and is added inside the backward
    kernel instead of by an accumulation pass of autograd's."""

    @staticmethod
    def forward(ctx, x, w, b, eps):
        x = x.contiguous()
        y = torch.empty_like(x)
        _lib.call("mssvt_layer_norm", _lib.ptr(x), _i(x.shape[0]), _i(x.shape[1]), _lib.ptr(w), _lib.ptr(b),
                  ctypes.c_float(eps), _lib.ptr(y), _lib.stream())
        ctx.save_for_backward(x, w)
        ctx.eps = eps
        return x.view_as(x), y

    @staticmethod
    def backward(ctx, dres, dy):
        x, w = ctx.saved_tensors
        N, C = x.shape
        if dy is None:  # the normalised branch was not used
            return dres, torch.zeros_like(w), torch.zeros_like(w), None
        dy = dy.contiguous()
        dres = None if dres is None else dres.contiguous()
        dx, dw, db = torch.empty_like(x), torch.empty_like(w), torch.empty_like(w)
        ws = _wgrad_ws.get(("ln", x.device))
        if ws is None:
            ws = _wgrad_ws[("ln", x.device)] = torch.empty((512 * 2 * 256,), dtype=torch.float32, device=x.device)
        _lib.call("mssvt_layer_norm_backward_residual", _lib.ptr(x), _lib.ptr(dy), _lib.ptr(dres), _i(N), _i(C), _lib.ptr(w),
                  ctypes.c_float(ctx.eps), _lib.ptr(dx), _lib.ptr(dw), _lib.ptr(db), _lib.ptr(ws), _lib.stream())
        return dx, dw, db, None


def layer_norm(norm, x):
    """``norm(x)`` for an nn.LayerNorm over the last dimension of (N, C) rows."""
    return layer_norm_residual(norm, x)[1]


def layer_norm_residual(norm, x):
    """``(x, norm(x))``: use the returned x for every other use of x (see _LayerNorm)."""
    C = x.shape[1]
    if (C not in fused.LN_WIDTHS or x.dtype != torch.float32 or not x.is_cuda or x.shape[0] == 0 or not torch.is_grad_enabled()
            or norm.weight is None or norm.bias is None):
        return x, norm(x)
    return _LayerNorm.apply(x, norm.weight, norm.bias, float(norm.eps))


def ffn(block, x):
    """block._ffn with the deterministic weight gradients (ref mssvt_backbone.py:341-343)."""
    return ffn_residual(block, x)[1]


def ffn_residual(block, x):
    """``(x, block._ffn(x))``: the returned x is the one to add the result to (its gradient joins norm2's backward)."""
    x, xn = layer_norm_residual(block.norm2, x)
    if isinstance(block.activation, torch.nn.ReLU):  # the clamp rides in the first product's epilogue
        h = linear(block.linear1, xn, relu=True)
    else:
        h = block.activation(linear(block.linear1, xn))
    return x, linear(block.linear2, block.dropout1(h))


def add_drop_path(block, x, z):
    """``x + block.drop_path(z)`` in one launch when DropPath is active (the module's own row mask as a factor)."""
    dp = block.drop_path
    p = float(getattr(dp, "drop_prob", 0.0))
    if not (getattr(dp, "training", False) and p > 0.0):
        return x + z
    keep = 1.0 - p
    mask = z.new_empty((z.shape[0],) + (1,) * (z.dim() - 1)).bernoulli_(keep)
    if keep > 0.0 and getattr(dp, "scale_by_keep", True):
        mask.div_(keep)
    return torch.addcmul(x, z, mask)


# ---------------------------------------------------------------------------------------------------------
# tokens = gathered features + positional embedding (csrc/train_tok.hip)
# ---------------------------------------------------------------------------------------------------------

TOK_WIDTHS = (16, 32, 64, 128, 256)
KEY_SETS = True  # False: the key sets of a plan from framework ops (mask / nonzero / gathers), kept for A/B and K > 64
TOKENS = True  # False: the autograd composition (gather_sum + _pos6), kept for A/B and for widths outside TOK_WIDTHS


class _Tokens(torch.autograd.Function):
    """All token sets of a Block in one node: ``tok_k = xhat[rows_k][:, c0_k:c1_k] + relu(pos_proj(geo_k))[:, c0_k:c1_k]``
    for every set k of `parts` (dicts: rows int32 (M), geo (M, 8), csr = Csr.gather(rows, N), c0, c1).  One launch per
    set forward; backward = one strided segmented sum per set straight into its columns of d xhat, one weight-gradient
    slab per set and one ordered reduce (ref pos_proj mssvt_backbone.py:43-47, token sums :270-285)."""

    @staticmethod
    def forward(ctx, xhat, w6, b6, parts):
        xhat, w, b = xhat.contiguous(), w6.detach().contiguous(), b6.detach().contiguous()
        N, C = xhat.shape
        outs = []
        for pt in parts:
            M, c0, cg = pt["rows"].numel(), pt["c0"], pt["c1"] - pt["c0"]
            tok = torch.empty((M, cg), dtype=torch.float32, device=xhat.device)
            if M:
                _lib.call("mssvt_train_tok_forward", _i(M), _i(C), _i(c0), _i(cg), _lib.ptr(pt["rows"]), _lib.ptr(xhat),
                          _lib.ptr(pt["geo"]), _lib.ptr(w), _lib.ptr(b), _lib.ptr(tok), _lib.stream())
            outs.append(tok)
        ctx.save_for_backward(w, b)
        ctx.parts, ctx.shape, ctx.wshape = parts, (N, C), tuple(w6.shape)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        w, b = ctx.saved_tensors
        parts = ctx.parts
        N, C = ctx.shape
        dev = w.device
        grads = [torch.zeros((pt["rows"].numel(), pt["c1"] - pt["c0"]), dtype=torch.float32, device=dev) if g is None
                 else g.contiguous() for g, pt in zip(grads, parts)]
        dx = None
        if ctx.needs_input_grad[0]:
            ranges = sorted(set((pt["c0"], pt["c1"]) for pt in parts))
            tiled = ranges[0][0] == 0 and ranges[-1][1] == C and all(a[1] == b_[0] for a, b_ in zip(ranges, ranges[1:]))
            dx = (torch.empty if tiled else torch.zeros)((N, C), dtype=torch.float32, device=dev)
            seen = set()
            for pt, g in zip(parts, grads):
                key = (pt["c0"], pt["c1"])
                _sum_into(pt["csr"].bwd, g, dx, pt["c0"], key in seen)
                seen.add(key)
        dw = db = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            n = len(parts)
            Ms = [pt["rows"].numel() for pt in parts]
            sizes = [int(_lib.lib().mssvt_train_tok_slab_floats(_i(M), _i(pt["c1"] - pt["c0"]))) for M, pt in zip(Ms, parts)]
            slab = torch.empty((max(1, sum(sizes)),), dtype=torch.float32, device=dev)
            addr, at = [], 0
            for pt, g, M, sz in zip(parts, grads, Ms, sizes):
                a = slab.data_ptr() + 4 * at
                addr.append(a)
                at += sz
                if M:
                    _lib.call("mssvt_train_tok_backward_partial", _i(M), _i(pt["c0"]), _i(pt["c1"] - pt["c0"]), _lib.ptr(pt["geo"]),
                              _lib.ptr(w), _lib.ptr(b), _lib.ptr(g), ctypes.c_void_p(a), _lib.stream())
            dw = torch.empty((C, 6), dtype=torch.float32, device=dev)
            db = torch.empty((C,), dtype=torch.float32, device=dev)
            arr = ctypes.c_int * n
            _lib.call("mssvt_train_tok_backward_reduce", _i(C), _i(n), arr(*Ms), arr(*[pt["c0"] for pt in parts]),
                      arr(*[pt["c1"] - pt["c0"] for pt in parts]), (ctypes.c_void_p * n)(*addr), _lib.ptr(dw), _lib.ptr(db),
                      _lib.stream())
            dw = dw.reshape(ctx.wshape)
        return dx, dw, db, None


def tokens(xhat, conv, parts):
    """Token sets of `parts` (see _Tokens) for a Conv1d(6, C, 1) positional embedding `conv`."""
    return _Tokens.apply(xhat, conv.weight, conv.bias, parts)


def tokens_supported(block, C):
    conv = block.pos_proj[0]
    return (TOKENS and all(cg in TOK_WIDTHS for cg in block.ms_attn.scale_dims) and conv.bias is not None
            and tuple(conv.weight.shape[:2]) == (C, 6) and len(block.ms_attn.scale_dims) <= 2)


# ---------------------------------------------------------------------------------------------------------
# Block
# ---------------------------------------------------------------------------------------------------------

def block_supported(block, sp):
    ma = block.ms_attn
    return (sp.features.is_cuda and sp.features.dtype == torch.float32 and block.win2_size is not None
            and len(ma.num_heads) == 2 and sp.features.shape[1] % 4 == 0 and all(d % 4 == 0 for d in ma.scale_dims)
            and max(block.max_num_win1, block.max_num_win2) < 2048 and max(block.win2_size) <= 120
            and max(block.win1_size) <= 60 and _pair_attention_covers(ma))


def _pair_attention_covers(ma):
    """Shapes csrc/pair_attn.hip is instantiated for: <= 128 channels per head group, head dimension 4 .. 64 (a power of
    two); anything else trains through the operator path instead of failing inside mssvt_pair_attention_fwd."""
    return max(ma.scale_dims) <= 128 and ma.per_head_dim in (4, 8, 16, 32, 64)


@torch.no_grad()
def _plan_index_sets(block, sp, p):
    """What the Blocks of a plan share: the window count, the compact key rows of both scales (their inverted indices,
    geometry, per-window ranges) and the query rows' count of EVERY query pattern of the plan's Blocks -- all sizes in
    ONE host read, before the first Block runs: the later Blocks of the plan add no host wait of their own."""
    common = getattr(p, "train_common", None)
    if common is not None and block.cbs_pattern in common["pats"]:
        return common
    dev = sp.indices.device
    N = sp.indices.shape[0]
    group = [b for b in (getattr(sp, "_plan_group", None) or ()) if b.plan_key() == block.plan_key()
             and fused._qmeta(b, p) is not None]
    pats = {}
    for b in [block] + group:
        if b.cbs_pattern not in pats:
            q_ind, nq, owner_q = fused._query(b, p)
            pats[b.cbs_pattern] = dict(q_ind=q_ind, nq=nq, owner_q=owner_q, od=fused._work_order(b, p, nq, N))
    order = sorted(pats)
    # valid key slots of each scale: counted per window on the device (windows past num_wins hold stale memory and count 0)
    kdim = [int(p.kmeta[g].shape[1]) for g in range(2)]
    fast = KEY_SETS and all(K <= 64 for K in kdim)
    if fast:
        totals = torch.zeros(2, dtype=torch.int32, device=dev)
        cnts = [torch.empty(p.cap, dtype=torch.int32, device=dev) for _ in range(2)]
        for g in range(2):
            _lib.call("mssvt_train_key_counts", _i(p.cap), _i(kdim[g]), _lib.ptr(p.num_wins), _lib.ptr(p.kmeta[g]), _lib.ptr(cnts[g]),
                      ctypes.c_void_p(totals.data_ptr() + 4 * g), _lib.stream())
        words = [p.num_wins.reshape(1), totals]
    else:
        in_use = torch.arange(p.cap, device=dev).unsqueeze(1) < p.num_wins
        valid_g, rows_g = [], []
        for g in range(2):
            rows = p.kmeta[g][..., 3].contiguous().view(torch.int32)  # (cap, K)
            rows_g.append(rows)
            valid_g.append(((rows >= 0) & in_use).reshape(-1))
        words = [p.num_wins.reshape(1)] + [v.sum().int().reshape(1) for v in valid_g]
    host = _read_sizes(torch.cat(words + [pats[k]["od"]["n_rows"].reshape(1) for k in order]))
    nw = host[0]
    for k, R in zip(order, host[3:]):
        pats[k]["R"] = R
    centre = p.wcentre[:nw, :3].contiguous()
    pad2 = torch.zeros((1, 2), dtype=torch.float32, device=dev)
    keys = []  # valid key slots of each scale, window-major (FPS pick order inside a window)
    for g, Kg in enumerate(host[1:3]):
        km = p.kmeta[g]
        K = kdim[g]
        if fast:  # two launches: prefix of the counts, then every valid slot to its place (csrc/train_tok.hip)
            nk = cnts[g][:nw]
            koff = (torch.cumsum(nk, 0, dtype=torch.int32) - nk).contiguous()
            k_rows = torch.empty(Kg, dtype=torch.int32, device=dev)
            k_win = torch.empty(Kg, dtype=torch.int32, device=dev)
            k_geo = torch.empty((Kg, 8), dtype=torch.float32, device=dev)
            if nw and Kg:
                _lib.call("mssvt_train_key_compact", _i(nw), _i(K), _lib.ptr(km), _lib.ptr(p.wcentre), _lib.ptr(koff),
                          _lib.ptr(k_rows), _lib.ptr(k_win), _lib.ptr(k_geo), _lib.stream())
            keys.append(dict(k_rows=k_rows, k_win=k_win, k_geo=k_geo, k_csr=Csr.gather(k_rows, N), k_off=koff,
                             k_cnt=nk.contiguous()))
            continue
        flat = _nonzero_known(valid_g[g], Kg)
        k_win = flat // K
        nk = valid_g[g].reshape(p.cap, K)[:nw].sum(1)  # keys per window
        koff = torch.cumsum(nk, 0) - nk
        k_rows = rows_g[g].reshape(-1)[flat].contiguous()
        k_rel = km[..., :3].reshape(-1, 3)[flat].contiguous()
        k_geo = torch.cat([k_rel, centre[k_win], pad2.expand(k_rel.shape[0], 2)], dim=1).contiguous()
        keys.append(dict(k_rows=k_rows, k_win=k_win, k_geo=k_geo, k_csr=Csr.gather(k_rows, N),
                         k_off=koff.int().contiguous(), k_cnt=nk.int().contiguous()))
    _deferred.extend(k["k_csr"].bwd for k in keys)
    common = p.train_common = dict(nw=nw, centre=centre, keys=keys, pats=pats, pad2=pad2)
    p.train_sets = {}
    return common


@torch.no_grad()
def _block_index_sets(block, sp, p):
    """Compact index sets of (plan, cbs_pattern): cached on the plan, shared by the Blocks that use it."""
    common = _plan_index_sets(block, sp, p)
    cache = p.train_sets
    key = (block.cbs_pattern, 1 if block.use_feature_interpolation else 0)
    if key in cache:
        return cache[key]
    dev = sp.indices.device
    N = sp.indices.shape[0]
    pat = common["pats"][block.cbs_pattern]
    q_ind, nq, owner_q, od, R, nw = pat["q_ind"], pat["nq"], pat["owner_q"], pat["od"], pat["R"], common["nw"]
    s = {"nw": nw, "R": R, "nq": nq, "centre": common["centre"]}
    qs = pat.get("q_sets")
    if qs is None:  # the query rows of this pattern (shared by its with / without interpolation variants)
        q_rows = torch.empty(R, dtype=torch.int32, device=dev)
        q_geo = torch.empty((R, 8), dtype=torch.float32, device=dev)
        if R:
            _lib.call("mssvt_train_query_sets", _i(R), _lib.ptr(od["row_meta"]), _lib.ptr(od["row_src"]), _lib.ptr(p.wcentre),
                      _lib.ptr(q_rows), _lib.ptr(q_geo), _lib.stream())
        qs = {"q_rows": q_rows, "q_geo": q_geo, "q_rel": q_geo[:, :3], "q_centre": q_geo[:, 3:6]}
        # (a voxel is on one window's query list when every window size is odd, ref mssvt_backbone.py:94-97, AND no
        # offset of a custom table leaves the window: then the inverse of the gather is one scatter, else a sort)
        distinct = all(int(wsz) % 2 == 1 for wsz in block.win1_size) and fused._lists_disjoint(block)
        qs["q_csr"] = _gather_unique(q_rows, N) if distinct else Csr.gather(q_rows, N)
        if qs["q_csr"].bwd.pending is not None:
            _deferred.append(qs["q_csr"].bwd)
        qs["keys"] = [dict(k, wins=dict(q_off=od["q_off"][:nw].contiguous(), q_cnt=od["nq_valid"][:nw].contiguous(),
                                        k_off=k["k_off"], k_cnt=k["k_cnt"])) for k in common["keys"]]
        pat["q_sets"] = qs
    s.update(qs)
    # interpolation / scatter table: 3 compact attention rows + weights per voxel (row R = the zero row)
    interp = key[1]
    upd_ind, n_upd, owner = (p.ind_win1, block.max_num_win1, p.owner_win1) if interp else (q_ind, nq, owner_q)
    zero_row = p.cap * nq  # a virtual row of the (never allocated) padded attention buffer
    tab_row = torch.full((max(N, 1), 4), -1, dtype=torch.int32, device=dev)
    tab_w = torch.zeros((max(N, 1), 4), dtype=torch.float32, device=dev)
    _lib.call("mssvt_block_interp_table", _i(nq), _i(n_upd), _i(interp), _lib.ptr(sp.indices), _lib.ptr(p.win_ind),
              _lib.ptr(p.num_wins), _i(p.cap), _lib.ptr(p.win_vstart), _lib.ptr(q_ind), _lib.ptr(upd_ind),
              _lib.ptr(owner), fused._f3(sp.voxel_size), fused._f3(sp.point_cloud_range[0:3]), _i(zero_row),
              _lib.ptr(tab_row), _lib.ptr(tab_w), _lib.stream())
    inv = torch.full((zero_row + 1,), R, dtype=torch.int32, device=dev)  # padded attention row -> compact row
    inv[od["row_src"][:R, 1].long()] = torch.arange(R, dtype=torch.int32, device=dev)
    owned = torch.empty(N, dtype=torch.bool, device=dev)
    idx3 = torch.empty(3 * N, dtype=torch.int32, device=dev)
    w3 = torch.empty(3 * N, dtype=torch.float32, device=dev)
    _lib.call("mssvt_train_interp_compact", _i(N), _i(R), _lib.ptr(inv), _lib.ptr(tab_row), _lib.ptr(tab_w), _lib.ptr(idx3),
              _lib.ptr(w3), ctypes.c_void_p(owned.data_ptr()), _lib.stream())
    off3 = torch.arange(0, 3 * N + 1, 3, dtype=torch.int32, device=dev)
    s["interp_csr"] = Csr(off3, idx3, w3, R + 1, drop_src=R, fwd_longest=3)
    s["owned"] = owned
    # the longest-list words of the four inverted indices ride with the next host read (_read_sizes), or are read on
    # first use: no host wait of their own
    _deferred.append(s["interp_csr"].bwd)
    cache[key] = s
    return s


BLOCK_TAIL = os.environ.get("MSSVT_TRAIN_BLOCK_TAIL", "1") != "0"  # 0: the autograd composition (gather_sum + where + DropPath + add)


class _BlockTail(torch.autograd.Function):
    """new = x_in + DropPath(where(owned, interpolated attention rows, x_in)) in one launch (mssvt_segment_sum_rows_residual:
    ref mssvt_backbone.py:298-340): new[v] = row_a[v] x_in[v] + row_b[v] sum_e w_e attn[idx_e], row_b = the row's DropPath
    factor, row_a = 1 on a voxel the attention updates and 1 + row_b elsewhere (its list holds zero weights only).
    Backward: d attn = the inverted index's segmented sum of row_b-scaled gradient rows, d x_in = row_a-scaled rows."""

    @staticmethod
    def forward(ctx, attn_ext, x_in, csr, row_a, row_b):
        attn_ext, x_in = attn_ext.contiguous(), x_in.contiguous()
        N, C = x_in.shape
        new = torch.empty_like(x_in)
        f = csr.fwd
        _lib.call("mssvt_segment_sum_rows_residual", _i(C), _i(N), _lib.ptr(f.start), _lib.ptr(f.end), _lib.ptr(f.idx),
                  _lib.ptr(f.w), _lib.ptr(attn_ext), _lib.ptr(x_in), _lib.ptr(row_a), _lib.ptr(row_b), _lib.ptr(new),
                  _lib.stream())
        ctx.csr = csr
        ctx.save_for_backward(row_a, row_b)
        return new

    @staticmethod
    def backward(ctx, g):
        row_a, row_b = ctx.saved_tensors
        d_attn = ctx.csr.bwd.sum(g * row_b.unsqueeze(1)) if ctx.needs_input_grad[0] else None
        d_x = g * row_a.unsqueeze(1) if ctx.needs_input_grad[1] else None
        return d_attn, d_x, None, None, None


def block_tail(block, attn_ext, x_in, s):
    """The Block's update of the voxel features up to the FFN: see _BlockTail."""
    dp = block.drop_path
    N = x_in.shape[0]
    p = float(getattr(dp, "drop_prob", 0.0))
    if getattr(dp, "training", False) and p > 0.0:  # the module's own mask (DropPath.forward), kept as a row factor
        keep = 1.0 - p
        row_b = x_in.new_empty((N, 1)).bernoulli_(keep)
        if keep > 0.0 and getattr(dp, "scale_by_keep", True):
            row_b.div_(keep)
        row_b = row_b.reshape(N)
        row_a = torch.where(s["owned"], torch.ones_like(row_b), row_b + 1.0)
    else:
        if "tail_rows" not in s:  # constant per index set
            ones = torch.ones((N,), dtype=torch.float32, device=x_in.device)
            s["tail_rows"] = (torch.where(s["owned"], ones, ones + 1.0), ones)
        row_a, row_b = s["tail_rows"]
    return _BlockTail.apply(attn_ext, x_in, s["interp_csr"], row_a, row_b)


def _pos6(conv, rel, centre, c0=None, c1=None):
    """relu(Conv1d(6 -> C, 1)([rel ; centre])) on compact rows (ref pos_proj, mssvt_backbone.py:43-47).  The six input
    columns are padded to eight so that the weight gradient (rows x 6 -> C x 6: the library's slowest shape, 370 us per
    call) goes through the split-K kernel as well."""
    w, b = conv.weight.squeeze(-1), conv.bias
    if c0 is not None:
        w, b = w[c0:c1], b[c0:c1]
    x = torch.cat([rel, centre, rel.new_zeros((rel.shape[0], 2))], dim=1)
    return F.relu(linear((F.pad(w, (0, 2)), b), x))


def block_forward(block, sp):
    """Differentiable forward of a MixedScaleSparseTransformerBlock on compact rows (ref mssvt_backbone.py:201-346)."""
    if not block_supported(block, sp) or sp.features.shape[0] == 0:
        return block.forward_ops(sp)
    x_in = sp.features
    N, C = x_in.shape
    x_in, xhat = layer_norm_residual(block.norm1, x_in)  # (the later uses of x_in send their gradient through norm1's backward)
    p = fused.two_scale_plan(block, sp)
    s = _block_index_sets(block, sp, p)
    _compress_ahead(sp)
    R, ma = s["R"], block.ms_attn
    hd = ma.per_head_dim
    if R > 0:
        fast = tokens_supported(block, C)
        if fast:  # the four token sets (queries and keys of both head groups) as one autograd node
            parts, c0 = [], 0
            for g, cg in enumerate(ma.scale_dims):
                parts.append(dict(rows=s["q_rows"], geo=s["q_geo"], csr=s["q_csr"], c0=c0, c1=c0 + cg))
                c0 += cg
            c0 = 0
            for g, cg in enumerate(ma.scale_dims):
                ks = s["keys"][g]
                parts.append(dict(rows=ks["k_rows"], geo=ks["k_geo"], csr=ks["k_csr"], c0=c0, c1=c0 + cg))
                c0 += cg
            toks = tokens(xhat, block.pos_proj[0], parts)
        else:
            tok_q = gather_sum(xhat, s["q_csr"]) + _pos6(block.pos_proj[0], s["q_rel"], s["q_centre"])
        outs, c0 = [], 0
        ng = len(ma.scale_dims)
        for g, (heads, cg) in enumerate(zip(ma.num_heads, ma.scale_dims)):
            c1 = c0 + cg
            ks = s["keys"][g]
            if fast:
                tok_qg, tok_k = toks[g], toks[ng + g]
            else:
                tok_k = gather_sum(xhat[:, c0:c1].contiguous(), ks["k_csr"]) + _pos6(
                    block.pos_proj[0], ks["k_geo"][:, :3], ks["k_geo"][:, 3:6], c0, c1)
                tok_qg = tok_q[:, c0:c1].contiguous()
            q = linear(ma.to_qs[g], tok_qg, scale=ma.scale)  # (R, cg), scaled in the product's epilogue
            kv = linear(ma.to_kvs[g], tok_k)  # (Kg, 2 cg) = [K | V]
            o = pair_attention(q, kv, ks["wins"], heads, hd)  # (R, cg)
            outs.append(linear(ma.projs[g], o.reshape(R, cg)))
            c0 = c1
        attn = torch.cat(outs + [x_in.new_zeros((R, C - c0))] if c0 < C else outs, dim=1)
    else:
        attn = x_in.new_zeros((0, C))
    attn_ext = torch.cat([attn, attn.new_zeros((1, C))], dim=0)  # row R = zeros (empty slots, zero weights)
    if BLOCK_TAIL and s["interp_csr"].fwd.heavy is None and s["interp_csr"].fwd.pending is None:
        new = block_tail(block, attn_ext, x_in, s)
    else:
        upd = gather_sum(attn_ext, s["interp_csr"])  # (N, C): interpolated / scattered update of every owned voxel
        feats = torch.where(s["owned"].unsqueeze(1), upd, x_in)  # untouched voxels keep x_in (ref :317-338)
        new = block.drop_path(feats) + x_in
    new, z = ffn_residual(block, new)
    new = add_drop_path(block, new, block.dropout1(z))
    if hasattr(block, "out_linear"):
        new = linear(block.out_linear, new)
    sp.features = new
    sp.gather_dict = None
    sp._xhat = None
    return sp


# ---------------------------------------------------------------------------------------------------------
# CompressBlock
# ---------------------------------------------------------------------------------------------------------

def compress_supported(block, sp):
    ma = block.ms_attn
    return (sp.features.is_cuda and sp.features.dtype == torch.float32 and ma.num_head_groups == 1
            and sp.features.shape[1] % 4 == 0 and len(block.pos_proj) >= 3 and _pair_attention_covers(ma)
            and fused._table_covers_window(block))


@torch.no_grad()
def _compress_index_sets(block, sp, p):
    dev = sp.indices.device
    N, nw, ns = sp.indices.shape[0], p.nw, block.max_num_win1
    if KEY_SETS and ns <= 64 and nw > 0:  # two launches (csrc/train_tok.hip) instead of ~35 framework ones
        total = torch.zeros(1, dtype=torch.int32, device=dev)
        cnt = torch.empty(nw, dtype=torch.int32, device=dev)
        _lib.call("mssvt_train_list_counts", _i(nw), _i(ns), _lib.ptr(p.k_ind), _lib.ptr(cnt), _lib.ptr(total), _lib.stream())
        P, = _read_sizes(total)  # one sync: the pair count (+ the pending words of the Blocks' index sets)
        koff = (torch.cumsum(cnt, 0, dtype=torch.int32) - cnt).contiguous()
        pair_vox = torch.empty(P, dtype=torch.int32, device=dev)
        pair_win = torch.empty(P, dtype=torch.int32, device=dev)
        geo = torch.empty((P, 8), dtype=torch.float32, device=dev)
        if P:
            _lib.call("mssvt_train_pairs_compact", _i(nw), _i(ns), _lib.ptr(p.k_ind), _lib.ptr(p.win_vstart), _lib.ptr(koff),
                      _lib.ptr(sp.indices), _lib.ptr(p.win_ind), fused._f3(sp.voxel_size), fused._f3(sp.point_cloud_range[0:3]),
                      fused._f3(p.win_size_m), _lib.ptr(pair_vox), _lib.ptr(pair_win), _lib.ptr(geo), _lib.stream())
        wins = dict(q_off=torch.arange(nw, dtype=torch.int32, device=dev), q_cnt=torch.ones(nw, dtype=torch.int32, device=dev),
                    k_off=koff, k_cnt=cnt)
        vox_csr = _gather_unique(pair_vox, N) if p.disjoint else Csr.gather(pair_vox, N)
        return dict(cnt=cnt.long(), vox_csr=vox_csr, wins=wins, rel=geo[:, :3], pair_centre=geo[:, 3:6],
                    full=(cnt >= ns))
    k = p.k_ind[:nw]
    valid = k >= 0
    P, = _read_sizes(valid.sum().reshape(1))  # one sync: the pair count (+ the pending words of the Blocks' index sets)
    flat = _nonzero_known(valid.reshape(-1), P)
    pair_win = flat // ns
    pair_vox = (k.reshape(-1)[flat].long() + p.win_vstart[:nw].long()[pair_win]).int().contiguous()
    cnt = valid.sum(1)
    vox_xyz = _metric(sp.indices, sp.point_cloud_range, sp.voxel_size)
    centre = _metric(p.win_ind[:nw], sp.point_cloud_range, p.win_size_m)
    wins = dict(q_off=torch.arange(nw, dtype=torch.int32, device=dev), q_cnt=torch.ones(nw, dtype=torch.int32, device=dev),
                k_off=(torch.cumsum(cnt, 0) - cnt).int().contiguous(), k_cnt=cnt.int().contiguous())
    # disjoint windows: a voxel is on one list, once -> the inverse is one scatter; a custom table whose offsets leave
    # the window puts a voxel on several lists -> the general inverted index (every contribution kept)
    vox_csr = _gather_unique(pair_vox, N) if p.disjoint else Csr.gather(pair_vox, N)
    return dict(cnt=cnt, vox_csr=vox_csr, wins=wins, rel=(vox_xyz[pair_vox.long()] - centre[pair_win]).contiguous(),
                pair_centre=centre[pair_win], full=(cnt >= ns))


def _compress_sets(block, sp):
    """(plan, index sets) of the CompressBlock that closes this level -- built ahead by the level's first Block when there
    is one (`_compress_ahead`), else here: K2 + K4 on the device and the host reads of the output shape / pair count."""
    ahead = getattr(sp, "_train_cmp", None)
    if ahead and ahead[0] is block and ahead[1] is sp.indices:
        st = getattr(sp, "_level", None)
        if st is not None and len(st.get("status_words", [])) > ahead[4]:
            st["status_checked"] = False
            fused.check_level_status(sp)  # plans made after the early read: their overflow words
        return ahead[2], ahead[3]
    p = fused.one_scale_plan(block, sp)
    return p, _compress_index_sets(block, sp, p)


def _compress_ahead(sp):
    """The CompressBlock's plan and index sets depend on the voxel indices only: the level's first Block builds them, so
    that their host reads happen while the queue is still short -- a read after the Blocks waits for all of them and
    leaves the device idle until the host has caught up again (1.6 ms per step went there)."""
    cmp_blk = getattr(sp, "_next_compress", None)
    if cmp_blk is None or getattr(sp, "_train_cmp", None) is not None:
        return
    sp._train_cmp = ()
    C = sp.features.shape[1]
    if not (compress_supported(cmp_blk, sp) and cmp_blk.linear1.in_features == C):
        return
    p = fused.one_scale_plan(cmp_blk, sp)
    st = getattr(sp, "_level", None)
    sp._train_cmp = (cmp_blk, sp.indices, p, _compress_index_sets(cmp_blk, sp, p),
                     len(st.get("status_words", [])) if st is not None else 0)


def _metric(indices, point_cloud_range, cell):
    from .mssvt_backbone import metric_centres
    return metric_centres(indices, point_cloud_range, cell)


def compress_forward(block, sp):
    """Differentiable forward of a MixedScaleSparseTransformerCompressBlock (ref mssvt_backbone.py:351-398)."""
    if not compress_supported(block, sp) or sp.features.shape[0] == 0:
        return block.forward_ops(sp)
    x = layer_norm(block.norm1, sp.features)
    C = x.shape[1]
    p, s = _compress_sets(block, sp)
    ma = block.ms_attn
    heads, hd = ma.num_heads[0], ma.per_head_dim
    xk = gather_sum(x, s["vox_csr"])  # (P, C) window-major key features
    # query = channel-wise max over the zero padded list (ref :370): the zeros take part unless the list is full
    q_tok = torch.segment_reduce(xk, "max", lengths=s["cnt"], unsafe=True)
    q_tok = torch.where(s["full"].unsqueeze(1), q_tok, torch.clamp(q_tok, min=0.0))
    pos = _pos6(block.pos_proj[0], s["rel"], s["pair_centre"])
    pos = F.relu(linear(block.pos_proj[2], pos))
    tok_k = xk + pos
    q = linear(ma.to_qs[0], q_tok, scale=ma.scale)  # (nw, C)
    kv = linear(ma.to_kvs[0], tok_k)  # (P, 2C)
    o = pair_attention(q, kv, s["wins"], heads, hd)  # (nw, C)
    new = linear(ma.projs[0], o.reshape(-1, C))
    new, z = ffn_residual(block, new)
    new = new + block.dropout1(z)  # no residual to the block input (ref :383-385)
    if hasattr(block, "out_linear"):
        new = linear(block.out_linear, new)
    sp.features = new
    sp._xhat = None
    return fused._compress_finish(sp, p, new)
