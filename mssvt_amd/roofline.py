"""Roofline accounting of the dominant kernel for bench.py ("Measurement").

``measure`` re-launches the dominant kernel of the current execution path on the bench
inputs, times it with HIP events on the launching stream, and divides the kernel's
ALGORITHMIC bytes (computed from the actual input, not hard-coded) by that time.
"""
import ctypes

import torch

from . import _lib, mssvt_ops

_i = ctypes.c_int


def _gather_two_window(net, vc, batch, event_time_ms, peak_gbs):
    """ops path: K3 (k_gather_two_window).  Algorithmic bytes per launch =
    8 B per in-grid probed offset (one hash slot) + 16 B per window row read
    + 16 B per list entry written (index + 3 offset ints), SURVEY.md 8(d)."""
    blk = net.backbone[0]
    sp_shape = net.grid_size
    cnt = torch.bincount(vc[:, 0].long(), minlength=batch).to(torch.int32)
    table = mssvt_ops.build_hash_table(batch, net.hash_size, sp_shape, vc, cnt)
    wgrid = [sp_shape[i] // blk.win1_size[i] for i in range(3)]
    win, _ = mssvt_ops.get_non_empty_window_center(blk.win1_size, blk.max_num_wins, batch, net.hash_size,
                                                   wgrid, vc)
    t = blk._tables_on(vc.device)
    maxes = (blk.max_num_odd, blk.max_num_even, blk.max_num_win1, blk.max_num_win2)
    nw = win.shape[0]
    inds = [torch.full((nw, m), -1, dtype=torch.int32, device=vc.device) for m in maxes]
    coords = [torch.zeros((nw, m, 3), dtype=torch.int32, device=vc.device) for m in maxes]
    tabs = [t['odd'], t['even'], t['win1'], t['win2']]

    def launch():
        _lib.call("mssvt_gather_two_window_voxels_with_hash", *[_i(v) for v in sp_shape],
                  *[_i(v) for v in blk.win1_size], *[_i(m) for m in maxes], _i(nw), _i(net.hash_size),
                  *[_i(x.shape[0]) for x in tabs], *[_lib.ptr(x) for x in inds], *[_lib.ptr(x) for x in coords],
                  *[_lib.ptr(x) for x in tabs], _lib.ptr(win), _lib.ptr(table), _lib.stream())

    ms = event_time_ms(launch, 20)
    # in-grid probes: window centre + offset inside [0, shape)
    offs = torch.cat(tabs, 0).long()  # (Q,3) x,y,z
    cx = win[:, 3].long() * blk.win1_size[0] + blk.win1_size[0] // 2
    cy = win[:, 2].long() * blk.win1_size[1] + blk.win1_size[1] // 2
    cz = win[:, 1].long() * blk.win1_size[2] + blk.win1_size[2] // 2
    sx = cx[:, None] + offs[None, :, 0]
    sy = cy[:, None] + offs[None, :, 1]
    sz = cz[:, None] + offs[None, :, 2]
    inb = (sx >= 0) & (sx < sp_shape[0]) & (sy >= 0) & (sy < sp_shape[1]) & (sz >= 0) & (sz < sp_shape[2])
    probes = int(inb.sum())
    written = sum(int((x >= 0).sum()) for x in inds)
    alg_bytes = 8 * probes + 16 * nw + 16 * written
    achieved = alg_bytes / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": "k_gather_two_window", "achieved": achieved, "peak": peak_gbs,
            "unit": "GB/s", "frac": achieved / peak_gbs, "traffic": None,
            "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_us": ms * 1e3,
            "units_per_launch": {"windows": nw, "probes": probes, "list_entries": written}}


def measure(net, vc, feats, batch, event_time_ms, peak_gbs, live=None):
    impl = net.backbone[0].impl
    if impl == "fused":
        from . import fused
        return fused.roofline(net, vc, feats, batch, event_time_ms, peak_gbs, live=live)
    return _gather_two_window(net, vc, batch, event_time_ms, peak_gbs)
