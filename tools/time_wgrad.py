"""Time mssvt_linear_wgrad against the library GEMM dY^T X + column sum (investigation helper)."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mssvt_amd import _lib, train_path  # noqa: E402

_i = ctypes.c_int
dev = torch.device("cuda", 0)
for M, cin, cout in [(74270, 128, 256), (74270, 256, 128), (33000, 64, 64), (185000, 64, 128), (74270, 128, 128),
                     (600000, 128, 256)]:
    x, dy = torch.randn(M, cin, device=dev), torch.randn(M, cout, device=dev)
    dw, db = torch.empty(cout, cin, device=dev), torch.empty(cout, device=dev)
    ws = train_path._wgrad_workspace(dev, M, cin, cout)

    def mine():
        _lib.call("mssvt_linear_wgrad", _i(M), _i(cin), _i(cout), _lib.ptr(x), _lib.ptr(dy), _lib.ptr(dw), _lib.ptr(db),
                  _lib.ptr(ws), _lib.stream())

    def lib():
        return dy.t() @ x, dy.sum(0)

    res = []
    for fn in (mine, lib):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 * 1e3)
    flop = 2.0 * M * cin * cout
    print("M=%d %d->%d: split-K %.1f us (%.1f TFLOP/s), library %.1f us" % (M, cin, cout, res[0], flop / res[0] / 1e6, res[1]))
