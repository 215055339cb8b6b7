#!/usr/bin/env python3
"""mssvt_linear_rows_h (split-fp16 row-streaming linear, csrc/linear_rows_h.hip) against the library GEMM on the training
path's shapes: python tools/time_linear.py"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mssvt_amd import train_path  # noqa: E402


def t_ms(fn, it=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


for M, K, N in [(74270, 128, 256), (74270, 256, 128), (600000, 64, 128), (600000, 128, 64), (19307, 128, 128)]:
    x = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") / K ** 0.5
    b = torch.randn(N, device="cuda")
    dy = torch.randn(M, N, device="cuda")
    a = t_ms(lambda: train_path._linear_rows(x, w, False, b, True, N, 1.0, split16=True))
    lib = t_ms(lambda: F.linear(x, w, b).clamp_(min=0))
    at = t_ms(lambda: train_path._linear_rows(dy, w, True, None, False, K, 1.0, split16=True))
    libt = t_ms(lambda: dy @ w)
    gb = (M * (K + N) * 4) / 1e9
    print("M %6d  %3d -> %3d: split-fp16 %.1f us (%.2f TB/s) vs library %.1f us | dx: %.1f us vs %.1f us" %
          (M, K, N, a * 1e3, gb / a / 1e-3 / 1e3 * 1e-3 * 1e3, lib * 1e3, at * 1e3, libt * 1e3))
