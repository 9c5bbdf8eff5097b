"""Time the residual-epilogue projections (out-proj / FF2 shapes) of every UNet level.  python tools/res_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoswap_amd import ops  # noqa: E402
from tools.kbench import timeit, line  # noqa: E402

dev = 'cuda'
for (M, C) in [(131072, 320), (32768, 640), (8192, 1280), (65536, 320), (16384, 640), (4096, 1280)]:
    for K in (C, 4 * C):
        x = torch.randn(M, K, device=dev, dtype=torch.float16)
        w = torch.randn(C, K, device=dev, dtype=torch.float16) * 0.02
        b = torch.randn(C, device=dev, dtype=torch.float16)
        res = torch.randn(M, C, device=dev, dtype=torch.float16)
        ms = timeit(lambda: ops.linear(x, w, b, residual=res), iters=20, warm=3)
        line(f'linear+res M={M} N={C} K={K}', ms, 2.0 * M * C * K, 2.0 * (M * K + 2 * M * C + C * K))
        ms = timeit(lambda: ops.linear(x, w, b), iters=20, warm=3)
        line(f'linear     M={M} N={C} K={K}', ms, 2.0 * M * C * K, 2.0 * (M * K + M * C + C * K))
