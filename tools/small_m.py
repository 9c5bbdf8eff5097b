"""Small-M projections (the 16x16 / 8x8 UNet levels at B=1): time per tile choice.  VSX_TUNE_TILE=n python tools/small_m.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoswap_amd import ops  # noqa: E402
from tools.kbench import timeit  # noqa: E402

for (M, N, K) in [(4096, 1280, 1280), (1024, 1280, 1280), (8192, 1280, 1280), (2048, 1280, 1280), (4096, 3840, 1280),
                  (1024, 3840, 1280), (4096, 1280, 5120), (16384, 640, 640), (16384, 1920, 640)]:
    x = torch.randn(M, K, device='cuda', dtype=torch.float16)
    w = torch.randn(N, K, device='cuda', dtype=torch.float16) * 0.02
    b = torch.randn(N, device='cuda', dtype=torch.float16)
    r = torch.randn(M, N, device='cuda', dtype=torch.float16)
    ms = timeit(lambda: ops.linear(x, w, b, residual=r), iters=20, warm=3)
    print(f'tile {os.environ.get("VSX_TUNE_TILE", "auto"):4s} M={M:6d} N={N:5d} K={K:5d}: {ms * 1e3:7.1f} us  {2.0 * M * N * K / ms / 1e9:7.1f} TF/s', flush=True)
