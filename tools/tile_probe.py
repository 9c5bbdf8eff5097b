"""Per-CU vs whole-chip operand delivery: one forced tile shape, long K, 8..1024 tiles.  VSX_TUNE_TILE=1 python tools/tile_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoswap_amd import ops  # noqa: E402
from tools.kbench import timeit  # noqa: E402

tile = int(os.environ.get('VSX_TUNE_TILE', '1'))
BM = {1: 128, 2: 128, 3: 256}[tile]
BN = {1: 320, 2: 160, 3: 320}[tile]
K = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
w = torch.randn(320, K, device='cuda', dtype=torch.float16) * 0.01
for tiles_m in (8, 16, 32, 64, 128, 256, 512):
    M = BM * tiles_m
    x = torch.randn(M, K, device='cuda', dtype=torch.float16)
    ms = timeit(lambda: ops.linear(x, w), iters=10, warm=2)
    ntile = tiles_m * (320 // BN)
    staged = ntile * (BM + BN) * 2 * K
    print(f'tile {BM}x{BN} K={K}: {ntile:5d} tiles  {ms * 1e3:8.1f} us  {2.0 * M * 320 * K / ms / 1e9:7.1f} TF/s  '
          f'staged {staged / ms / 1e6:8.1f} GB/s total, {staged / ms / 1e6 / min(ntile, 256):7.1f} GB/s per busy CU', flush=True)
