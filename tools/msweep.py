"""K=N=320 projection time vs M (cache-resident to HBM-resident) + a plain device copy for the streaming ceiling."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoswap_amd import ops  # noqa: E402
from tools.kbench import timeit  # noqa: E402

dev = 'cuda'
for C in (320, 640):
    w = torch.randn(C, C, device=dev, dtype=torch.float16) * 0.02
    for M in (4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288):
        x = torch.randn(M, C, device=dev, dtype=torch.float16)
        y = torch.empty_like(x)
        ms = timeit(lambda: ops.linear(x, w), iters=20, warm=3)
        cp = timeit(lambda: y.copy_(x), iters=20, warm=3)
        ln = timeit(lambda: ops.layer_norm(x, w[0], w[1], 1e-5), iters=20, warm=3)
        print(f'C={C} M={M:7d}: gemm {ms * 1e3:7.1f} us ({2.0 * M * C * C / ms / 1e9:6.1f} TF/s, {4.0 * M * C / ms / 1e6:7.1f} GB/s)'
              f'  copy {cp * 1e3:7.1f} us ({4.0 * M * C / cp / 1e6:7.1f} GB/s)  layernorm {ln * 1e3:7.1f} us', flush=True)
