"""Run one GEMM/conv shape repeatedly (for rocprofv3 --pmc / --kernel-trace).  python tools/gemm_one.py conv|lin|geglu"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoswap_amd import ops  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else 'conv'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = 'cuda'
if kind == 'conv':
    x = torch.randn(32, 64, 64, 320, device=dev, dtype=torch.float16)
    w = torch.randn(320, 3, 3, 320, device=dev, dtype=torch.float16) * 0.01
    b = torch.randn(320, device=dev, dtype=torch.float16)
    fn = lambda: ops.conv2d(x, w, b)
elif kind == 'conv2':
    x = torch.randn(32, 32, 32, 1280, device=dev, dtype=torch.float16)
    w = torch.randn(640, 3, 3, 1280, device=dev, dtype=torch.float16) * 0.01
    b = torch.randn(640, device=dev, dtype=torch.float16)
    fn = lambda: ops.conv2d(x, w, b)
elif kind == 'lin':
    x = torch.randn(32768, 2560, device=dev, dtype=torch.float16)
    w = torch.randn(640, 2560, device=dev, dtype=torch.float16) * 0.01
    fn = lambda: ops.linear(x, w)
else:
    x = torch.randn(131072, 320, device=dev, dtype=torch.float16)
    w = torch.randn(2560, 320, device=dev, dtype=torch.float16) * 0.02
    b = torch.randn(2560, device=dev, dtype=torch.float16)
    fn = lambda: ops.linear(x, w, b, geglu=True)
for _ in range(n):
    fn()
torch.cuda.synchronize()
