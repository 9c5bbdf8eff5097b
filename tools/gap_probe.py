"""N identical B=2 forwards for a kernel trace: wall per forward vs the sum of kernel durations (inter-kernel idle).

    rocprofv3 --kernel-trace -d out -- python tools/gap_probe.py ; python tools/rocpd_summary.py out/..._results.db
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_pipeline  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N = 6
pipe = build_pipeline(torch.device('cuda'), 16)
x = torch.randn(B, 4, 16, 64, 64, device='cuda', dtype=torch.float16)
txt = torch.randn(B, 77, 768, device='cuda', dtype=torch.float16)
with torch.no_grad():
    for _ in range(2):
        pipe.unet(x, 481, txt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        pipe.unet(x, 481, txt)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
print(f'B={B}: {1000 * (t1 - t0) / N:.2f} ms/forward over {N} forwards (+2 warmup)')
