"""Host-side enqueue time of one UNet forward (python + ctypes + allocator) vs its GPU time."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_pipeline  # noqa: E402

pipe = build_pipeline(torch.device('cuda'), 16)
for B in (1, 2):
    x = torch.randn(B, 4, 16, 64, 64, device='cuda', dtype=torch.float16)
    txt = torch.randn(B, 77, 768, device='cuda', dtype=torch.float16)
    with torch.no_grad():
        for _ in range(2):
            pipe.unet(x, 481, txt)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            pipe.unet(x, 481, txt)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print(f'B={B}: host enqueue {1000 * (t1 - t0) / 5:.1f} ms/forward, total {1000 * (t2 - t0) / 5:.1f} ms/forward')
import cProfile, pstats
with torch.no_grad():
    pr = cProfile.Profile(); pr.enable()
    for _ in range(3):
        pipe.unet(x, 481, txt)
    pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
