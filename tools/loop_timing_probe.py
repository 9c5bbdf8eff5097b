#!/usr/bin/env python
"""Why did tests/test_fullwidth_gpu.py time the product's 50+50 loop at 18.2 s when bench.py times 4.19 s per clip
?  The probe times the same product loop (n + n steps, T = 16, 64x64) in one process

  A  fresh (only libvsx has run),
  B  after the PyTorch-ROCm fp32 + fp16 eager oracles ran n + n steps in the same process (rocBLAS / MIOpen / the
     caching allocator have state now) — the order the test uses,
  C  after torch.cuda.empty_cache(),
  D  with the allocator's statistics around a loop (hipMalloc / hipFree retries per loop),

and prints wall time, hipEvent time and allocator counters for each.  The fp16 oracle time is also the same-box reading
of the reference's GPU-style path (`torch_rocm_eager_fp16` in bench.py).

    python tools/loop_timing_probe.py [--steps 10]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def timed(fn):
    torch.cuda.synchronize()
    s0 = torch.cuda.memory_stats()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    s1 = torch.cuda.memory_stats()
    return {'wall_s': round(wall, 3), 'event_s': round(e0.elapsed_time(e1) / 1e3, 3),
            'device_mallocs': s1['num_device_alloc'] - s0['num_device_alloc'],
            'device_frees': s1['num_device_free'] - s0['num_device_free'],
            'alloc_retries': s1['num_alloc_retries'] - s0['num_alloc_retries'],
            'reserved_GB': round(s1['reserved_bytes.all.current'] / 2 ** 30, 2),
            'allocated_GB': round(s1['allocated_bytes.all.current'] / 2 ** 30, 2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'loop_timing_probe.json'))
    args = ap.parse_args()
    import test_fullwidth_gpu as T
    cfg, _, ora_dev, ora_h, prod = T.build_models()
    x, txt = T._inputs(1, 16, 64, 64, seed=157)
    neg = torch.randn(1, 77, 768, generator=torch.Generator().manual_seed(7))
    n = args.steps
    res = {'steps': n}
    T._product_loops(prod, x, txt, neg, 1)                       # first-call costs (packing, module load) out of the way
    res['A_product_fresh'] = timed(lambda: T._product_loops(prod, x, txt, neg, n))
    res['A2_product_fresh_again'] = timed(lambda: T._product_loops(prod, x, txt, neg, n))
    res['oracle_fp32_torch'] = timed(lambda: T._oracle_loops(ora_dev, x, txt, neg, n))
    res['oracle_fp16_torch'] = timed(lambda: T._oracle_loops(ora_h, x, txt, neg, n))
    res['B_product_after_torch_oracles'] = timed(lambda: T._product_loops(prod, x, txt, neg, n))
    res['B2_product_again'] = timed(lambda: T._product_loops(prod, x, txt, neg, n))
    torch.cuda.empty_cache()
    res['C_product_after_empty_cache'] = timed(lambda: T._product_loops(prod, x, txt, neg, n))
    res['C2_product_again'] = timed(lambda: T._product_loops(prod, x, txt, neg, n))
    for k, v in res.items():
        print(k, json.dumps(v) if isinstance(v, dict) else v)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, 'w') as f:
        json.dump(res, f, indent=1)


if __name__ == '__main__':
    main()
