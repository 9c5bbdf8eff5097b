"""Calibration of the persistent GEMM's main loop on SQUARE problems: gemm_pp_kernel<2, 0, plain> on
4096 x 4160 x 4096 and 8192 x 8320 x 8192 fp16 (N a multiple of the 320-column tile), uniform random [-1, 1) operands, beside
the guide's plain-HIP 256^2 8-phase template (cdna_hip_programming.md §5: ~1.32-1.34 PF/s at 4096^3, ~1.47 at 8192^3 on the
same kind of data).  Also: the workgroup-per-tile 256x320 kernel on the same problems, zero-filled operands (the DVFS
give-back the guide describes), and the UNet's own long-K shape for reference.

    python tools/gemm_square.py > gpurun_out/gemm_square.txt
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoswap_amd import ops  # noqa: E402

DEV, H16 = 'cuda', torch.float16


def uni(*s):
    return (torch.rand(*s, device=DEV, dtype=torch.float32) * 2.0 - 1.0).to(H16)


def time_med(fn, reps=5, rounds=7):
    ts = []
    for _ in range(rounds):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    return sorted(ts)[len(ts) // 2]


def main():
    print('# persistent kernel (gemm_pp = 2 forces it), tile kernel 256x320 (gemm_pp 0, tile_tune 3); median of 7 rounds x 5 launches')
    print(f'{"problem":44s} {"data":8s} {"kernel":12s} {"us":>9s} {"TF/s":>8s} {"of 2.5 PF":>9s}')
    # 4096 x 5120 and 8192 x 10240: 256 / 1024 tiles of 256 x 320 = 1 / 4 per CU (4160 / 8320 columns leave 208 / 832 tiles: 81 % of the
    # CU-rounds busy — the first table of profiles/r05_gemm_square_calibration.txt)
    for (M, N, K) in ((4096, 5120, 4096), (8192, 10240, 8192), (4096, 4160, 4096), (8192, 8320, 8192), (8192, 8320, 2880), (131072, 320, 2880), (32768, 1280, 2560)):
        for data in ('uniform', 'zeros'):
            if data == 'uniform':
                x, w = uni(M, K), uni(N, K)
            else:
                x, w = torch.zeros(M, K, device=DEV, dtype=H16), torch.zeros(N, K, device=DEV, dtype=H16)
            out = torch.empty(M, N, device=DEV, dtype=H16)
            for name, pp, tune in (('persistent', 2, 0), ('tile 256x320', 0, 3)):
                ops.set_option('gemm_pp', pp)
                ops.set_option('tile_tune', tune)
                fn = lambda: ops.linear(x, w, None, out=out)
                fn()
                ms = time_med(fn)
                tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
                print(f'{f"M={M} N={N} K={K}":44s} {data:8s} {name:12s} {ms * 1e3:9.1f} {tf:8.0f} {tf / 2500:9.3f}', flush=True)
            del x, w, out
    ops.set_option('gemm_pp', 1)
    ops.set_option('tile_tune', 0)


if __name__ == '__main__':
    main()
