"""Fixed cost against per-slab cost of the small-M GEMMs (M <= 16 384, the 16x16 / 8x8 levels and the B = 1 half of the loop):
T(K) at fixed (M, N) for K = 64 ... 5120 under the automatic dispatch and under forced tile configurations.  A launch whose time
does not shrink with K is paying for something else than its K loop (launch ramp, first-touch misses, the epilogue's residual read
and store tail); the slope between two K values is the steady-state cost of a 64-wide slab.

    python tools/k_sweep.py > gpurun_out/k_sweep.txt
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoswap_amd import ops  # noqa: E402

DEV, H16 = 'cuda', torch.float16
CONFIGS = [('auto', 1, 0), ('128x160', 0, 2), ('128x160d', 0, 2 + 16), ('128x320', 0, 1), ('64x64', 0, 6), ('pp128', 3, 0), ('pp256', 2, 0)]
KS = (64, 320, 1280, 2560, 5120)


def r(*s, scale=1.0):
    return (torch.randn(*s, device=DEV, dtype=torch.float32) * scale).to(H16)


def time_once(fn, reps):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    rounds, reps = 7, 8
    print(f'# us per launch, median of {rounds} rounds x {reps} back-to-back launches (+res epilogue); "copy" = torch add of the '
          'residual-sized tensor (reads 2, writes 1 tensor of M x N: the epilogue\'s own traffic as a stand-alone pass)')
    print(f'{"M x N":14s} {"config":10s} ' + ' '.join(f'{"K=" + str(k):>9s}' for k in KS) + '   us/slab(1280->5120)   T(0) extrapolated')
    for M, N in ((1024, 1280), (2048, 1280), (4096, 1280), (8192, 1280), (16384, 640), (4096, 3840)):
        res = r(M, N)
        bias = r(N)
        a, b = r(M, N), r(M, N)
        cp = sorted(time_once(lambda: torch.add(a, b, out=res), reps) for _ in range(rounds))[rounds // 2]
        print(f'{f"{M}x{N}":14s} {"copy":10s} {cp:9.1f}')
        xs = {k: r(M, k) for k in KS}
        ws = {k: r(N, k, scale=k ** -0.5) for k in KS}
        for name, pp, tune in CONFIGS:
            ops.set_option('gemm_pp', pp)
            ops.set_option('tile_tune', tune)
            row = []
            for k in KS:
                fn = lambda: ops.linear(xs[k], ws[k], bias, residual=res)
                try:
                    fn()
                    torch.cuda.synchronize()
                    row.append(sorted(time_once(fn, reps) for _ in range(rounds))[rounds // 2])
                except Exception:
                    row.append(float('nan'))
            slope = (row[4] - row[2]) / ((5120 - 1280) / 64)
            t0 = row[2] - slope * (1280 / 64)
            print(f'{"":14s} {name:10s} ' + ' '.join(f'{v:9.1f}' for v in row) + f'   {slope:8.3f}              {t0:8.1f}', flush=True)
    ops.set_option('gemm_pp', 1)
    ops.set_option('tile_tune', 0)


if __name__ == '__main__':
    main()
