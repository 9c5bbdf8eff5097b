"""Which kernel configuration is fastest for the small-M GEMMs of the 16x16 / 8x8 levels (M = 1024 ... 8192, K, N >= 1280)?
Times every configuration the library can be forced into (option "tile_tune": tile, ring depth, K slices; option "gemm_pp":
persistent kernel with 128- / 256-row tiles) against the automatic choice, in one process, the order rotating round by round.

    python tools/small_m_sweep.py [--rounds 8] > gpurun_out/small_m_sweep.txt
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoswap_amd import ops  # noqa: E402

DEV, H16 = 'cuda', torch.float16

# name -> (gemm_pp, tile_tune)
CONFIGS = [('auto', 1, 0), ('128x160', 0, 2), ('128x160d', 0, 2 + 16), ('128x320', 0, 1), ('128x128', 0, 4), ('64x128', 0, 5),
           ('64x64', 0, 6), ('split2', 0, 2 << 8), ('split3', 0, 3 << 8), ('split4', 0, 4 << 8), ('pp128', 3, 0), ('pp256', 2, 0)]


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
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rounds', type=int, default=8)
    ap.add_argument('--reps', type=int, default=6)
    args = ap.parse_args()
    shapes = []
    for M in (1024, 2048, 4096, 8192):
        shapes += [(f'proj 1280->1280 +res M={M}', M, 1280, 1280, True, False), (f'qkv 1280->3840 M={M}', M, 3840, 1280, False, False),
                   (f'qk 1280->2560 M={M}', M, 2560, 1280, False, False), (f'geglu 1280->5120 M={M}', M, 5120, 1280, False, True),
                   (f'ff2 5120->1280 +res M={M}', M, 1280, 5120, True, False)]
    print(f'# median of {args.rounds} rounds x {args.reps} launches, us; order rotates round by round')
    print(f'{"shape":34s} ' + ' '.join(f'{c[0]:>9s}' for c in CONFIGS) + '   best')
    for name, M, N, K, res, geglu in shapes:
        torch.manual_seed(0)
        x, w = r(M, K), r(2 * N if geglu else N, K, scale=K ** -0.5)
        b = r(2 * N if geglu else N)
        rs = r(M, N) if res else None
        fn = lambda: ops.linear(x, w, b, residual=rs, geglu=geglu)
        ts = {c[0]: [] for c in CONFIGS}
        ref = None
        ok = {}
        for c in CONFIGS:
            ops.set_option('gemm_pp', c[1]); ops.set_option('tile_tune', c[2])
            try:
                o = fn()
                torch.cuda.synchronize()
                if ref is None:
                    ref = o.float()
                ok[c[0]] = float((o.float() - ref).norm() / ref.norm()) < 2e-3
            except Exception as e:      # a configuration the shape cannot take
                ok[c[0]] = False
        live = [c for c in CONFIGS if ok[c[0]]]
        for rnd in range(args.rounds):
            k = rnd % len(live)
            for c in live[k:] + live[:k]:
                ops.set_option('gemm_pp', c[1]); ops.set_option('tile_tune', c[2])
                ts[c[0]].append(time_once(fn, args.reps))
        med = {k: (sorted(v)[len(v) // 2] * 1000.0 if v else float('nan')) for k, v in ts.items()}
        best = min((k for k in med if med[k] == med[k]), key=lambda k: med[k])
        print(f'{name:34s} ' + ' '.join(f'{med[c[0]]:9.1f}' for c in CONFIGS) + f'   {best} ({med["auto"] / med[best]:.2f}x)', flush=True)
    ops.set_option('gemm_pp', 1); ops.set_option('tile_tune', 0)


if __name__ == '__main__':
    main()
