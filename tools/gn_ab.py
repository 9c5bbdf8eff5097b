"""A/B of the GroupNorm launch sequence at the UNet's shapes: statistics finalized inside the apply kernel (option gn_fuse = 1: two
launches) against the stand-alone finalize kernel (gn_fuse = 0: three), interleaved rounds in ONE process, median per variant.

    python tools/gn_ab.py [--rounds 9] > gpurun_out/gn_ab.txt"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoswap_amd import ops  # noqa: E402

DEV, H16 = 'cuda', torch.float16


def time_once(fn, reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rounds', type=int, default=9)
    ap.add_argument('--reps', type=int, default=20)
    args = ap.parse_args()
    # (images, rows per image, channels, launches per UNet forward at B = 1): the per-frame GroupNorms in front of proj_in
    shapes = [(16, 4096, 320, 10), (32, 4096, 320, 10), (16, 1024, 640, 10), (32, 1024, 640, 10), (16, 256, 1280, 10), (32, 256, 1280, 10),
              (64, 4096, 320, 10), (128, 4096, 320, 10)]
    print(f'# group_norm (statistics + [finalize] + apply), median of {args.rounds} rounds x {args.reps} calls; us per call')
    print(f'{"images x rows x C":24s} {"chunks":>6s} {"3 launches":>11s} {"2 launches":>11s} {"ratio":>7s}')
    tot = [0.0, 0.0]
    for nimg, rows, C, n in shapes:
        x = (torch.randn(nimg, rows, C, device=DEV) * 1.5 + 0.3).to(H16)
        gamma, beta = (torch.randn(C, device=DEV) * 0.3 + 1).to(H16), (torch.randn(C, device=DEV) * 0.2).to(H16)
        fn = lambda: ops.group_norm(x, gamma, beta, 32, 1e-6, nimg)      # noqa: E731
        ts = {0: [], 1: []}
        for v in (0, 1):
            ops.set_option('gn_fuse', v)
            fn()
        torch.cuda.synchronize()
        for rnd in range(args.rounds):
            for v in ((0, 1) if rnd % 2 == 0 else (1, 0)):
                ops.set_option('gn_fuse', v)
                ts[v].append(time_once(fn, args.reps))
        med = {v: sorted(t)[len(t) // 2] * 1000.0 for v, t in ts.items()}
        from videoswap_amd import _lib
        chunks = _lib.load().vsx_groupnorm_chunks(rows, nimg)
        print(f'{f"{nimg} x {rows} x {C}":24s} {chunks:6d} {med[0]:11.1f} {med[1]:11.1f} {med[1] / med[0]:7.3f}', flush=True)
        if nimg <= 32:
            tot[0] += med[0] * n / 2
            tot[1] += med[1] * n / 2
    ops.set_option('gn_fuse', 0)
    print(f'# per pair of UNet forwards (B = 1 + B = 2), the 30 + 30 per-frame GroupNorms listed: {tot[0] / 1e3:.3f} -> {tot[1] / 1e3:.3f} ms')


if __name__ == '__main__':
    main()
