"""A/B of the temporal-attention kernel's output path at the UNet's shapes (T = 16, head dims 40 / 80 / 160): staged through LDS
(default: 16-byte stores, the heads of a frame contiguous) against the store from the accumulator layout (option temporal_out = 1:
8 bytes into each of 32 rows per instruction), interleaved in one process; bit-identical outputs.

    python tools/temporal_ab.py [--batch 8] [--rounds 7]"""
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
    return a.elapsed_time(b) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--rounds', type=int, default=7)
    ap.add_argument('--reps', type=int, default=4)
    args = ap.parse_args()
    B, F = args.batch, 16
    print(f'# UNet batch {B}, {F} frames; median of {args.rounds} rounds x {args.reps} launches, us; bytes = q, k, v read + o written')
    print(f'{"level":28s} {"direct":>9s} {"staged":>9s}  speedup   TB/s staged')
    for hw, c, heads in ((64 * 64, 320, 8), (32 * 32, 640, 8), (16 * 16, 1280, 8), (8 * 8, 1280, 8)):
        d = c // heads
        g = torch.Generator(device=DEV).manual_seed(hw)
        qkv = torch.randn(B * F * hw, 3 * c, device=DEV, generator=g).to(H16)
        q, k, v = qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:]
        fn = lambda: ops.temporal_attention(q, k, v, B, F, F, hw, heads, d ** -0.5)
        outs, ts = {}, {0: [], 1: []}
        for mode in (1, 0):
            ops.set_option('temporal_out', mode)
            outs[mode] = fn()
        torch.cuda.synchronize()
        assert torch.equal(outs[0], outs[1]), 'the two output paths must agree bit for bit'
        for rnd in range(args.rounds):
            for mode in ((0, 1) if rnd % 2 else (1, 0)):
                ops.set_option('temporal_out', mode)
                ts[mode].append(time_once(fn, args.reps))
        med = {m: sorted(x)[len(x) // 2] for m, x in ts.items()}
        nbytes = 4 * B * F * hw * c * 2
        print(f'{f"{int(hw ** 0.5)}x{int(hw ** 0.5)} C={c} d={d}":28s} {med[1]:9.1f} {med[0]:9.1f}  {med[1] / med[0]:6.2f}x   {nbytes / med[0] / 1e6:6.2f}', flush=True)
    ops.set_option('temporal_out', 0)


if __name__ == '__main__':
    main()
