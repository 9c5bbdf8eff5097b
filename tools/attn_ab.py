"""A/B of the flash-attention workgroup size in the development library (csrc/experimental/attention.hip):
128-query workgroups (4 waves, what the shipped library runs) against 256-query workgroups (VSX_FLASH_WAVES=8), at the
UNet's self-attention shapes.  Interleaved in one process; the two must agree bit for bit.

    VSX_LIB_VARIANT=next python tools/attn_ab.py [--batch 2] > gpurun_out/attn_ab.txt"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoswap_amd import ops  # noqa: E402


def r(*s):
    return torch.randn(*s, device='cuda', dtype=torch.float32).to(torch.float16)


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
    ap.add_argument('--batch', type=int, default=2)
    ap.add_argument('--rounds', type=int, default=5)
    ap.add_argument('--reps', type=int, default=4)
    args = ap.parse_args()
    nb = 16 * args.batch
    print(f'# self-attention, {nb} images x 8 heads; median of {args.rounds} rounds x {args.reps} launches; us')
    print(f'{"shape":28s} {"waves=4":>9s} {"waves=8":>9s}  TF/s(4)  TF/s(8)  speedup  identical')
    for n, c in ((4096, 320), (1024, 640), (2304, 320), (5376, 320)):      # 64x64, 32x32, 48x48, 56x96 latents
        d = c // 8
        torch.manual_seed(0)
        q, k, vt = r(nb, n, c), r(nb, n, c), r(nb, c, n)
        fn = lambda: ops.attention(q, k, vt, 8, d ** -0.5)      # noqa: E731
        outs, ts = {}, {4: [], 8: []}
        for w in (4, 8):
            os.environ['VSX_FLASH_WAVES'] = str(w)
            outs[w] = fn()
        torch.cuda.synchronize()
        same = torch.equal(outs[4], outs[8])
        for _ in range(args.rounds):
            for w in (4, 8):
                os.environ['VSX_FLASH_WAVES'] = str(w)
                ts[w].append(time_once(fn, args.reps))
        med = {w: sorted(x)[len(x) // 2] * 1000.0 for w, x in ts.items()}
        flop = 4.0 * nb * 8 * n * n * d
        print(f'N={n:5d} d={d:3d}              {med[4]:9.1f} {med[8]:9.1f}  {flop / med[4] / 1e6:7.1f}  '
              f'{flop / med[8] / 1e6:7.1f}  {med[4] / med[8]:6.2f}x  {same}', flush=True)
    os.environ.pop('VSX_FLASH_WAVES', None)


if __name__ == '__main__':
    main()
