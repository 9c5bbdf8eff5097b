"""A/B of flash attention with 32 / 64 queries per wave (option attn_qb, csrc/attention.hip) at the UNet's self-attention shapes:
interleaved rounds in ONE process, median per variant, and each variant's error against an fp32 PyTorch reference on a
smaller problem of the same head dim.

    python tools/attn_ab.py [--rounds 7] > gpurun_out/attn_ab.txt"""
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


def problem(nb, n, heads, d, seed=0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    c = heads * d
    q = torch.randn(nb, n, c, device=DEV, generator=g).to(H16)
    k = torch.randn(nb, n, c, device=DEV, generator=g).to(H16)
    v = torch.randn(nb, n, c, device=DEV, generator=g).to(H16)
    vt = v.transpose(1, 2).contiguous()                  # [nb, C, n]: what the V projection's c_mode 1 writes
    return q, k, v, vt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rounds', type=int, default=7)
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--vars', default='1,2', help='values of the option (attn_qb: query blocks per wave)')
    ap.add_argument('--option', default='attn_qb', help='the option to vary: attn_qb, or attn_o16 (--vars 0,1: 32-row / 16-row O^T tiles at d = 40)')
    args = ap.parse_args()
    variants = [int(v) for v in args.vars.replace('+', ',').split(',')]
    tag = {'attn_qb': 'qb', 'attn_o16': 'o16='}.get(args.option, args.option + '=')
    print(f'# flash attention, median of {args.rounds} rounds x {args.reps} launches; us per launch, TFLOP/s of 4*nb*heads*n*n*d')
    for nb, n, heads, d in ((32, 4096, 8, 40), (16, 4096, 8, 40), (32, 1024, 8, 80), (32, 256, 8, 160), (32, 5376, 8, 40)):
        q, k, v, vt = problem(nb, n, heads, d)
        fn = lambda: ops.attention(q, k, vt, heads, d ** -0.5)      # noqa: E731
        ts = {x: [] for x in variants}
        for x in variants:
            ops.set_option(args.option, x)
            fn()
        torch.cuda.synchronize()
        for rnd in range(args.rounds):
            for x in variants[rnd % len(variants):] + variants[:rnd % len(variants)]:
                ops.set_option(args.option, x)
                ts[x].append(time_once(fn, args.reps))
        flop = 4.0 * nb * heads * n * n * d
        med = {x: sorted(t)[len(t) // 2] * 1000.0 for x, t in ts.items()}
        print(f'nb={nb:3d} n={n:5d} d={d:3d}  ' + '  '.join(f'{tag}{x}: {med[x]:8.1f} us {flop / med[x] / 1e6:7.1f} TF/s' for x in variants),
              flush=True)
    # accuracy: every variant against fp32 softmax(QK^T)V on a problem small enough to materialise (incl. a ragged key count)
    for nb, n, heads, d in ((2, 1001, 8, 40), (2, 1024, 8, 80)):
        q, k, v, vt = problem(nb, n, heads, d, seed=3)
        if n % 8:
            vt = torch.nn.functional.pad(vt, (0, 8 - n % 8)).contiguous()
        qh = q.float().view(nb, n, heads, d).transpose(1, 2)
        kh = k.float().view(nb, n, heads, d).transpose(1, 2)
        vh = v.float().view(nb, n, heads, d).transpose(1, 2)
        ref = (torch.softmax(qh @ kh.transpose(-1, -2) * d ** -0.5, -1) @ vh).transpose(1, 2).reshape(nb, n, heads * d)
        errs = []
        for x in variants:
            ops.set_option(args.option, x)
            o = ops.attention(q, k, vt, heads, d ** -0.5).float()
            errs.append(f'{tag}{x}: rel-L2 {float((o - ref).norm() / ref.norm()):.3e} max {float((o - ref).abs().max()):.2e}')
        print(f'accuracy nb={nb} n={n} d={d}: ' + '  '.join(errs), flush=True)
    ops.set_option('attn_qb', 0)
    ops.set_option('attn_o16', 0)


if __name__ == '__main__':
    main()
