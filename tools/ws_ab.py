"""A/B of the weight-stationary K = N = 320 kernel (csrc/gemm_pp.hip: gemm_ws320_kernel, option gemm_ws) against the product's dispatch
(the persistent ping-pong kernel) at the UNet's 64 x 64-level projections: bias / + residual / + row statistics, interleaved rounds in
one process, bit-for-bit comparison of the outputs.

    python tools/ws_ab.py [--rounds 7] > gpurun_out/ws_ab.txt"""
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
    ap.add_argument('--rounds', type=int, default=7)
    ap.add_argument('--reps', type=int, default=6)
    ap.add_argument('--waves', type=int, default=10, help='option ws_waves: 10 waves of 32 columns or 5 of 64')
    args = ap.parse_args()
    ops.set_option('ws_waves', args.waves)
    print(f'# ws_waves = {args.waves}')
    print(f'# K = N = 320 projections, median of {args.rounds} interleaved rounds x {args.reps} launches; us per launch; GB/s of the algorithmic bytes (A + C [+ residual])')
    print(f'{"shape":44s} {"persistent":>11s} {"weight-st.":>11s} {"ratio":>6s} {"GB/s":>7s} {"GB/s":>7s}  outputs')
    g = torch.Generator(device=DEV).manual_seed(1)
    for M in (131072, 65536, 524288):
        x = torch.randn(M, 320, device=DEV, generator=g).to(H16)
        w = (torch.randn(320, 320, device=DEV, generator=g) * 320 ** -0.5).to(H16)
        b = torch.randn(320, device=DEV, generator=g).to(H16)
        res = torch.randn(M, 320, device=DEV, generator=g).to(H16)
        for kind in ('bias', '+res', 'bias +stats', '+res +stats'):
            kw = dict(residual=res if '+res' in kind else None, row_stats='stats' in kind)
            fn = lambda: ops.linear(x, w, b, **kw)          # noqa: E731
            outs, parts, ts = {}, {}, {0: [], 2: []}
            for v in (0, 2):
                ops.set_option('gemm_ws', v)
                y = fn()
                outs[v] = y.clone()
                parts[v] = getattr(y, '_vsx_rowparts', None)
            torch.cuda.synchronize()
            same = torch.equal(outs[0], outs[2])
            note = 'bit-identical' if same else 'DIFFERENT (rel-L2 %.2e)' % float((outs[0].float() - outs[2].float()).norm() / outs[0].float().norm())
            if 'stats' in kind:
                have = [p is not None for p in parts.values()]
                if all(have):
                    s0, s1 = (p.float().reshape(M, -1, 2).sum(1) if torch.is_tensor(p) else None for p in parts.values())
                    if s0 is not None and s1 is not None:
                        note += ', statistics sums rel %.1e' % float((s0 - s1).abs().max() / s0.abs().max())
                else:
                    note += f', rowparts present: {have}'
            for rnd in range(args.rounds):
                for v in ((0, 2) if rnd % 2 == 0 else (2, 0)):
                    ops.set_option('gemm_ws', v)
                    ts[v].append(time_once(fn, args.reps))
            med = {v: sorted(t)[len(t) // 2] * 1e3 for v, t in ts.items()}
            byt = M * 320 * 2 * (3 if '+res' in kind else 2)
            print(f'{"M=%d 320->320 %s" % (M, kind):44s} {med[0]:11.1f} {med[2]:11.1f} {med[0] / med[2]:6.2f} {byt / med[0] / 1e3:7.0f} {byt / med[2] / 1e3:7.0f}  {note}',
                  flush=True)
        del x, res
    # the LayerNorm-folded projections of the level (round 6: column slices): qk 320 -> 640, qkv 320 -> 960 with the temporal positional row vector
    print(f'{"shape":44s} {"persistent":>11s} {"weight-st.":>11s} {"ratio":>6s} {"GB/s":>7s} {"GB/s":>7s}  outputs')
    for M in (131072, 65536, 524288):
        x = torch.randn(M, 320, device=DEV, generator=g).to(H16)
        gam, bet = torch.randn(320, device=DEV, generator=g).to(H16), torch.randn(320, device=DEV, generator=g).to(H16)
        pe = torch.randn(24, 320, device=DEV, generator=g).to(H16)
        for N, kind in ((640, 'LN fold (qk)'), (960, 'LN fold + PE row vector (qkv)'), (320, 'LN fold (q)')):
            w = (torch.randn(N, 320, device=DEV, generator=g) * 320 ** -0.5).to(H16)
            b = torch.randn(N, device=DEV, generator=g).to(H16)
            kw = dict(pe=pe, rows_per_frame=4096, frames=16) if 'PE' in kind else {}
            ln = ops.DeferredLN(x, gam, bet, 1e-5, **kw)
            ln.stats()
            fn = lambda: ops.linear(ln, w, b)          # noqa: E731
            outs, ts = {}, {0: [], 2: []}
            for v in (0, 2):
                ops.set_option('gemm_ws', v)
                outs[v] = fn().clone()
            torch.cuda.synchronize()
            same = torch.equal(outs[0], outs[2])
            note = 'bit-identical' if same else 'DIFFERENT (rel-L2 %.2e)' % float((outs[0].float() - outs[2].float()).norm() / outs[0].float().norm())
            for rnd in range(args.rounds):
                for v in ((0, 2) if rnd % 2 == 0 else (2, 0)):
                    ops.set_option('gemm_ws', v)
                    ts[v].append(time_once(fn, args.reps))
            med = {v: sorted(t)[len(t) // 2] * 1e3 for v, t in ts.items()}
            byt = M * (320 + N) * 2
            print(f'{"M=%d 320->%d %s" % (M, N, kind):44s} {med[0]:11.1f} {med[2]:11.1f} {med[0] / med[2]:6.2f} {byt / med[0] / 1e3:7.0f} {byt / med[2] / 1e3:7.0f}  {note}',
                  flush=True)
        del x
    ops.set_option('gemm_ws', 1)
    ops.set_option('ws_waves', 10)


if __name__ == '__main__':
    main()
