"""Per-(kernel, shape) time breakdown of ONE UNet forward at the headline shapes (B=2 CFG, T=16, 64x64 latent).

Wraps every videoswap_amd.ops entry point with torch.cuda.Event pairs on the launch stream and aggregates by
(op, shape signature).  Usage: python tools/shape_prof.py [--batch 2] [--frames 16] [--latent 64] [--top 40]
"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoswap_amd import ops  # noqa: E402

RECORDS = []


def sig(name, args, kwargs):
    def s(t):
        return 'x'.join(map(str, t.shape)) if torch.is_tensor(t) else ''
    if name == 'gemm':
        d = args[0]
        mode = 'conv%dx%d%s%s' % (d.ks, d.ks, '/s2' if d.stride == 2 else '', '+up' if d.upsample else '') if d.a_mode else 'plain'
        extra = ('geglu ' if d.geglu else '') + ('vt ' if d.c_mode else '') + (f'batch{d.batch0 * d.batch1} ' if d.batch0 * d.batch1 > 1 else '')
        return f'{mode} {extra}M={d.M} N={d.N} K={d.K}', 2.0 * d.M * d.N * (2 if d.geglu else 1) * d.K * d.batch0 * d.batch1
    if name == 'attention':
        q, k = args[0], args[1]
        heads = args[3]
        nb, nq, C = q.shape
        nk = k.shape[1]
        return f'nb={nb} nq={nq} nk={nk} d={C // heads}', 4.0 * nb * nq * nk * C
    return ' '.join(s(a) for a in args if torch.is_tensor(a))[:60], 0.0


def wrap(name):
    fn = getattr(ops, name)

    def inner(*args, **kwargs):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        key, flop = sig(name, args, kwargs)
        a.record()
        out = fn(*args, **kwargs)
        b.record()
        RECORDS.append((name, key, flop, a, b))
        return out
    setattr(ops, name, inner)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=2)
    ap.add_argument('--frames', type=int, default=16)
    ap.add_argument('--latent', type=int, default=64)
    ap.add_argument('--top', type=int, default=45)
    args = ap.parse_args()
    from bench import build_pipeline
    pipe = build_pipeline(torch.device('cuda'), args.frames)
    x = torch.randn(args.batch, 4, args.frames, args.latent, args.latent, device='cuda', dtype=torch.float16)
    txt = torch.randn(args.batch, 77, 768, device='cuda', dtype=torch.float16)
    with torch.no_grad():
        pipe.unet(x, 481, txt)          # warm
        torch.cuda.synchronize()
        for n in ('gemm', 'attention', 'temporal_attention', 'group_norm', 'layer_norm', 'silu', 'axpy', 'pack_latents',
                  'unpack_latents'):
            wrap(n)
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        pipe.unet(x, 481, txt)
        t1.record()
    torch.cuda.synchronize()
    agg = collections.OrderedDict()
    for name, key, flop, a, b in RECORDS:
        ms = a.elapsed_time(b)
        e = agg.setdefault((name, key), [0, 0.0, 0.0])
        e[0] += 1; e[1] += ms; e[2] += flop
    total = sum(v[1] for v in agg.values())
    print(f'forward wall {t0.elapsed_time(t1):.2f} ms; sum of op times {total:.2f} ms; {len(RECORDS)} op calls')
    byop = collections.Counter()
    for (name, key), v in agg.items():
        byop[name] += v[1]
    print('by op:', ', '.join(f'{k} {v:.2f} ms' for k, v in byop.most_common()))
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])[:args.top]
    for (name, key), (n, ms, flop) in rows:
        tf = f'{flop / ms / 1e9:7.1f} TF/s' if flop else ''
        print(f'{ms:8.3f} ms {100 * ms / total:5.1f}%  x{n:<3d} {name:18s} {key:52s} {tf}')


if __name__ == '__main__':
    main()
