"""Micro-benchmark of the libvsx kernels at the UNet's real shapes (B=2 CFG, T=16, 64x64 latent).

    python tools/kbench.py            # prints one line per kernel/shape: time, TFLOP/s or GB/s

Timing: torch.cuda.Event on the current stream (the stream the kernels are launched on), median of N.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoswap_amd import ops  # noqa: E402

DEV = 'cuda'
H16 = torch.float16


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


def r(*s):
    return torch.randn(*s, device=DEV, dtype=H16)


def line(name, ms, flop=None, bytes_=None):
    extra = ''
    if flop:
        extra += f'  {flop / ms / 1e9:8.1f} TFLOP/s ({flop / ms / 1e9 / 2500 * 100:5.1f}% of 2.5 PF)'
    if bytes_:
        extra += f'  {bytes_ / ms / 1e6:8.1f} GB/s ({bytes_ / ms / 1e6 / 8000 * 100:5.1f}% of 8 TB/s)'
    print(f'{name:58s} {ms:9.3f} ms{extra}', flush=True)


def main():
    BF = 32
    levels = [(64, 320), (32, 640), (16, 1280), (8, 1280)]
    print('== conv3x3 (implicit GEMM) ==')
    for (hw, c) in levels:
        for cin in sorted({c, 2 * c}):
            x = r(BF, hw, hw, cin)
            w = r(c, 3, 3, cin) * 0.01
            b = r(c)
            ms = timeit(lambda: ops.conv2d(x, w, b))
            line(f'conv3x3 {hw}x{hw} Cin={cin} Cout={c}', ms, 2.0 * BF * hw * hw * c * 9 * cin)
    print('== linear / GEGLU ==')
    for (hw, c) in levels:
        M = BF * hw * hw
        x = r(M, c)
        w = r(c, c) * 0.02
        b = r(c)
        ms = timeit(lambda: ops.linear(x, w, b, residual=x))
        line(f'linear M={M} {c}->{c} (+bias+res)', ms, 2.0 * M * c * c)
        w8 = r(8 * c, c) * 0.02
        b8 = r(8 * c)
        ms = timeit(lambda: ops.linear(x, w8, b8, geglu=True))
        line(f'GEGLU  M={M} {c}->{4 * c}', ms, 2.0 * M * c * 8 * c)
        x4 = r(M, 4 * c)
        w4 = r(c, 4 * c) * 0.02
        ms = timeit(lambda: ops.linear(x4, w4, b, residual=x))
        line(f'linear M={M} {4 * c}->{c}', ms, 2.0 * M * c * 4 * c)
        ms = timeit(lambda: ops.linear_vt(x, w, None, hw * hw))
        line(f'linear_vt M={M} {c}->{c} (V^T store)', ms, 2.0 * M * c * c)
    print('== fused attention ==')
    for (hw, c) in levels:
        n, d = hw * hw, c // 8
        q, k, v = r(BF, n, c), r(BF, n, c), r(BF, c, n)
        ms = timeit(lambda: ops.attention(q, k, v, 8, d ** -0.5))
        line(f'self-attn N={n} d={d}', ms, 4.0 * BF * 8 * n * n * d)
        kt, vt = r(2, 77, c), r(2, c, 80)
        ms = timeit(lambda: ops.attention(q, kt, vt, 8, d ** -0.5, kv_div=16))
        line(f'cross-attn N={n} d={d} keys=77', ms, 4.0 * BF * 8 * n * 77 * d)
    print('== temporal attention ==')
    for (hw, c) in levels:
        n = hw * hw
        q, k, v = r(BF * n, c), r(BF * n, c), r(BF * n, c)
        ms = timeit(lambda: ops.temporal_attention(q, k, v, 2, 16, 16, n, 8, (c // 8) ** -0.5))
        line(f'temporal N={n} C={c}', ms, bytes_=4.0 * BF * n * c * 2)
    print('== norms ==')
    for (hw, c) in levels:
        n = hw * hw
        x = r(BF, n, c)
        g, b = r(c), r(c)
        ms = timeit(lambda: ops.group_norm(x, g, b, 32, 1e-5, 2, silu=True))
        line(f'GroupNorm5D+SiLU N={n} C={c} (stats+apply)', ms, bytes_=3.0 * BF * n * c * 2)
        ms = timeit(lambda: ops.layer_norm(x, g, b))
        line(f'LayerNorm N={n} C={c}', ms, bytes_=2.0 * BF * n * c * 2)


if __name__ == '__main__':
    main()
