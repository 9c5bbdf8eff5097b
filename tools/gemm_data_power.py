"""Is the persistent GEMM's rate on random data set by its schedule or by the power the DATA makes the matrix pipes draw?

VERDICT r5 (weak 5, next 4): on zero-filled operands gemm_pp_kernel<2, 0, plain> equals the guide's 256^2 8-phase template
(1 562 / 1 709 TF/s at 4096 x 5120 x 4096 / 8192 x 10240 x 8192 against 1 563 / 1 728), on uniform random fp16 it drops 34 % where
the guide's template — measured by its authors in BF16 on another box — drops 15 %.  The guide's example file
(examples/gemm_256sq_8phase_bf16.cpp) is not in this image (only its prose description is, cdna_hip_programming.md §5), so the
comparison is made on the variable that differs instead of on a retyped kernel: the SAME kernel, the SAME box, ONE process, the
operand families alternating round by round —

    zeros                    no toggling at all (the guide's 1 563-TF figure)
    uniform fp16             uniform [-1, 1), all 10 mantissa bits random (round 5's 1 032 TF/s)
    uniform, 7-bit mantissa  the same values rounded to what BF16 can hold (8-bit significands, as in the guide's random-data runs):
                             if the gap to fp16 is the multiplier array's switching energy, THIS is the guide's 1 320 - 1 340
    uniform, 3-bit mantissa  / sign only (+-1) / powers of two: fewer significand bits still
    normal(0, 1), normal(0, 0.02): what activations and weights look like

If the rate follows the number of live significand bits on one box, the loop is power-bound and its schedule is not what the
random-data rate measures; if the 7-bit family stays at the fp16 rate, the guide's template has something this loop lacks.

    python tools/gemm_data_power.py > gpurun_out/gemm_data_power.txt
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoswap_amd import ops  # noqa: E402

DEV, H16 = 'cuda', torch.float16


def trunc_mantissa(t, bits):
    """round-to-nearest onto fp16 values with `bits` explicit mantissa bits (10 = unchanged)"""
    if bits >= 10:
        return t
    m, e = torch.frexp(t.float())                      # t = m * 2^e, |m| in [0.5, 1): `bits` explicit bits = bits + 1 significant
    return torch.ldexp(torch.round(m * 2.0 ** (bits + 1)) / 2.0 ** (bits + 1), e).to(H16)


def family(name, *s, seed=0):
    g = torch.Generator(device=DEV).manual_seed(1234 + seed)
    u = torch.rand(*s, device=DEV, dtype=torch.float32, generator=g) * 2.0 - 1.0
    if name == 'zeros':
        return torch.zeros(*s, device=DEV, dtype=H16)
    if name == 'ones':
        return torch.ones(*s, device=DEV, dtype=H16)
    if name == 'uniform fp16 (10-bit mantissa)':
        return u.to(H16)
    if name.startswith('uniform, '):
        return trunc_mantissa(u.to(H16), int(name.split()[1].split('-')[0]))
    if name == 'sign only (+-1)':
        return torch.where(u < 0, -1.0, 1.0).to(H16)
    if name == 'powers of two 2^-8..2^0, random sign':
        e = torch.randint(-8, 1, s, device=DEV, generator=g).float()
        return (torch.where(u < 0, -1.0, 1.0) * torch.exp2(e)).to(H16)
    if name == 'normal(0, 1)':
        return torch.randn(*s, device=DEV, dtype=torch.float32, generator=g).to(H16)
    if name == 'normal(0, 0.02)':
        return (torch.randn(*s, device=DEV, dtype=torch.float32, generator=g) * 0.02).to(H16)
    raise ValueError(name)


FAMILIES = ['zeros', 'uniform fp16 (10-bit mantissa)', 'uniform, 7-bit mantissa (BF16-representable)', 'uniform, 3-bit mantissa',
            'sign only (+-1)', 'powers of two 2^-8..2^0, random sign', 'ones', 'normal(0, 1)', 'normal(0, 0.02)']


def main():
    rounds, reps = 7, 4
    ops.set_option('gemm_pp', 2)
    for (M, N, K) in ((4096, 5120, 4096), (8192, 10240, 8192)):
        out = torch.empty(M, N, device=DEV, dtype=H16)
        data = {}
        for f in FAMILIES:
            data[f] = (family(f, M, K), family(f, N, K, seed=1))
        ts = {f: [] for f in FAMILIES}
        for f in FAMILIES:      # warm
            ops.linear(data[f][0], data[f][1], None, out=out)
        torch.cuda.synchronize()
        for rnd in range(rounds):
            order = FAMILIES[rnd % len(FAMILIES):] + FAMILIES[:rnd % len(FAMILIES)]     # rotate: a family's time depends on what ran before it
            for f in order:
                x, w = data[f]
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    ops.linear(x, w, None, out=out)
                e1.record()
                e1.synchronize()
                ts[f].append(e0.elapsed_time(e1) / reps)
        print(f'# gemm_pp_kernel<2, 0, plain>, M={M} N={N} K={K}; median of {rounds} interleaved rounds x {reps} launches')
        print(f'{"operand family":48s} {"us":>9s} {"TF/s":>8s} {"of zeros":>9s}')
        base = None
        for f in FAMILIES:
            ms = sorted(ts[f])[len(ts[f]) // 2]
            tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
            base = base or tf
            print(f'{f:48s} {ms * 1e3:9.1f} {tf:8.0f} {tf / base:9.3f}', flush=True)
        del data, out
    ops.set_option('gemm_pp', 1)


if __name__ == '__main__':
    main()
