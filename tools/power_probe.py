"""Same-box evidence for "the matrix kernels run at the chip's power-managed clock": board power and shader clock, sampled from the
SMI while ONE kernel streams for a few seconds on each operand family (the power-bound conclusion rested on a
cross-box comparison).

For every (kernel, operand family) the launch loop runs `--seconds` of back-to-back launches; a sampler thread (videoswap_amd/telemetry.py) reads the average
socket power and the shader clock of THIS device (matched by PCI address) from sysfs about every 50 ms.  Reported: TF/s of the loop, mean / max power, mean / min shader clock over the steady part
(the first quarter of the samples is dropped).  A schedule-bound kernel would hold its clock and vary its power with the data; a
power-bound one holds the power (at the cap) and gives up clock.

    python tools/power_probe.py [--seconds 4] > gpurun_out/power_probe.txt
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoswap_amd import ops  # noqa: E402
from tools.gemm_data_power import family  # noqa: E402
from videoswap_amd.telemetry import BoardPower  # noqa: E402

DEV, H16 = 'cuda', torch.float16


def run_for(fn, seconds):
    fn()
    torch.cuda.synchronize()
    n = 0
    with BoardPower(torch.cuda.current_device()) as bp:
        t0 = time.time()
        while time.time() - t0 < seconds:
            for _ in range(8):
                fn()
            n += 8
            torch.cuda.synchronize()
        dt = time.time() - t0
    return n / dt, bp.summary(skip_frac=0.25)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seconds', type=float, default=4.0)
    args = ap.parse_args()
    probe = BoardPower(torch.cuda.current_device())
    print(f'# device {probe.pci}: hwmon {probe.hw}, power cap {probe.cap_W()} W, idle reading (W, MHz) {probe.one()}')
    if not probe.available:
        print('# no readable power file for this device: nothing to measure')
        return
    print(f'{"kernel / operands":84s} {"TF/s":>7s} {"W mean":>8s} {"W max":>7s} {"at cap":>7s} {"MHz mean":>9s} {"MHz min":>8s} {"J/launch":>9s}')

    def report(name, rate, flop, sm):
        if sm is None:
            print(f'{name:84s} {rate * flop / 1e12:7.0f}   (no samples)', flush=True)
            return
        print(f'{name:84s} {rate * flop / 1e12:7.0f} {sm["mean_W"]:8.0f} {sm["max_W"]:7.0f} {sm.get("share_at_cap", 0):7.2f} '
              f'{sm.get("sclk_mean_MHz", 0):9.0f} {sm.get("sclk_min_MHz", 0):8.0f} {sm["mean_W"] / rate:9.3f}', flush=True)

    # the persistent GEMM on a square problem: one kernel, the operand families of tools/gemm_data_power.py
    ops.set_option('gemm_pp', 2)
    M, N, K = 8192, 10240, 8192
    out = torch.empty(M, N, device=DEV, dtype=H16)
    for fam in ('zeros', 'uniform fp16 (10-bit mantissa)', 'uniform, 7-bit mantissa (BF16-representable)', 'sign only (+-1)', 'normal(0, 1)',
                'zeros'):
        x, w = family(fam, M, K), family(fam, N, K, seed=1)
        rate, smp = run_for(lambda: ops.linear(x, w, None, out=out), args.seconds)
        report(f'gemm_pp_kernel<2,0,plain> {M}x{N}x{K}: {fam}', rate, 2.0 * M * N * K, smp)
        del x, w
    ops.set_option('gemm_pp', 1)
    del out
    # the UNet's own launches: a K = 320 GEGLU and a 3 x 3 convolution at the 64 x 64 level (B = 2), normal data
    g = torch.Generator(device=DEV).manual_seed(5)
    Mg = 131072
    for fam in ('zeros', 'normal(0, 1)'):
        x = family(fam, Mg, 320)
        w, b = family('normal(0, 0.02)' if fam != 'zeros' else 'zeros', 2560, 320, seed=2), torch.zeros(2560, device=DEV, dtype=H16)
        rate, smp = run_for(lambda: ops.linear(x, w, b, geglu=True), args.seconds)
        report(f'geglu M={Mg} 320->1280: {fam}', rate, 2.0 * Mg * 2560 * 320, smp)
        del x, w
    # a 3 x 3 convolution of the 64 x 64 level (B = 2: 32 images), a byte-bound K = 320 projection with a residual, a GroupNorm
    for fam in ('zeros', 'normal(0, 1)'):
        wfam = 'normal(0, 0.02)' if fam != 'zeros' else 'zeros'
        xc = family(fam, 32, 64, 64, 320)
        wc, bc = family(wfam, 320, 3, 3, 320, seed=6), torch.zeros(320, device=DEV, dtype=H16)
        rate, smp = run_for(lambda: ops.conv2d(xc, wc, bc), args.seconds)
        report(f'conv3x3 64x64 320->320, 32 images: {fam}', rate, 2.0 * Mg * 320 * 2880, smp)
        x = family(fam, Mg, 320)
        w, b, res = family(wfam, 320, 320, seed=7), torch.zeros(320, device=DEV, dtype=H16), family(fam, Mg, 320, seed=8)
        rate, smp = run_for(lambda: ops.linear(x, w, b, residual=res), args.seconds)
        report(f'gemm M={Mg} 320->320 +res (byte-bound): {fam}', rate, 2.0 * Mg * 320 * 320, smp)
        gam, bet = torch.ones(320, device=DEV, dtype=H16), torch.zeros(320, device=DEV, dtype=H16)
        rate, smp = run_for(lambda: ops.group_norm(x, gam, bet, 32, 1e-5, 2, silu=True), args.seconds)
        report(f'group_norm + SiLU [2, 65536, 320] (TF/s column = GB/s moved / 1000 x 3 passes): {fam}', rate, 3.0 * Mg * 320 * 2 * 1e3, smp)
        del xc, wc, x, w, res
    # the workgroup-per-tile kernels (two free-running workgroups per CU): the 16 x 16 / 32 x 32 convolutions of one clip per step
    for nimg, hw, c in ((32, 16, 1280), (16, 16, 1280), (16, 32, 640), (16, 8, 1280)):
        xc = family('normal(0, 1)', nimg, hw, hw, c)
        wc, bc = family('normal(0, 0.02)', c, 3, 3, c, seed=9), torch.zeros(c, device=DEV, dtype=H16)
        rate, smp = run_for(lambda: ops.conv2d(xc, wc, bc), args.seconds)
        report(f'conv3x3 {hw}x{hw} {c}->{c}, {nimg} images (tile kernels unless persistent): normal(0, 1)', rate, 2.0 * nimg * hw * hw * c * c * 9, smp)
        del xc, wc
    # flash attention at the 64 x 64 level (32 images, 8 heads, d = 40)
    nb, heads, n, d = 32, 8, 4096, 40
    for fam in ('zeros', 'normal(0, 1)'):
        q, k = family(fam, nb, n, heads * d), family(fam, nb, n, heads * d, seed=3)
        vt = family(fam, nb, heads * d, n, seed=4)
        fn = lambda: ops.attention(q, k, vt, heads, d ** -0.5)      # noqa: E731
        rate, smp = run_for(fn, args.seconds)
        report(f'flash_attn<40> nb={nb} n={n}: {fam}', rate, 4.0 * nb * heads * n * n * d, smp)
    del g


if __name__ == '__main__':
    main()
