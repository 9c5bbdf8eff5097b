"""Which PyTorch (non-libvsx) kernels does the denoising loop launch, how often, and from which line of the host mirror?

The hot path's arithmetic lives in libvsx.so; PyTorch is plumbing (device memory, views, the odd cat / copy).  rocprof shows ~7 000
small at::native launches per 10 + 10 steps at one clip per step (profiles/r05_kernel_stats_one_clip.txt: elementwise, dtype copies,
fills, cat: ~3.5 % of kernel time, 3 - 5 us each, every one of them a launch-latency-bound gap between two libvsx kernels).  This probe
runs a few steps of both loops under torch.profiler with stacks and prints, per aten op that launched a device kernel: calls per UNet
forward, device microseconds, and the innermost frame inside videoswap_amd/ that issued it.

    python tools/torch_glue_probe.py [--steps 2] [--clips 1] > gpurun_out/torch_glue_probe.txt
"""
import argparse
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--clips', type=int, default=1)
    args = ap.parse_args()
    from torch.profiler import ProfilerActivity, profile
    from videoswap_amd.synthetic import synthetic_clip
    dev = torch.device('cuda', 0)
    pipe = bench.build_pipeline(dev, 16)
    clips = [synthetic_clip(seed=i, frames=16, height=64, width=64, device=dev) for i in range(args.clips)]
    data = bench.stack_clips(clips)
    bench.one_clip(pipe, data, args.steps)            # warm: packing, caches of the first call
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        bench.one_clip(pipe, data, args.steps)
        torch.cuda.synchronize()
    forwards = 2 * args.steps
    rows = collections.defaultdict(lambda: [0, 0.0, collections.Counter(), collections.Counter()])
    for ev in prof.events():
        dev_us = getattr(ev, 'device_time', None)
        if dev_us is None:
            dev_us = getattr(ev, 'cuda_time', 0.0)
        if not dev_us or not ev.name.startswith('aten::'):
            continue
        if any(k.name.startswith('aten::') for k in (ev.cpu_children or []) if (getattr(k, 'device_time', None) or getattr(k, 'cuda_time', 0.0))):
            continue                                   # count the innermost aten op that owns the kernel
        frame = 'outside videoswap_amd'
        for fr in (ev.stack or []):
            if 'videoswap_amd' in fr and 'site-packages' not in fr:
                frame = fr.split('videoswap_amd/')[-1]
                break
        r = rows[ev.name]
        r[0] += 1
        r[1] += dev_us
        r[2][frame] += 1
        r[3][str([tuple(s) for s in (ev.input_shapes or [])][:2])] += 1
    tot_n = sum(r[0] for r in rows.values())
    tot_us = sum(r[1] for r in rows.values())
    print(f'# {args.steps} inversion + {args.steps} CFG steps at {args.clips} clip(s) per step = {forwards} UNet forwards; aten ops that launched device '
          f'kernels: {tot_n} calls ({tot_n / forwards:.1f} per forward), {tot_us:.0f} us device time ({tot_us / forwards:.0f} us per forward)')
    for name, (n, us, frames, shapes) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        print(f'{name:40s} {n / forwards:7.1f} per forward {us / forwards:8.1f} us per forward')
        for fr, c in frames.most_common(6):
            print(f'        {c / forwards:6.1f} x  {fr}')
        for sh, c in shapes.most_common(3):
            print(f'        shapes {c / forwards:6.1f} x  {sh}')


if __name__ == '__main__':
    main()
