"""The CPU baseline leg of bench.py at the FULL T = 16 (one inversion step B = 1 + one CFG step B = 2 of the
fp32 oracle on the GPU box's host cores, ~5 min on 128 cores) beside the bounded T = 2 sample the default bench line carries:
checks the per-frame extrapolation once.  -> profiles/r04_cpu_baseline_T16.json

    python tools/cpu_baseline_T16.py [out.json]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    out = {}
    for frames in (2, 16):
        evals_per_s, threads, sample = bench.cpu_baseline(frames, (64, 64))
        out[f'T={frames}'] = {'unet_frame_evals_per_s': round(evals_per_s, 4), 'frames_per_s': round(evals_per_s / 150.0, 6),
                              'cores': threads, 'sample': sample}
        print(frames, out[f'T={frames}'], flush=True)
    out['extrapolation_error'] = round(out['T=2']['unet_frame_evals_per_s'] / out['T=16']['unet_frame_evals_per_s'] - 1.0, 4)
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                              'gpurun_out', 'cpu_baseline_T16.json')
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, 'w') as f:
        json.dump(out, f, indent=1)


if __name__ == '__main__':
    main()
