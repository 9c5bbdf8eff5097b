#!/usr/bin/env python
"""Where does a configs[2] clip spend its time?  Wall time (with a device sync) of every phase of
VideoSwapPipeline.validation at the benchmark size: inversion with the AttentionStore, weight snapshot, ED-LoRA merge,
controller construction, guided sampling with the edit controller, weight restore.

    python tools/cfg3_phases.py [--ddim-steps 50]"""
import argparse
import copy
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ddim-steps', type=int, default=50)
    args = ap.parse_args()
    import bench
    from videoswap_amd import edlora, pipeline as P
    from videoswap_amd.synthetic import synthetic_clip
    dev = torch.device('cuda', 0)
    pipe = bench.build_pipeline(dev, 16, swap=True)
    lora = bench.synthetic_edlora(pipe.unet.state_dict())
    clip = synthetic_clip(seed=0, frames=16, height=64, width=64, device=dev)
    phases = {}

    def timed(name, fn):
        def wrap(*a, **k):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = fn(*a, **k)
            torch.cuda.synchronize()
            phases[name] = phases.get(name, 0.0) + time.perf_counter() - t0
            return r
        return wrap
    pipe.prepare_ddim_inverted_latents = timed('inversion (50 steps, AttentionStore)', pipe.prepare_ddim_inverted_latents)
    P.copy.deepcopy = timed('deepcopy (weight snapshot, conditions)', copy.deepcopy)
    P.convert_edlora = timed('convert_edlora (LoRA merge)', edlora.convert_edlora)
    pipe.get_edit_controller = timed('get_edit_controller', pipe.get_edit_controller)
    P.register_attention_control = timed('register_attention_control', P.register_attention_control)
    orig_call = type(pipe).__call__
    pipe_call = timed('guided sampling (50 steps, edit controller, adapter)', lambda *a, **k: orig_call(pipe, *a, **k))
    type(pipe).__call__ = lambda self, *a, **k: pipe_call(*a, **k)
    pipe.unet.load_state_dict = timed('load_state_dict (restore)', pipe.unet.load_state_dict)
    for rep in range(2):
        phases.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bench.swap_clip(pipe, clip, args.ddim_steps, lora)
        torch.cuda.synchronize()
        total = time.perf_counter() - t0
    for k, v in phases.items():
        print(f'{v * 1e3:9.1f} ms  {100 * v / total:5.1f} %  {k}')
    print(f'{total * 1e3:9.1f} ms  total ({sum(phases.values()) * 1e3:.1f} ms in the phases above)')


if __name__ == '__main__':
    main()
