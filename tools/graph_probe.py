"""Diagnostic: capture and replay the UNet forward as a HIP graph, with timings and a stack dump if anything stalls.

    timeout 150 python tools/graph_probe.py [--full]     # tiny model by default, --full = SD-1.5 width B=1 T=16
"""
import faulthandler
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.dump_traceback_later(60, repeat=True)


def main():
    from videoswap_amd.synthetic import synth_weights_
    from videoswap_amd.unet import SD15_UNET_CONFIG, AnimateDiffUNet3DModel, inference_kwargs
    full = '--full' in sys.argv
    cfg = dict(SD15_UNET_CONFIG)
    cfg.update(inference_kwargs())
    if not full:
        cfg.update(block_out_channels=(64, 128, 256, 256), cross_attention_dim=64)
    with torch.device('cuda'):
        unet = AnimateDiffUNet3DModel(**cfg)
    unet = synth_weights_(unet).half().eval()
    T, hw, dim = (16, 64, 768) if full else (4, 16, 64)
    x = torch.randn(1, 4, T, hw, hw, device='cuda', dtype=torch.float16)
    txt = torch.randn(1, 77, dim, device='cuda', dtype=torch.float16)

    def timed(label, fn, n=1):
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(n):
            out = fn()
        torch.cuda.synchronize()
        print(f'{label}: {(time.time() - t0) / n * 1e3:.2f} ms', flush=True)
        return out
    with torch.no_grad():
        ref = timed('eager first', lambda: unet(x, 481, txt).sample)
        timed('eager steady', lambda: unet(x, 481, txt).sample, 5)
        unet.enable_hip_graphs(True)
        out = timed('graph capture + first replay', lambda: unet(x, 481, txt).sample)
        print('identical to eager:', bool(torch.equal(out, ref)), flush=True)
        timed('graph replay', lambda: unet(x, 481, txt).sample, 10)
        timed('graph replay, other timestep', lambda: unet(x, 21, txt).sample, 5)
        print('captures', unet._graphs.captures, 'replays', unet._graphs.replays, flush=True)


if __name__ == '__main__':
    main()
