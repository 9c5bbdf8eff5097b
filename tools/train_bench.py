"""One adapter training step at the reference's training configuration (options/train_videoswap/**: 16 frames, 512x512,
fp16, batch 1) on synthetic weights and inputs: forward through the frozen SD-1.5-width UNet with the adapter residuals,
masked MSE, backward on the kernel gradient path, AdamW on the adapter.  Prints ms per step, the split forward / backward
and the peak device memory.  Needs the backward kernels: VSX_LIB_VARIANT=next until they are part of libvsx.so.

    VSX_LIB_VARIANT=next python tools/train_bench.py [--frames 16] [--latent 64] [--steps 3]"""
import argparse
import os
import random
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=16)
    ap.add_argument('--latent', type=int, default=64)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--points', type=int, default=8)
    args = ap.parse_args()
    from videoswap_amd.adapter import SparsePointAdapter
    from videoswap_amd.compat import SD15_SCHEDULER_CONFIG, DDPMScheduler
    from videoswap_amd.synthetic import synth_weights_
    from videoswap_amd.trainer import VideoSwapTrainer
    from videoswap_amd.unet import SD15_UNET_CONFIG, AnimateDiffUNet3DModel, inference_kwargs
    dev = torch.device('cuda', 0)
    cfg = dict(SD15_UNET_CONFIG)
    cfg.update(inference_kwargs(max_len=max(24, args.frames)))
    with torch.device(dev):
        unet = AnimateDiffUNet3DModel(**cfg)
    unet = synth_weights_(unet, seed=1234).half().eval()
    adapter = synth_weights_(SparsePointAdapter(), seed=3).to(dev).train()          # fp32 master weights
    sched = DDPMScheduler(**{k: SD15_SCHEDULER_CONFIG[k] for k in ('num_train_timesteps', 'beta_start', 'beta_end',
                                                                   'beta_schedule')})
    opt = torch.optim.AdamW(adapter.parameters(), lr=5e-4, weight_decay=0.01)
    trainer = VideoSwapTrainer(unet=unet, scheduler=sched, adapter=adapter, optimizer=opt,
                               tune_cfg={'min_timestep': 0.5, 'drop_rate': 0.2, 'loss_type': 'global'})
    g = torch.Generator().manual_seed(0)
    side = args.latent * 8
    tracks = torch.rand(1, args.frames, args.points, 2, generator=g) * side
    batch = {'pred_tracks': tracks, 'img_size': (side, side),
             'point_embedding': torch.randn(1, args.points, 1280, generator=g).to(dev)}
    latents = torch.randn(1, 4, args.frames, args.latent, args.latent, generator=g).to(dev).half()
    text = torch.randn(1, 77, 768, generator=g).to(dev).half()
    random.seed(0)
    torch.manual_seed(0)
    for step in range(args.steps + 1):                 # step 0 = warm-up (allocator, weight copies of the gradient path)
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        t0 = time.perf_counter()
        noise = torch.randn_like(latents)
        t = trainer.sample_timesteps(1, dev)
        loss, _ = trainer.loss_from(latents, noise, t, text, batch)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        applied = trainer.backward_and_update(loss)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f'step {step}{" (warm-up)" if step == 0 else ""}: loss {float(loss.detach()):.4f}  forward {1e3 * (t1 - t0):.1f} ms  '
              f'backward + update {1e3 * (t2 - t1):.1f} ms  total {1e3 * (t2 - t0):.1f} ms  update applied {applied}  '
              f'loss scale {trainer.loss_scale:g}  peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB', flush=True)


if __name__ == '__main__':
    main()
