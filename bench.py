#!/usr/bin/env python
"""bench.py — VideoSwap denoising-path benchmark on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = the full hot path for ONE clip of the headline workload (BASELINE.json configs[1]): a 16-frame
512x512 clip (latents [1,4,16,64,64]) through the SD-1.5 UNet3D + AnimateDiff motion modules, 50-step DDIM inversion
(UNet batch 1) followed by 50-step classifier-free-guided DDIM sampling (UNet batch 2, guidance 7.5), synthetic
seeded weights and inputs (no checkpoints/datasets exist offline), fp16 storage / fp32 accumulation.  Inputs are
resident in HBM before the timed region.  With N > 1 every rank processes its own clips (clip-parallel, no
data-path collective: weak scaling) and the value is the whole-job aggregate.

Rank 0 prints ONE JSON line: metric value = denoised frames/s end to end (R1e = frames / wall(inversion + sampling)),
plus R1s / R2 readings, the roofline of the dominant kernel (vsx_gemm_f16: implicit-GEMM conv + GEMMs, 84 % of the
FLOPs) measured live with hipEvents on the launch stream, and the CPU baseline (the oracle — a port of the
reference's PyTorch path — timed on the host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0     # dense fp16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
FRAME_EVAL_TFLOP = 1.1046     # algorithmic TFLOP of one frame-evaluation at T=16, 64x64


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2, help='timed clips per rank')
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--frames', type=int, default=16)
    ap.add_argument('--latent', type=int, default=64)
    ap.add_argument('--ddim-steps', type=int, default=50)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--prof-samples', type=int, default=400000)
    # hipEvent pairs around every 7th vsx_gemm_f16 launch (7 is coprime with the ~470 GEMM launches of a UNet call, so
    # every shape is sampled over the 100 calls of a clip); bracketing EVERY launch costs 5 % of the loop
    ap.add_argument('--prof-stride', type=int, default=7)
    ap.add_argument('--graphs', action='store_true',
                    help='HIP-graph replay of the UNet forward (every --prof-stride-th UNet call stays eager and '
                         'carries the hipEvent brackets).  Off by default: the replay path has not been timed on '
                         'hardware yet (round 2 ran out of GPU minutes), so the headline number is the eager one')
    ap.add_argument('--no-graphs', action='store_true', help='(default) launch every kernel eagerly')
    return ap.parse_args()


def build_pipeline(device, frames):
    from videoswap_amd.compat import SD15_SCHEDULER_CONFIG, DDIMScheduler
    from videoswap_amd.pipeline import VideoSwapPipeline
    from videoswap_amd.synthetic import synth_weights_
    from videoswap_amd.unet import SD15_UNET_CONFIG, AnimateDiffUNet3DModel, inference_kwargs
    cfg = dict(SD15_UNET_CONFIG)
    cfg.update(inference_kwargs(max_len=max(24, frames)))
    with torch.device(device):
        unet = AnimateDiffUNet3DModel(**cfg)
    unet = synth_weights_(unet, seed=1234).half().eval()
    pipe = VideoSwapPipeline(unet=unet, scheduler=DDIMScheduler(**SD15_SCHEDULER_CONFIG))
    pipe.to(device)
    return pipe


def one_clip(pipe, data, ddim_steps):
    """The hot path for one clip: inversion (B=1) then guided sampling (B=2).  Every clip starts with cold
    step-invariant caches (time-embedding rows, text K/V): nothing computed for one clip is reused by the next."""
    pipe.unet.clear_step_caches()
    inv = pipe.invert(latents=data['latents'], prompt_embeds=data['text'], num_inference_steps=ddim_steps).latents
    embeds = torch.cat([data['negative'], data['text']])
    out = pipe(prompt=None, conditions=None, prompt_embeds=embeds[1:], negative_prompt_embeds=embeds[:1],
               latents=inv, num_inference_steps=ddim_steps, guidance_scale=7.5, output_type='latent').videos
    return out


def cpu_baseline(frames_sample=2, latent=64, repeats=2):
    """The oracle (CPU port of the reference's PyTorch path, fp32) on the host cores: one inversion-step UNet forward
    (B=1) on a bounded sample of `frames_sample` frames at the full SD-1.5 width and the same 64x64 latent; one
    warm-up forward, then the median of `repeats` timed forwards (with the sample bounded to ~30 s of
    CPU work as the bench contract asks: a full T=16 step pair is ~5 min on 128 cores)."""
    from oracle import unet3d
    torch.manual_seed(0)
    # PyTorch's default intra-op thread count (= physical cores).  Forcing os.cpu_count() (the SMT thread count) made
    # the same forward 3.5x SLOWER on the 128-core box (round-2 run 1: 41 s instead of 11.8 s for T = 2)
    threads = torch.get_num_threads()
    model = unet3d.AnimateDiffUNet3DModel(**unet3d.full_config()).eval()
    for n, p in model.named_parameters():           # proj_out is zero-initialised: make the temporal path live
        if 'temporal_transformer.proj_out' in n:
            torch.nn.init.normal_(p, std=0.02)
    x = torch.randn(1, 4, frames_sample, latent, latent)
    txt = torch.randn(1, 77, 768)
    times = []
    with torch.no_grad():
        model(x, torch.tensor(481), txt)            # warm-up (allocator, thread pool, oneDNN primitive caches)
        for _ in range(repeats):
            t0 = time.time()
            model(x, torch.tensor(481), txt)
            times.append(time.time() - t0)
    dt = sorted(times)[len(times) // 2]
    evals_per_s = frames_sample / dt
    return evals_per_s, threads, (f'1 UNet forward (inversion step), B=1, T={frames_sample}, {latent}x{latent} latent, '
                                  f'fp32, 1 warm-up + median of {repeats}: {dt:.1f} s')


def gemm_traffic(frames, latent):
    """HBM bytes per vsx_gemm_f16 launch from the PMC passes of tools/pmc_traffic.sh — only if that file was measured
    on THIS build of the library (source digest) at the benchmark shape; a stale file is not a measurement."""
    from videoswap_amd.build import source_digest
    path = os.path.join(ROOT, 'profiles', 'r02_gemm_hbm_traffic.json')
    if not os.path.exists(path) or frames != 16 or latent != 64:
        return None, 'no PMC traffic file for this shape'
    with open(path) as f:
        t = json.load(f)
    if t.get('lib_digest') != source_digest():
        return None, f'PMC traffic file is from another build ({str(t.get("lib_digest"))[:12]})'
    return round(t['hbm_bytes_per_launch']), f'{t["launches"]} launches of one inversion + one CFG step'


def main():
    args = parse()
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch multi-GPU runs with torch.distributed.run (one process per GPU)')
    distributed = world > 1
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if distributed:
        import torch.distributed as dist
        dist.init_process_group('nccl', device_id=device)

    from videoswap_amd import ops
    from videoswap_amd.synthetic import synthetic_clip
    pipe = build_pipeline(device, args.frames)
    # every rank owns different clips (seeded by rank); inputs resident in HBM before the timed region
    clips = [synthetic_clip(seed=1000 * rank + i, frames=args.frames, height=args.latent, width=args.latent,
                            device=device) for i in range(max(args.steps, 1))]

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    graphs = args.graphs and not args.no_graphs
    if graphs:
        # graph replay for the UNet forward; every prof_stride-th call runs eagerly and is the one whose GEMM launches
        # are bracketed by hipEvents (ALL of them: same 1/stride sampling fraction as the eager mode's every-7th-launch)
        pipe.unet.enable_hip_graphs(True, eager_every=args.prof_stride if args.prof_samples > 0 else 0)
        if args.warmup == 0:
            one_clip(pipe, clips[0], 1)              # capture the two graphs (B=1, B=2) outside the timed region
    for i in range(args.warmup):
        one_clip(pipe, clips[i % len(clips)], args.ddim_steps)
    barrier()
    if graphs:
        pipe.unet._graphs.on_eager = (lambda on: ops.prof_pause(not on))

    ops.FlopCounter.reset(True)
    ops.prof_enable(args.prof_samples > 0, args.prof_samples, stride=1 if graphs else args.prof_stride)
    if graphs:
        ops.prof_pause(True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_clip(pipe, clips[i % len(clips)], args.ddim_steps)
    barrier()
    elapsed = time.perf_counter() - t0
    ops.FlopCounter.enabled = False
    ops.prof_pause(False)
    n_launch, gemm_ms, gemm_flop = ops.prof_collect()
    ops.prof_enable(False, 0)

    from videoswap_amd.distributed import max_over_ranks
    elapsed = max_over_ranks(elapsed, device)        # whole-job time = slowest rank

    frames_total = world * args.steps * args.frames
    value = frames_total / elapsed
    total_flop = (ops.FlopCounter.gemm + ops.FlopCounter.attention) * world
    evals = world * args.steps * args.frames * 3 * args.ddim_steps          # (1 + 2) UNet frame-evals per DDIM step
    out = {
        'metric': 'denoised frames/sec, 16-frame 512^2 clip @ 50 DDIM steps (end to end: inversion + CFG sampling)',
        'value': round(value, 4), 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(1000.0 * elapsed / max(args.steps, 1), 2), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f16', 'data': 'synthetic',
        'config': {'launch': 'hip-graph' if graphs else 'eager', 'workload': f'{args.frames}-frame {args.latent * 8}x{args.latent * 8} clip, SD-1.5 UNet3D + '
                               f'AnimateDiff motion modules, {args.ddim_steps}-step DDIM inversion (B=1) + '
                               f'{args.ddim_steps}-step CFG-7.5 DDIM sampling (B=2), one clip per GPU per step',
                   'latents': [1, 4, args.frames, args.latent, args.latent], 'parallelism': f'clip-parallel x{world}'},
        'readings': {'R1e_frames_per_s': round(value, 4),
                     'R2_unet_frame_evals_per_s': round(evals / elapsed, 2),
                     'loop_algorithmic_tflop': round(total_flop / 1e12, 1),
                     'loop_tflops': round(total_flop / elapsed / 1e12, 1),
                     'loop_mfma_frac': round(total_flop / elapsed / 1e12 / (MFMA_PEAK_TFLOPS * world), 4),
                     'ceiling_R1e_at_100pct_mfma': 15.1},
    }
    if n_launch > 0 and gemm_ms > 0:
        ach = gemm_flop / (gemm_ms * 1e-3) / 1e12
        traffic, traffic_note = gemm_traffic(args.frames, args.latent)
        out['roofline'] = {'bound': 'mfma', 'kernel': 'vsx_gemm_f16 (implicit-GEMM conv + GEMM, all shapes)',
                           'achieved': round(ach, 1), 'peak': MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                           'frac': round(ach / MFMA_PEAK_TFLOPS, 4), 'traffic': traffic, 'traffic_source': traffic_note,
                           'launches_sampled': int(n_launch), 'sample_stride': args.prof_stride,
                           'sampling': ('all GEMM launches of every %d-th UNet call (eager); the other calls are '
                                        'HIP-graph replays' % args.prof_stride) if graphs else
                                       ('every %d-th GEMM launch' % args.prof_stride),
                           'avg_launch_us': round(1000.0 * gemm_ms / n_launch, 2),
                           'avg_launch_gflop': round(gemm_flop / n_launch / 1e9, 2),
                           'kernel_time_share_of_wall': round(gemm_ms * 1e-3 * args.prof_stride / (elapsed * 1.0), 4)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            evals_per_s, threads, sample = cpu_baseline()
            out['cpu_baseline'] = {'value': round(evals_per_s / (3 * args.ddim_steps), 6), 'unit': 'frames/s',
                                   'cores': threads, 'kind': 'port', 'sample': sample,
                                   'unet_frame_evals_per_s': round(evals_per_s, 4)}
        except Exception as e:  # the baseline must never take the GPU number down with it
            out['cpu_baseline'] = {'value': None, 'unit': 'frames/s', 'cores': torch.get_num_threads(),
                                   'kind': 'port', 'sample': f'failed: {e!r}'}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
