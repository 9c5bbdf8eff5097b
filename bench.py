#!/usr/bin/env python
"""bench.py — VideoSwap denoising-path benchmark on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = the full hot path for ONE clip of the headline workload — BASELINE.json configs[1] = SURVEY.md §8(d) Config 2, what
the reference's test.py drives (test.py:102-109, pipeline_videoswap.py:376-394: one clip per option file, batch_size 1): a
16-frame 512x512 clip (latents [1,4,16,64,64]) through the SD-1.5 UNet3D + AnimateDiff motion modules, 50-step DDIM inversion
(UNet batch 1) followed by 50-step classifier-free-guided DDIM sampling (UNet batch 2, guidance 7.5), synthetic seeded weights
and inputs (no checkpoints / datasets exist offline), fp16 storage / fp32 accumulation.  Inputs are resident in HBM before the
timed region.  `value`, `ms_per_step`, `roofline` and `roofline.traffic` all belong to THAT workload over the full --steps.
With N > 1 every rank denoises its own clips, one per step (clip-parallel = configs[4] / Config 5, no data-path collective: weak
scaling) and the value is the whole-job aggregate.

After the timed region the single-GPU run measures a second, separately labelled leg, `throughput_mode`: --throughput-clips
(4) INDEPENDENT configs[1] clips denoised together (latents [4,4,16,64,64], the batch axis of the reference's own pipeline,
pipeline_videoswap.py:478-550; cf. configs[4] "throughput mode"), one warm-up + --throughput-steps (3) timed steps with its own
roofline.  It is never `value`.  --clips-per-step B makes the batch the headline of a run (and says so in `metric`).

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
# what a register-resident stream of v_mfma_f32_32x32x16_f16 delivers on normal(0, 1) fp16 operands with all 256 CUs busy: the board's
# power / current management holds the shader clock at 1.55 - 1.76 GHz (profiles/r06_ubench_mfma_power.txt: 1 580 / 1 679 TF/s on two
# boxes; 2 459 - 2 467 on zeros at 2.40 GHz).  Reported beside the nominal peak, never instead of it.
MFMA_REAL_DATA_TFLOPS = 1630.0
NOMINAL_SCLK_MHZ = 2400.0
FRAME_EVAL_TFLOP = 1.1046     # algorithmic TFLOP of one frame-evaluation at T=16, 64x64
HBM_COPY_TBPS = 6.29          # measured float4-copy rate, /opt/skills/guides/MI355X_MICROARCH.md (8.0 TB/s spec): the byte roofline


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2, help='timed clips per rank')
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--frames', type=int, default=0, help='frames per clip (default 16; 64 with --config 4)')
    ap.add_argument('--exchange', default='auto', choices=('auto', 'kv', 'sites'), help='--config 4: FrameShard exchange')
    ap.add_argument('--cpu-frames', type=int, default=2,
                    help='frames of the CPU-baseline sample (a T = 16 step pair is ~5 min on 128 cores: '
                         'profiles/r04_cpu_baseline_T16.json records one)')
    ap.add_argument('--latent', type=int, default=64)
    ap.add_argument('--ddim-steps', type=int, default=50)
    ap.add_argument('--config', type=int, default=2, choices=(2, 3, 4),
                    help='4 = BASELINE.json configs[3]: ONE 64-frame clip (--frames defaults to 64), its frame axis sharded '
                         'over the --gpus ranks (videoswap_amd.distributed.FrameShard: all-to-all re-shard frames <-> sites '
                         'around every motion module / all-gather of the temporal K|V over RCCL; strong scaling; on one '
                         'GPU the whole clip runs unsharded).  2 = '
                         'BASELINE.json configs[1] (default; the headline: plain text embedding, no adapter, no '
                         'controller) or configs[2]: the full swap path through VideoSwapPipeline.validation — AttentionStore '
                         'during the inversion, ED-LoRA merge + per-layer embeddings [2,16,77,768], adapter residuals for '
                         'the first half of the sampling steps, AttentionRefine + two SpatialBlenders (use_blend: true)')
    ap.add_argument('--latent-h', type=int, default=0, help='latent height (default: --latent); 56 with --latent-w 96 = the '
                    '448x768 frames of 26 of the 30 reference option files')
    ap.add_argument('--latent-w', type=int, default=0)
    ap.add_argument('--no-cpu-baseline', action='store_true', help='skip both baseline legs (CPU port, torch-ROCm eager)')
    ap.add_argument('--prof-samples', type=int, default=400000)
    # hipEvent pairs around every 7th vsx_gemm_f16 launch (7 is coprime with the ~470 GEMM launches of a UNet call, so
    # every shape is sampled over the 100 calls of a clip); bracketing EVERY launch costs 5 % of the loop
    ap.add_argument('--prof-stride', type=int, default=7)
    ap.add_argument('--clips-per-step', type=int, default=1,
                    help='independent clips denoised TOGETHER in one step of the HEADLINE: latents [B,4,T,h,w], the batch axis of the '
                         'reference\'s own pipeline (pipeline_videoswap.py:478-550) — UNet batch B in the inversion, 2B under CFG.  '
                         'Default 1 = SURVEY.md §8(d) Config 2, the shape the reference\'s test.py drives; B > 1 is a throughput mode '
                         'and is named in `metric`.  The default run measures B = 4 as its own leg (`throughput_mode`)')
    ap.add_argument('--throughput-clips', type=int, default=4,
                    help='clips per step of the second leg (`throughput_mode`, single-GPU default-config runs only; 0 = skip)')
    ap.add_argument('--throughput-steps', type=int, default=3, help='timed steps of the second leg (after one warm-up step)')
    ap.add_argument('--no-extra-reading', action='store_true', help='skip the `throughput_mode` leg (profiling runs pass it)')
    args = ap.parse_args()
    if args.frames == 0:
        args.frames = 64 if args.config == 4 else 16
    return args


def build_pipeline(device, frames, swap=False):
    from videoswap_amd.compat import SD15_SCHEDULER_CONFIG, DDIMScheduler
    from videoswap_amd.pipeline import VideoSwapPipeline
    from videoswap_amd.synthetic import synth_weights_
    from videoswap_amd.unet import SD15_UNET_CONFIG, AnimateDiffUNet3DModel, inference_kwargs
    cfg = dict(SD15_UNET_CONFIG)
    cfg.update(inference_kwargs(max_len=max(24, frames)))
    with torch.device(device):
        unet = AnimateDiffUNet3DModel(**cfg)
    unet = synth_weights_(unet, seed=1234).half().eval()
    extra = {}
    if swap:        # configs[2]: point adapter, (synthetic) tokenizer / text encoder for the P2P word bookkeeping
        from videoswap_amd.adapter import SparsePointAdapter
        from videoswap_amd.synthetic import SyntheticTextEncoder, WhitespaceTokenizer
        with torch.device(device):
            adapter = SparsePointAdapter(embedding_channels=1280, channels=list(cfg['block_out_channels']))
        extra = dict(adapter=synth_weights_(adapter, seed=3).half().eval(), tokenizer=WhitespaceTokenizer(),
                     text_encoder=SyntheticTextEncoder(dim=768, dtype=torch.float16, device=device))
    pipe = VideoSwapPipeline(unet=unet, scheduler=DDIMScheduler(**SD15_SCHEDULER_CONFIG), **extra)
    pipe.to(device)
    return pipe


SWAP_SOURCE = 'a silver jeep driving down a curvy road in the countryside'


def synthetic_edlora(state_dict, rank=4, seed=4):
    """rank-4 LoRA factors N(0, 0.01^2) on the keys convert_edlora_to_diffusers.py:46-53 merges, two new
    concept tokens with 16 per-layer embeddings each."""
    g = torch.Generator().manual_seed(seed)
    lora = {}
    for k, w in state_dict.items():
        if (k.endswith(('to_q.weight', 'to_k.weight', 'to_v.weight', 'to_out.0.weight', 'ff.net.0.proj.weight',
                        'ff.net.2.weight', 'proj_in.weight', 'proj_out.weight'))
                and 'motion_modules' not in k and 'attentions' in k):
            down = torch.randn(rank, w.shape[1], generator=g) * 0.01
            up = torch.randn(w.shape[0], rank, generator=g) * 0.01
            if w.dim() == 4:
                down, up = down[:, :, None, None], up[:, :, None, None]
            lora[k[:-6] + 'lora_down.weight'], lora[k[:-6] + 'lora_up.weight'] = down, up
    emb = {'<porsche1>': torch.zeros(16, 768), '<porsche2>': torch.zeros(16, 768)}
    return {'params': {'new_concept_embedding': emb, 'unet': lora}}


def swap_clip(pipe, data, ddim_steps, lora, marks=None):
    """configs[2], one clip through the reference's own orchestration (pipeline_videoswap.py:272-423 = VideoSwapPipeline.
    validation): inversion with the AttentionStore -> ED-LoRA merge -> edit controller -> guided sampling with adapter
    residuals for steps 0..25 and the blend callbacks -> weights restored.  Everything is inside the timed region."""
    pipe.unet.clear_step_caches()
    cfg = dict(use_invertion_latents=True, use_blend=True, num_inference_steps=ddim_steps, guidance_scale=7.5,
               t2i_guidance_scale=0.5, t2i_start=0.0, t2i_end=0.5,
               editing_prompts={'0': dict(replace='silver jeep -> <porsche1> <porsche2>',
                                          lora_path='synthetic_edlora.pth---1.0',
                                          blend_cfg=dict(cross_replace_steps=0.3, self_replace_steps=0.3, blend_th=0.3))})
    video = data['latents'][0].permute(1, 0, 2, 3).contiguous()          # [F,4,h,w]: 4 channels = already latents
    if marks is not None:
        invert = pipe.invert

        def marked(*a, **k):
            r = invert(*a, **k)
            marks.append(_event())
            return r
        pipe.invert = marked
    try:
        return pipe.validation(video, data['conditions'], SWAP_SOURCE, cfg, lora_loader=lambda path: lora)['0']
    finally:
        if marks is not None:
            pipe.invert = invert


class _HostMark:
    """stand-in for a device event where there is no device (the gloo plumbing test)"""
    def __init__(self):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return 1e3 * (other.t - self.t)


def _event():
    if not torch.cuda.is_available():
        return _HostMark()
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def one_clip(pipe, data, ddim_steps, marks=None):
    """The hot path for one clip: inversion (B=1) then guided sampling (B=2).  Every clip starts with cold
    step-invariant caches (time-embedding rows, text K/V): nothing computed for one clip is reused by the next.
    `marks` receives a device event between the two loops (R1s = the sampling half alone; no host sync)."""
    pipe.unet.clear_step_caches()
    inv = pipe.invert(latents=data['latents'], prompt_embeds=data['text'], num_inference_steps=ddim_steps).latents
    if marks is not None:
        marks.append(_event())
    out = pipe(prompt=None, conditions=None, prompt_embeds=data['text'], negative_prompt_embeds=data['negative'],
               latents=inv, num_inference_steps=ddim_steps, guidance_scale=7.5, output_type='latent').videos
    return out


def cpu_baseline(frames_sample=2, latent=(64, 64)):
    """The oracle (CPU port of the reference's PyTorch path, fp32) on the host cores, on a bounded sample of what one
    DDIM step pair costs (1 inversion step = a B=1 UNet forward, 1 CFG step = a B=2 forward): both
    forwards at the full SD-1.5 width and the same 64x64 latent with T = `frames_sample` frames (a full T=16 step pair is
    ~5 min on 128 cores; the bench contract asks for ~10-30 s).  One warm-up forward (B=1), then each forward once.
    -> UNet frame-evaluations per second over the pair (3 * T evals)."""
    from oracle import unet3d
    torch.manual_seed(0)
    # PyTorch's default intra-op thread count (= physical cores).  Forcing os.cpu_count() (the SMT thread count) made
    # the same forward 3.5x SLOWER on the 128-core box (round-2 run 1: 41 s instead of 11.8 s for T = 2)
    threads = torch.get_num_threads()
    model = unet3d.AnimateDiffUNet3DModel(**unet3d.full_config()).eval()
    for n, p in model.named_parameters():           # proj_out is zero-initialised: make the temporal path live
        if 'temporal_transformer.proj_out' in n:
            torch.nn.init.normal_(p, std=0.02)
    txt = torch.randn(2, 77, 768)
    times = {}
    lh, lw = latent
    with torch.no_grad():
        x1 = torch.randn(1, 4, frames_sample, lh, lw)
        model(x1, torch.tensor(481), txt[:1])       # warm-up (allocator, thread pool, oneDNN primitive caches)
        for b in (1, 2):
            x = torch.randn(b, 4, frames_sample, lh, lw)
            t0 = time.time()
            model(x, torch.tensor(481), txt[:b])
            times[b] = time.time() - t0
    evals_per_s = 3 * frames_sample / (times[1] + times[2])
    return evals_per_s, threads, (f'1 inversion step (UNet B=1) + 1 CFG step (UNet B=2) at T={frames_sample}, '
                                  f'{lh}x{lw} latent, fp32, after 1 warm-up forward: {times[1]:.1f} s + {times[2]:.1f} s')


def torch_rocm_baseline(device, frames=16, latent=(64, 64), repeats=2):
    """Second stated baseline: the SAME oracle module (plain PyTorch, the reference's GPU-style path: fp16 weights,
    torch-ROCm eager ops = rocBLAS / MIOpen / SDPA) on this GPU — 1 inversion step (B=1) + 1 CFG step (B=2) at the full
    T and latent, warm-up + median, extrapolated to the 50 + 50 steps (every step is identical work).  Runs AFTER the
    timed region; never on the product path."""
    from oracle import unet3d
    with torch.device(device):
        model = unet3d.AnimateDiffUNet3DModel(**unet3d.full_config()).eval()
    for n, p in model.named_parameters():
        if 'temporal_transformer.proj_out' in n:
            torch.nn.init.normal_(p, std=0.02)
    model = model.half()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 4, frames, latent[0], latent[1], generator=g).to(device, torch.float16)
    txt = torch.randn(2, 77, 768, generator=g).to(device, torch.float16)

    def pair():
        model(x[:1], torch.tensor(481), txt[:1])
        model(x, torch.tensor(481), txt)
    times = []
    with torch.no_grad():
        pair()
        torch.cuda.synchronize()
        for _ in range(repeats):
            t0 = time.perf_counter()
            pair()
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
    dt = sorted(times)[len(times) // 2]
    del model
    torch.cuda.empty_cache()
    return dt


def long_clip_exchange(unet, shard, args, world, lh, lw):
    """configs[3]: what a rank exchanges per UNet forward — expected (closed form) beside what FrameShard
    counted over the timed clips, so that the first multi-GPU run can be checked against the design figure.
    exchange='sites': every motion module re-shards its activation frames -> sites and back: 2 * (N-1)/N * B * f_local *
    hw * C * 2 bytes per module; exchange='kv': every temporal attention receives the other ranks' K|V rows:
    (N-1) * B * f_local * hw * 2C * 2 bytes per attention layer (two per motion module)."""
    chans = list(getattr(unet.config, 'block_out_channels', (320, 640, 1280, 1280)))
    f_local = args.frames // world
    modules = []                                    # (sites, channels) of every motion module: 2 per down level, 3 per up level
    for lvl, c in enumerate(chans):
        hw = (lh >> lvl) * (lw >> lvl)
        modules += [(hw, c)] * (2 + 3)
    per_b = {'sites': sum(2 * (world - 1) / world * f_local * hw * c * 2 for hw, c in modules),
             'kv': sum(2 * (world - 1) * f_local * hw * 2 * c * 2 for hw, c in modules)}
    forwards = args.steps * args.ddim_steps * 2     # one B=1 and one B=2 forward per DDIM step pair
    mean_b = 1.5
    rec = {'ranks': world, 'frames_per_rank': f_local, 'mode': args.exchange,
           'expected_bytes_per_rank_per_forward_B1': {k: round(v) for k, v in per_b.items()},
           'expected_note': 'B = 2 forwards move twice as much; "auto" uses the site re-shard wherever a level\'s site count '
                            'splits over the ranks (every level of a 64x64 latent up to 8 ranks)'}
    if shard is not None:
        rec['counted_bytes_per_rank_per_forward_mean'] = round(shard.bytes_gathered / max(forwards, 1))
        key = 'kv' if args.exchange == 'kv' else 'sites'
        rec['expected_mean_for_counted'] = round(per_b[key] * mean_b)
        rec['backend'] = shard.backend
    return rec


def gemm_traffic(frames, latent, cps):
    """HBM bytes per vsx_gemm_f16 launch from the PMC passes of tools/pmc_by_shape.sh — only if the entry for THIS batch
    (profiles/gemm_hbm_traffic.json: {"b1": {...}, "b4": {...}}, one entry per clips-per-step) was measured on THIS build of
    the library (source digest) at the benchmark shape; a stale entry is not a measurement."""
    from videoswap_amd.build import source_digest
    path = os.path.join(ROOT, 'profiles', 'gemm_hbm_traffic.json')
    if not os.path.exists(path) or frames != 16 or latent != 64:
        return None, None, 'no PMC traffic file for this shape'
    with open(path) as f:
        t = json.load(f)
    if 'hbm_bytes_per_launch' in t:                  # (a file with a single, unkeyed entry)
        t = {'b%d' % int(t.get('clips_per_step', 1)): t}
    t = t.get('b%d' % cps)
    if t is None:
        return None, None, f'no PMC traffic entry for {cps} clip(s) per step'
    if t.get('lib_digest') != source_digest():
        return None, None, f'PMC traffic entry is from another build ({str(t.get("lib_digest"))[:12]})'
    return (round(t['hbm_bytes_per_launch']), round(t.get('algorithmic_bytes_per_launch', 0)) or None,
            f'{t["launches"]} launches of one inversion + one CFG step at {cps} clip(s) per step')


def roofline_object(roof, elapsed, stride, traffic_key):
    """The `roofline` object of one leg from the sampled launches (ops.prof_collect_roofline) of its timed region.
    traffic_key: (frames, latent or -1, clips per step) for the PMC file."""
    n_launch, gemm_ms, gemm_flop = roof['n'], roof['ms'], roof.get('flop', 0.0)
    if not (n_launch > 0 and gemm_ms > 0):
        return None
    ach = gemm_flop / (gemm_ms * 1e-3) / 1e12
    traffic, traffic_alg, traffic_note = gemm_traffic(*traffic_key)
    return {'bound': 'mfma', 'kernel': 'vsx_gemm_f16 (implicit-GEMM conv + GEMM, all shapes)',
            'achieved': round(ach, 1), 'peak': MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
            'frac': round(ach / MFMA_PEAK_TFLOPS, 4), 'traffic': traffic, 'traffic_source': traffic_note,
            'traffic_ratio': round(traffic / traffic_alg, 3) if traffic and traffic_alg else None,
            'traffic_algorithmic': traffic_alg,
            # both rooflines per launch: a launch cannot finish before max(FLOP / 2.5 PF/s, algorithmic bytes / 6.29 TB/s);
            # the sum of those floors over the sampled launches / their measured time
            'frac_of_attainable': round(roof['floor_ms'] / gemm_ms, 4),
            'byte_bound_time_share': round(roof['byte_bound_ms'] / gemm_ms, 4),
            'byte_peak_TBps': HBM_COPY_TBPS,
            'algorithmic_GB_per_s': round(roof['bytes'] / (gemm_ms * 1e-3) / 1e9, 1),
            'launches_sampled': int(n_launch), 'sample_stride': stride,
            'sampling': 'every %d-th GEMM launch' % stride,
            'avg_launch_us': round(1000.0 * gemm_ms / n_launch, 2),
            'avg_launch_gflop': round(gemm_flop / n_launch / 1e9, 2),
            'kernel_time_share_of_wall': round(gemm_ms * 1e-3 * stride / elapsed, 4)}


def stack_clips(group):
    """B clips denoised together: latents [B,4,T,h,w], text / negative [B,77,768] (the reference's batch axis, pipeline_videoswap.py:
    478-550: `batch_size = len(prompt)`).  configs[2]'s conditions stay per clip and are only used with one clip per step."""
    if len(group) == 1:
        return group[0]
    return dict(latents=torch.cat([c['latents'] for c in group]), text=torch.cat([c['text'] for c in group]),
                negative=torch.cat([c['negative'] for c in group]), conditions=group[0]['conditions'])


def timed_clips(run_clip, batches, ddim_steps, n_steps, barrier, marks=None):
    """EXACTLY n_steps steps (one step = one batch of clips through inversion + sampling) between two barriers."""
    barrier()
    t0 = time.perf_counter()
    for i in range(n_steps):
        if marks is not None:
            marks.append(_event())                   # clip start | (inside the clip) end of the inversion | clip end
        run_clip(batches[i % len(batches)], ddim_steps, marks)
        if marks is not None:
            marks.append(_event())
    barrier()
    return time.perf_counter() - t0


def workload_string(args, cps, lh, lw, swap=False, longclip=False):
    if cps == 1 or swap or longclip:
        head = f'BASELINE.json configs[{args.config - 1}]: {args.frames}-frame {lw * 8}x{lh * 8} clip'
        tail = 'the ranks share ONE clip per step' if longclip else '1 clip per GPU per step'
    else:       # a batch is not configs[1]: it is configs[1] clips batched (cf. configs[4], the clip-parallel throughput mode)
        head = (f'{cps} independent BASELINE.json configs[1] clips batched ({args.frames} frames, {lw * 8}x{lh * 8} each; cf. configs[4] '
                f'"throughput mode"; NOT the shape the reference\'s test.py drives, which is one clip)')
        tail = f'{cps} clips per GPU per step, denoised together: the batch axis of the latents'
    return (f'{head}, SD-1.5 UNet3D + AnimateDiff motion modules, {args.ddim_steps}-step DDIM inversion (B={cps}) + '
            f'{args.ddim_steps}-step CFG-7.5 DDIM sampling (B={2 * cps}), {tail}'
            + ('; full swap path (VideoSwapPipeline.validation): AttentionStore during the inversion, '
               'ED-LoRA merge + per-layer text embeddings [2,16,77,768], point-adapter residuals for '
               'sampling steps 0-25, AttentionRefine + latent / self-attention SpatialBlenders '
               '(use_blend), weights restored' if swap else ''))


def main():
    args = parse()
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch multi-GPU runs with torch.distributed.run (one process per GPU)')
    # VSX_FORCE_DISTRIBUTED=1 (tools/gpu_steps.sh dist1, under torchrun --nproc-per-node 1): take every multi-rank branch — the
    # nccl process group, dist.barrier, max-over-ranks, FrameShard on the RCCL communicator with its side-stream event ordering —
    # on ONE rank, so that the driver's first multi-GPU run is not the first execution of those lines
    distributed = world > 1 or os.environ.get('VSX_FORCE_DISTRIBUTED') == '1'
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
        device = torch.device('cuda', local_rank)
    else:       # only the gloo plumbing test (tests/test_distributed.py) gets here, with the clip loop stubbed
        device = torch.device('cpu')
    if distributed:
        import torch.distributed as dist
        if device.type == 'cuda':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group('gloo')

    from videoswap_amd import ops
    from videoswap_amd.distributed import max_over_ranks as max_over
    from videoswap_amd.synthetic import synthetic_clip
    swap = args.config == 3
    longclip = args.config == 4
    cps = max(args.clips_per_step, 1)
    if swap or longclip:
        cps = 1             # configs[2] carries per-clip conditions / controllers, configs[3] is ONE clip by definition
    lh, lw = args.latent_h or args.latent, args.latent_w or args.latent
    stub = os.environ.get('VSX_BENCH_STUB_CLIP') == '1'      # CPU plumbing test (tests/test_distributed.py): no model, no kernels
    if stub and device.type == 'cuda':
        raise SystemExit('VSX_BENCH_STUB_CLIP=1 replaces the clip loop with a sleep: refused on a box with a GPU (the line it '
                         'prints is labelled, but must never be mistaken for a measurement)')
    pipe = None if stub else build_pipeline(device, args.frames, swap=swap)
    shard = None
    n_batches = max(args.steps, 1)
    if longclip:
        # configs[3]: ONE clip for the whole job (same seed on every rank), rank r denoises frames [r*T/N, (r+1)*T/N)
        clips = [synthetic_clip(seed=7000 + i, frames=args.frames, height=lh, width=lw, device=device)
                 for i in range(n_batches)]
        if distributed and not stub:
            from videoswap_amd.distributed import FrameShard
            shard = FrameShard(args.frames, exchange=args.exchange)          # nccl process group -> the C-ABI collectives
            shard.install(pipe.unet)
            for c in clips:
                c['latents'] = shard.local_slice(c['latents'])
    else:
        # every rank owns different clips (seeded by rank); inputs resident in HBM before the timed region
        clips = [synthetic_clip(seed=1000 * rank + i, frames=args.frames, height=lh, width=lw,
                                device=device) for i in range(n_batches * cps)]
    batches = [stack_clips(clips[i * cps:(i + 1) * cps]) for i in range(n_batches)] if not longclip else clips
    if os.environ.get('VSX_GEMM_LOG') and rank == 0:      # tools/pmc_by_shape.py: which workload the logged launches belong to
        with open(os.environ['VSX_GEMM_LOG'] + '.meta.json', 'w') as f:
            json.dump({'clips_per_step': cps, 'frames': args.frames, 'latent': [lh, lw], 'config': args.config}, f)
    marks = []
    if swap:
        lora = synthetic_edlora(pipe.unet.state_dict())
        run_clip = lambda data, steps, m=None: swap_clip(pipe, data, steps, lora, m)      # noqa: E731
    else:
        run_clip = lambda data, steps, m=None: one_clip(pipe, data, steps, m)             # noqa: E731
    if stub:                                              # the metric / timing / collective code only
        def run_clip(data, steps, m=None):                # noqa: F811
            if m is not None:
                m.append(_event())
            time.sleep(0.01)

    def barrier():
        if distributed:
            dist.barrier()
        if device.type == 'cuda':
            torch.cuda.synchronize()

    for i in range(args.warmup):
        run_clip(batches[i % len(batches)], args.ddim_steps)
    barrier()

    if shard is not None:
        shard.bytes_gathered = 0
    ops.FlopCounter.reset(True)
    prof = args.prof_samples > 0 and device.type == 'cuda'
    if prof:
        ops.prof_enable(True, args.prof_samples, stride=args.prof_stride)
    from videoswap_amd.telemetry import BoardPower
    with BoardPower(local_rank if device.type == 'cuda' else -1) as bpw:      # host thread reading sysfs: nothing on the stream
        elapsed = timed_clips(run_clip, batches, args.ddim_steps, args.steps, barrier, marks)
    # device time of the two halves of every clip (events on the launch stream; nothing was synchronised in between)
    inv_s = sum(marks[3 * i].elapsed_time(marks[3 * i + 1]) for i in range(args.steps)) * 1e-3
    smp_s = sum(marks[3 * i + 1].elapsed_time(marks[3 * i + 2]) for i in range(args.steps)) * 1e-3
    if distributed:
        smp_s = max_over(smp_s, device)
        inv_s = max_over(inv_s, device)
    ops.FlopCounter.enabled = False
    roof = ops.prof_collect_roofline(MFMA_PEAK_TFLOPS * 1e12, HBM_COPY_TBPS * 1e12) if prof else dict(n=0, ms=0.0)
    if prof:
        ops.prof_enable(False, 0)

    elapsed = max_over(elapsed, device)              # whole-job time = slowest rank

    # clip-parallel: every rank its own clips (weak scaling); long clip: the ranks share ONE clip per step (strong scaling)
    clips_job = args.steps if longclip else world * args.steps * cps
    frames_total = clips_job * args.frames
    value = frames_total / elapsed
    total_flop = (ops.FlopCounter.gemm + ops.FlopCounter.attention) * world
    evals = clips_job * args.frames * 3 * args.ddim_steps                   # (1 + 2) UNet frame-evals per DDIM step
    per_clip = max(args.steps * cps, 1)
    batch_note = '' if cps == 1 else f', {cps} independent clips denoised together per step (throughput mode: latents [{cps},4,T,h,w])'
    metric = (f'denoised frames/sec, {args.frames}-frame 512^2 long clip @ {args.ddim_steps} DDIM steps, frame axis sharded over the GPUs '
              '(end to end: inversion + CFG sampling)' if longclip else
              f'denoised frames/sec, {args.frames}-frame {lw * 8}x{lh * 8} clip @ {args.ddim_steps} DDIM steps (end to end: inversion + CFG '
              f'sampling){batch_note}')
    if stub:
        metric = 'STUBBED CLIP LOOP (VSX_BENCH_STUB_CLIP=1: no model, no kernels; plumbing test only) - ' + metric
    out = {
        'metric': metric,
        'value': round(value, 4), 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(1000.0 * elapsed / max(args.steps, 1), 2), 'higher_is_better': True,
        'scaling': 'strong' if longclip else 'weak',
        'vs_baseline': None, 'dtype': 'f16', 'data': 'synthetic',
        'config': {'launch': 'eager',
                   'workload': workload_string(args, cps, lh, lw, swap, longclip),
                   'latents': [cps, 4, args.frames, lh, lw],
                   'parallelism': (f'frame-sharded x{world} ({args.frames // world} frames per rank, exchange={args.exchange})'
                                   if longclip else f'clip-parallel x{world}')},
        'readings': {'R1e_frames_per_s': round(value, 4),
                     # R1s: the guided-sampling half alone (frames / device time of the 50 CFG steps)
                     'R1s_frames_per_s': round(frames_total / smp_s, 4) if smp_s > 0 else None,
                     'inversion_s_per_clip': round(inv_s / per_clip, 4),
                     'sampling_s_per_clip': round(smp_s / per_clip, 4),
                     'R2_unet_frame_evals_per_s': round(evals / elapsed, 2),
                     # FLOP the launches multiply; `..._reference_form` adds what the reference's formulation multiplies on top
                     # (nine taps on the upsampled image where the sub-pixel form of Upsample3D's convolution runs four)
                     'loop_algorithmic_tflop': round(total_flop / 1e12, 1),
                     'loop_reference_form_tflop': round((total_flop + ops.FlopCounter.gemm_saved * world) / 1e12, 1),
                     'loop_tflops': round(total_flop / elapsed / 1e12, 1),
                     'loop_mfma_frac': round(total_flop / elapsed / 1e12 / (MFMA_PEAK_TFLOPS * world), 4),
                     'ceiling_R1e_at_100pct_mfma': 15.1},
    }
    if bpw.summary() is not None:           # rank 0's board over the timed region (MI355X: 1 400 W cap)
        out['readings']['board_power'] = bpw.summary()
        out['readings']['board_power']['joules_per_frame'] = round(bpw.summary()['mean_W'] * elapsed / max(frames_total / world, 1), 1)
    if distributed and world == 1:
        out['config']['forced_distributed'] = 'VSX_FORCE_DISTRIBUTED=1: the multi-rank branches on one rank (%s)' % dist.get_backend()
    if longclip and not stub:
        out['exchange'] = long_clip_exchange(pipe.unet, shard, args, world, lh, lw)
    if stub:
        out['config']['stub'] = True
    plain = lh == lw == args.latent and args.config == 2
    rl = roofline_object(roof, elapsed, args.prof_stride, (args.frames, args.latent if plain else -1, cps))
    if rl is not None:
        # footnotes, not rooflines: (1) the rate a bare MFMA stream reaches on real fp16 data under this board's power management,
        # (2) the nominal peak scaled to the mean shader clock THIS run's timed region was held at (readings.board_power)
        note = {'mfma_stream_on_real_data': MFMA_REAL_DATA_TFLOPS, 'frac': round(rl['achieved'] / MFMA_REAL_DATA_TFLOPS, 4),
                'source': 'tools/ubench/mfma_power.hip: v_mfma_f32_32x32x16_f16 on normal(0, 1) operands, registers only, 256 CUs: '
                          '1 580 - 1 679 TF/s at 1.55 - 1.76 GHz (2 459 - 2 467 on zeros at 2.40 GHz); the roofline object prices '
                          'against the nominal 2 500'}
        bp = out['readings'].get('board_power') or {}
        if bp.get('sclk_mean_MHz'):
            pk = MFMA_PEAK_TFLOPS * bp['sclk_mean_MHz'] / NOMINAL_SCLK_MHZ
            note['peak_at_measured_clock'] = round(pk, 1)
            note['frac_at_measured_clock'] = round(rl['achieved'] / pk, 4)
        out['readings']['gemm_tflops_vs_power_managed_peak'] = note
        out['roofline'] = rl
    tcl = args.throughput_clips
    if (rank == 0 and world == 1 and args.config == 2 and not args.no_extra_reading and not distributed and not stub
            and device.type == 'cuda' and tcl > 1 and tcl != cps and args.throughput_steps > 0):
        # SECOND LEG, never `value`: `tcl` independent configs[1] clips denoised together (UNet batch tcl in the inversion, 2 tcl
        # under CFG), fresh synthetic clips, one warm-up step + throughput_steps timed steps between barriers, its own sampled
        # roofline and its own PMC traffic entry
        try:
            nb = args.throughput_steps
            extra = [synthetic_clip(seed=1000 * rank + 100 + i, frames=args.frames, height=lh, width=lw, device=device)
                     for i in range(tcl * nb)]
            tb = [stack_clips(extra[i * tcl:(i + 1) * tcl]) for i in range(nb)]
            run_clip(tb[0], args.ddim_steps)
            barrier()
            if prof:
                ops.prof_enable(True, args.prof_samples, stride=args.prof_stride)
            with BoardPower(local_rank) as bpw2:
                t2 = timed_clips(run_clip, tb, args.ddim_steps, nb, barrier)
            roof2 = ops.prof_collect_roofline(MFMA_PEAK_TFLOPS * 1e12, HBM_COPY_TBPS * 1e12) if prof else dict(n=0, ms=0.0)
            if prof:
                ops.prof_enable(False, 0)
            v2 = tcl * nb * args.frames / t2
            out['throughput_mode'] = {
                'clips_per_step': tcl, 'value': round(v2, 4), 'unit': 'frames/s', 'steps': nb, 'warmup': 1,
                'ms_per_step': round(1e3 * t2 / nb, 1), 'ratio_to_value': round(v2 / value, 4),
                'latents': [tcl, 4, args.frames, lh, lw],
                'workload': workload_string(args, tcl, lh, lw),
                'roofline': roofline_object(roof2, t2, args.prof_stride, (args.frames, args.latent if plain else -1, tcl)),
                'board_power': bpw2.summary()}
            if out['throughput_mode']['board_power']:
                out['throughput_mode']['board_power']['joules_per_frame'] = round(
                    out['throughput_mode']['board_power']['mean_W'] * t2 / (tcl * nb * args.frames), 1)
            del extra, tb
        except Exception as e:  # the second leg must never take the headline down with it
            out['throughput_mode'] = {'clips_per_step': tcl, 'value': None, 'note': f'failed: {e!r}'}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and device.type == 'cuda':
        try:
            evals_per_s, threads, sample = cpu_baseline(args.cpu_frames, (lh, lw))
            out['cpu_baseline'] = {'value': round(evals_per_s / (3 * args.ddim_steps), 6), 'unit': 'frames/s',
                                   'cores': threads, 'kind': 'port', 'sample': sample,
                                   'unet_frame_evals_per_s': round(evals_per_s, 4)}
        except Exception as e:  # the baseline must never take the GPU number down with it
            out['cpu_baseline'] = {'value': None, 'unit': 'frames/s', 'cores': torch.get_num_threads(),
                                   'kind': 'port', 'sample': f'failed: {e!r}'}
        try:
            del pipe, clips, batches
            torch.cuda.empty_cache()
            pair_s = torch_rocm_baseline(device, args.frames, (lh, lw))
            out['torch_rocm_eager_fp16'] = {
                'value': round(args.frames / (args.ddim_steps * pair_s), 4), 'unit': 'frames/s',
                'sample': f'oracle module (plain PyTorch-ROCm eager, fp16) on the same GPU: 1 inversion step (B=1) + 1 CFG '
                          f'step (B=2) at T={args.frames}, median of 2 after warm-up = {pair_s * 1e3:.0f} ms, x {args.ddim_steps} steps',
                'speedup_of_value': round(value * args.ddim_steps * pair_s / args.frames, 2)}
        except Exception as e:
            out['torch_rocm_eager_fp16'] = {'value': None, 'unit': 'frames/s', 'sample': f'failed: {e!r}'}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
