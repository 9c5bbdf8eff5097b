"""Full-width parity: the SD-1.5-width UNet3D + motion modules that bench.py times (block_out_channels 320/640/1280/
1280, head dims 40/80/160, 64x64 and 56x96 latents) against the oracle.

The tiny-width goldens (tests/test_unet_gpu.py) dispatch different kernels than the benchmarked model: at full width the
library takes the 256x320 16-wave tile for the big-M convolutions / GEGLU, the N = 4096 d = 40 flash-attention path with
XCD-ordered workgroups, the MFMA temporal kernels (d = 40 / 80 / 160) and split-K at the real 16x16 / 8x8 shapes.  This
file pins exactly those.

Oracle placement.  `oracle.unet3d` is plain PyTorch, so the SAME fp32 oracle module also runs on the GPU box's device
(torch-ROCm fp32: rocBLAS / MIOpen, no TF32 on gfx950).  Test (1) runs the fp32 oracle on the HOST cores (~100 s on
the 128-core box for B*F = 16 frames at 64x64) and checks both the product and the device-placed fp32 oracle against it;
the larger cases (the B = 2, T = 16 benchmark shape, 56x96, sequential steps, the 50+50-step drift) then use the
device-placed fp32 oracle, which test (1) has just shown to agree with the host one to ~1e-5.

Tolerance rule: rel-L2(product, fp32 oracle) <= 2 x rel-L2(fp16-storage oracle, fp32 oracle), both
measured in the test; cosine similarity is printed.  Sequential loops compound fp16 error, so the same rule is applied
to the final latents, with the yardstick run through the same number of steps.
"""
import copy
import json
import os
import time

import pytest
import torch

from util import ROOT, cosine, rel_l2

pytestmark = pytest.mark.gpu

RESULTS = {}


def _record(name, **kw):
    RESULTS[name] = kw
    out = os.path.join(ROOT, 'gpurun_out')
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, 'parity_fullwidth.json'), 'w') as f:
            json.dump(RESULTS, f, indent=1, sort_keys=True)
    except OSError:
        pass
    print(name, json.dumps(kw))


@pytest.fixture(scope='module')
def models():
    return build_models()


def build_models():
    """Everything is built ON the device (seeded generator of videoswap_amd.synthetic: seconds instead of the minute
    a CPU initialisation of 860 M parameters takes); the oracle copies receive the product's fp16-exact weights."""
    from oracle import unet3d
    from videoswap_amd.synthetic import synth_weights_
    from videoswap_amd.unet import AnimateDiffUNet3DModel
    cfg = unet3d.full_config()
    t0 = time.time()
    with torch.device('cuda'):
        prod = AnimateDiffUNet3DModel(**cfg)
        ora_dev = unet3d.AnimateDiffUNet3DModel(**cfg).eval()            # fp32 oracle, device-placed
    prod = synth_weights_(prod, seed=1234).half().eval()
    missing, unexpected = ora_dev.load_state_dict({k: v.float() for k, v in prod.state_dict().items()}, strict=True)
    assert not missing and not unexpected
    ora_h = copy.deepcopy(ora_dev).half()                                # fp16-storage oracle: the error yardstick
    print(f'[fullwidth] models built in {time.time() - t0:.1f} s')
    return cfg, None, ora_dev, ora_h, prod


def _inputs(B, T, H, W, seed, layers=None):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 4, T, H, W, generator=g)
    shape = (B, 77, 768) if layers is None else (B, layers, 77, 768)
    return x, torch.randn(*shape, generator=g)


@torch.no_grad()
def _fwd(model, x, t, txt, res=None):
    dev = next(model.parameters()).device
    dt = next(model.parameters()).dtype
    r = None if res is None else [v.to(dev, dt) for v in res]
    out = model(x.to(dev, dt), torch.tensor(t), txt.to(dev, dt), down_block_additional_residuals=r)
    out = out.sample if hasattr(out, 'sample') else out[0]
    if dev.type == 'cuda':
        torch.cuda.synchronize()
    return out.float().cpu()


# Absolute caps (VERDICT round 4, next 5 iii): what the seeded synthetic weights have shown over four rounds is 1.3e-3 for one
# forward and 0.8 - 1.0e-2 for the final latents of the loops; a product error several times that would still pass a rule that is
# only relative to a yardstick, so the non-stress cases are also held to these.
CAP_FORWARD = 3e-3
CAP_FINAL_LATENTS = 2e-2


def _check(name, prod_out, ref, half_out, extra=None, cap=CAP_FORWARD):
    """The parity rule of one UNet forward.  Relative to the yardstick (the oracle with fp16 storage, against the same fp32
    result): (a) rel-L2 <= 2 x yardstick, (b) no frame worse than 4 x, (c) cosine >= the yardstick's - 1e-4, (d) the product
    stays within 1.5 x the yardstick's own error OF the fp16-storage oracle (two fp16 evaluations of the same network may differ
    by about the root sum of squares of their errors, not by more); absolute: (e) rel-L2 <= `cap` (None for the stress cases,
    whose yardstick is itself large)."""
    assert torch.isfinite(prod_out).all()
    e, e16 = rel_l2(prod_out, ref), rel_l2(half_out, ref)
    # per-frame check: no single frame (= one slab of M tiles) may hide behind the others
    B, C, T = ref.shape[:3]
    worst = max(rel_l2(prod_out[b, :, f], ref[b, :, f]) for b in range(B) for f in range(T))
    cos_p, cos_h = cosine(prod_out, ref), cosine(half_out, ref)
    e_ph = rel_l2(prod_out, half_out)
    _record(name, rel_l2=e, rel_l2_fp16_oracle=e16, cosine=cos_p, cosine_fp16_oracle=cos_h, worst_frame_rel_l2=worst,
            product_vs_fp16_oracle_rel_l2=e_ph, **(extra or {}))
    assert e <= 2 * e16, f'{name}: rel-L2 {e:.3e} > 2 x fp16-storage oracle error {e16:.3e}'
    assert worst <= 4 * e16, f'{name}: worst frame rel-L2 {worst:.3e} vs fp16-storage oracle error {e16:.3e}'
    assert cos_p >= cos_h - 1e-4, f'{name}: cosine {cos_p:.6f} below the fp16-storage oracle\'s {cos_h:.6f}'
    assert e_ph <= 1.5 * e16, f'{name}: product vs fp16-storage oracle {e_ph:.3e} > 1.5 x that oracle\'s own error {e16:.3e}'
    if cap is not None:
        assert e <= cap, f'{name}: rel-L2 {e:.3e} above the absolute cap {cap:.1e}'


def _check_loops(name, inv_p, out_p, inv_ref, out_ref, inv_h, out_h, extra=None):
    """The same rule for the two loops: inverted latents and final latents, yardstick pushed through the same steps."""
    assert torch.isfinite(out_p).all() and torch.isfinite(inv_p).all()
    e_inv, e16_inv = rel_l2(inv_p, inv_ref), rel_l2(inv_h, inv_ref)
    e_out, e16_out = rel_l2(out_p, out_ref), rel_l2(out_h, out_ref)
    cos_p, cos_h = cosine(out_p, out_ref), cosine(out_h, out_ref)
    e_ph = rel_l2(out_p, out_h)
    _record(name, inversion_rel_l2=e_inv, inversion_rel_l2_fp16_oracle=e16_inv, final_rel_l2=e_out,
            final_rel_l2_fp16_oracle=e16_out, final_cosine=cos_p, final_cosine_fp16_oracle=cos_h,
            product_vs_fp16_oracle_rel_l2=e_ph, **(extra or {}))
    assert e_inv <= 2 * e16_inv + 1e-4, f'{name}: inversion drift {e_inv:.3e} vs fp16-storage oracle {e16_inv:.3e}'
    assert e_out <= 2 * e16_out + 1e-4, f'{name}: final-latent drift {e_out:.3e} vs fp16-storage oracle {e16_out:.3e}'
    assert cos_p >= cos_h - 1e-4, f'{name}: final cosine {cos_p:.6f} below the fp16-storage oracle\'s {cos_h:.6f}'
    assert e_ph <= 1.5 * e16_out + 1e-4, f'{name}: product vs fp16-storage oracle {e_ph:.3e} > 1.5 x its own error {e16_out:.3e}'
    assert e_out <= CAP_FINAL_LATENTS and e_inv <= CAP_FINAL_LATENTS, f'{name}: {e_inv:.3e} / {e_out:.3e} above the absolute cap'


def test_forward_vs_host_oracle(models):
    """(1) B = 1, T = 2, 64x64 with the fp32 oracle on the HOST cores (~40 s on the 128-core box): anchors the
    device-placed fp32 oracle the larger cases use, and checks the product against the host oracle directly."""
    cfg, _, ora_dev, ora_h, prod = models
    ora = copy.deepcopy(ora_dev).cpu()
    x, txt = _inputs(1, 2, 64, 64, seed=101)
    t0 = time.time()
    ref = _fwd(ora, x, 481, txt)
    host_s = time.time() - t0
    ref_dev = _fwd(ora_dev, x, 481, txt)
    e_dev = rel_l2(ref_dev, ref)
    print(f'host fp32 oracle {host_s:.1f} s on {os.cpu_count()} cores; device-placed fp32 oracle vs host: {e_dev:.2e}')
    assert e_dev < 2e-4, 'the device-placed fp32 oracle must agree with the host oracle'
    _check('unet_B1_T2_64x64_vs_host_fp32_oracle', _fwd(prod, x, 481, txt), ref, _fwd(ora_h, x, 481, txt),
           extra=dict(host_oracle_s=host_s, device_oracle_vs_host=e_dev))


def test_forward_64x64_B2_T8(models):
    """(1b) B = 2, T = 8, 64x64: M = 65 536 rows at the top level -> big-tile convolutions, N = 4096 flash attention,
    MFMA temporal attention, split-K at the 8x8 level.  (Round-2 run 1 also checked this shape against the HOST
    oracle: rel-L2 1.30e-3, device-placed oracle vs host 1.7e-6 — profiles/r02_parity_fullwidth.json.)"""
    cfg, ora, ora_dev, ora_h, prod = models
    x, txt = _inputs(2, 8, 64, 64, seed=111)
    _check('unet_B2_T8_64x64', _fwd(prod, x, 481, txt), _fwd(ora_dev, x, 481, txt), _fwd(ora_h, x, 481, txt))


def test_forward_benchmark_shape(models):
    """(2) the exact CFG shape bench.py times: B = 2, T = 16, 64x64 (M = 131 072)."""
    cfg, ora, ora_dev, ora_h, prod = models
    x, txt = _inputs(2, 16, 64, 64, seed=102)
    _check('unet_B2_T16_64x64', _fwd(prod, x, 981, txt), _fwd(ora_dev, x, 981, txt), _fwd(ora_h, x, 981, txt))
    x1, txt1 = _inputs(1, 16, 64, 64, seed=103)
    _check('unet_B1_T16_64x64_inversion_step', _fwd(prod, x1, -19, txt1), _fwd(ora_dev, x1, -19, txt1),
           _fwd(ora_h, x1, -19, txt1))


def test_forward_clip_batch_shapes(models):
    """(2b) the UNet batches of bench.py's throughput leg, four clips per step: B = 4 in the inversion, B = 8 under CFG, T = 16,
    64x64 (M = 262 144 / 524 288 rows at the top level, 16 384 / 32 768 at 16x16, 4 096 / 8 192 at 8x8 — other tile choices than
    B = 1 / 2 everywhere, and the largest row counts the 32-bit element offsets of the kernels see).  The B = 4 launch and BOTH
    halves of the B = 8 launch go against the fp32 oracle under the full rule (the oracle runs the two halves as two B = 4
    forwards: batch items are independent and a B = 8 fp32 forward does not fit its attention scores); round 5 compared the
    second half only with the product's own B = 4 launch."""
    cfg, ora, ora_dev, ora_h, prod = models
    x, txt = _inputs(8, 16, 64, 64, seed=120)
    out8 = _fwd(prod, x, 521, txt)
    assert torch.isfinite(out8).all()
    ref4 = _fwd(ora_dev, x[:4], 521, txt[:4])
    out4 = _fwd(prod, x[:4], 521, txt[:4])
    half4 = _fwd(ora_h, x[:4], 521, txt[:4])
    _check('unet_B4_T16_64x64', out4, ref4, half4)
    _check('unet_B8_T16_64x64_first_half', out8[:4], ref4, half4, extra=dict(vs_B4_launch=rel_l2(out8[:4], out4)))
    # two fp16 evaluations of one network (other tiles, another fp32 summation order in the convolutions) differ by about the
    # root sum of squares of their errors: measured 1.5e-3 between the two launches, each 1.3e-3 from the fp32 oracle
    assert rel_l2(out8[:4], out4) <= 1.5 * rel_l2(half4, ref4)
    del ref4, half4, out4
    ref4b = _fwd(ora_dev, x[4:], 521, txt[4:])
    half4b = _fwd(ora_h, x[4:], 521, txt[4:])
    _check('unet_B8_T16_64x64_second_half', out8[4:], ref4b, half4b)
    assert not torch.equal(out8[0], out8[4])


def test_forward_56x96(models):
    """(3) 448x768 frames (56x96 latent, 26 of the 30 reference YAMLs): ragged M tiles, N = 5376 keys."""
    cfg, ora, ora_dev, ora_h, prod = models
    x, txt = _inputs(2, 4, 56, 96, seed=104)
    _check('unet_B2_T4_56x96', _fwd(prod, x, 501, txt), _fwd(ora_dev, x, 501, txt), _fwd(ora_h, x, 501, txt))


def test_forward_24_frames(models):
    """(3b) T = 24: the longest clip the reference's positional-encoding table allows (motion_module.py:237-255,
    `temporal_position_encoding_max_len: 24`); 24 query x 24 key frames per site in the temporal attention kernel."""
    cfg, ora, ora_dev, ora_h, prod = models
    x, txt = _inputs(1, 24, 64, 64, seed=124)
    _check('unet_B1_T24_64x64', _fwd(prod, x, 261, txt), _fwd(ora_dev, x, 261, txt), _fwd(ora_h, x, 261, txt))


def test_sequential_steps_448x768(models):
    """(5b) the loops at the size of 26 of the 30 reference option files (448 x 768 frames = 56 x 96 latents), T = 8,
    3 + 3 steps: ragged row tiles (M = 43 008 / 86 016) and N = 5 376 keys in every UNet call of both loops."""
    cfg, ora, ora_dev, ora_h, prod = models
    x, txt = _inputs(1, 8, 56, 96, seed=125)
    neg = torch.randn(1, 77, 768, generator=torch.Generator().manual_seed(8))
    inv_ref, out_ref = _oracle_loops(ora_dev, x, txt, neg, 3)
    inv_h, out_h = _oracle_loops(ora_h, x, txt, neg, 3)
    inv_p, out_p = _product_loops(prod, x, txt, neg, 3)
    _check_loops('loops_3+3_T8_56x96', inv_p, out_p, inv_ref, out_ref, inv_h, out_h)


def test_forward_edlora_text_and_adapter_residuals(models):
    """(4) config 3's inputs at full width: per-layer text embeddings [B,16,77,768] and the four adapter residuals."""
    from oracle import pipeline as opipe
    from videoswap_amd.edlora import revise_edlora_unet_attention_forward
    cfg, ora, ora_dev, ora_h, prod = models
    x, txt = _inputs(2, 4, 64, 64, seed=105, layers=16)
    g = torch.Generator().manual_seed(106)
    res = [torch.randn(8, c, 64 // s, 64 // s, generator=g) * 0.1 for c, s in ((320, 1), (640, 2), (1280, 4), (1280, 8))]
    opipe.use_edlora(ora_dev)
    opipe.use_edlora(ora_h)
    revise_edlora_unet_attention_forward(prod)
    try:
        _check('unet_B2_T4_64x64_edlora_adapter', _fwd(prod, x, 741, txt, list(res)), _fwd(ora_dev, x, 741, txt, list(res)),
               _fwd(ora_h, x, 741, txt, list(res)))
    finally:
        opipe.reset_processors(ora_dev)
        opipe.reset_processors(ora_h)
        from videoswap_amd.attention import AttnProcessor2_0
        for name, m in prod.named_modules():
            if m.__class__.__name__ == 'Attention' and 'attn2' in name:
                m.set_processor(AttnProcessor2_0())


def test_forward_outlier_stress(models):
    """(4b) activation-outlier stress of the fp16 epilogues: the seeded synthetic weights give
    well-behaved activations, real checkpoints do not.  A handful of GroupNorm / LayerNorm gains and one attention
    `to_out` / one feed-forward output weight are scaled x30 (in the product AND both oracles, restored afterwards), so
    that GEMM epilogues, the folded-LayerNorm identity, GEGLU and the fp16 residual stream see values two orders of
    magnitude above the rest — the folded LayerNorm (rstd * acc - rstd * mean * c1) is the place where a cancellation
    would show.  Same <= 2x fp16-storage-oracle rule; nothing may overflow to inf."""
    cfg, ora, ora_dev, ora_h, prod = models
    names = ['down_blocks.0.resnets.0.norm2.weight', 'down_blocks.1.attentions.0.norm.weight',
             'down_blocks.0.attentions.1.transformer_blocks.0.norm1.weight',
             'down_blocks.0.attentions.1.transformer_blocks.0.norm3.weight',
             'up_blocks.3.attentions.2.transformer_blocks.0.norm2.weight',
             'down_blocks.0.motion_modules.0.temporal_transformer.transformer_blocks.0.norms.0.weight',
             'up_blocks.3.motion_modules.1.temporal_transformer.transformer_blocks.0.ff_norm.weight',
             'up_blocks.2.attentions.0.transformer_blocks.0.attn1.to_out.0.weight',
             'mid_block.attentions.0.transformer_blocks.0.ff.net.2.weight',
             'up_blocks.3.resnets.1.norm1.weight']
    mods = (prod, ora_dev, ora_h)
    saved = []
    with torch.no_grad():
        for m in mods:
            sd = dict(m.named_parameters())
            for n in names:
                assert n in sd, n
                saved.append((sd[n], sd[n].detach().clone()))
                sd[n].mul_(30.0)
            if hasattr(m, 'bump_weights_epoch'):
                m.bump_weights_epoch()
    try:
        x, txt = _inputs(2, 4, 64, 64, seed=131)
        ref = _fwd(ora_dev, x, 621, txt)
        out = _fwd(prod, x, 621, txt)
        _check('unet_B2_T4_64x64_outlier_stress', out, ref, _fwd(ora_h, x, 621, txt),
               extra=dict(out_absmax=float(out.abs().max()), ref_absmax=float(ref.abs().max())), cap=None)
    finally:
        with torch.no_grad():
            for p_, v in saved:
                p_.copy_(v)
            for m in mods:
                if hasattr(m, 'bump_weights_epoch'):
                    m.bump_weights_epoch()


_ORACLE_LOOPS = {}      # (id(model), key) -> (inverted, final): the 50 + 50 oracle loops cost 45 s and two tests need the same ones


@torch.no_grad()
def _oracle_loops(model, x, txt, neg, steps, guidance=7.5, cache_key=None):
    if cache_key is not None and (id(model), cache_key) in _ORACLE_LOOPS:
        return _ORACLE_LOOPS[(id(model), cache_key)]
    r = _oracle_loops_uncached(model, x, txt, neg, steps, guidance)
    if cache_key is not None:
        _ORACLE_LOOPS[(id(model), cache_key)] = r
    return r


@torch.no_grad()
def _oracle_loops_uncached(model, x, txt, neg, steps, guidance=7.5):
    from oracle import pipeline as opipe
    dev, dt = next(model.parameters()).device, next(model.parameters()).dtype
    lat = opipe.invert(model, x.to(dev, dt), txt.to(dev, dt), steps)
    inv = lat.clone()
    out = opipe.sample(model, lat, txt.to(dev, dt), neg.to(dev, dt), steps, guidance)
    torch.cuda.synchronize()
    return inv.float().cpu(), out.float().cpu()


@torch.no_grad()
def _product_loops(prod, x, txt, neg, steps, guidance=7.5):
    from videoswap_amd.compat import SD15_SCHEDULER_CONFIG, DDIMScheduler
    from videoswap_amd.pipeline import VideoSwapPipeline
    pipe = VideoSwapPipeline(unet=prod, scheduler=DDIMScheduler(**SD15_SCHEDULER_CONFIG)).to('cuda')
    inv = pipe.invert(latents=x.half().cuda(), prompt_embeds=txt.half().cuda(), num_inference_steps=steps).latents
    out = pipe(prompt_embeds=txt.half().cuda(), negative_prompt_embeds=neg.half().cuda(), latents=inv,
               num_inference_steps=steps, guidance_scale=guidance, output_type='latent').videos
    torch.cuda.synchronize()
    return inv.float().cpu(), out.float().cpu()


def _loop_case(steps):
    x, txt = _inputs(1, 16, 64, 64, seed=107 + steps)
    return x, txt, torch.randn(1, 77, 768, generator=torch.Generator().manual_seed(7))


@pytest.mark.parametrize('steps', [5, 50])
def test_sequential_steps_full_width(models, steps):
    """(5) `steps` inversion steps (B = 1) + `steps` CFG-7.5 sampling steps (B = 2) at T = 16, 64x64: config 2 of
    BASELINE.json for steps = 50 — the very loop bench.py times (pipeline_videoswap.py:677-710, :555-601) — runs in
    every `-m gpu` pass since round 4 (about 45 s with the device-placed oracles; the 20 + 20 case it replaces cost 19 s).
    The final latents obey the same <= 2x rule against the fp16-storage oracle pushed through the same loops (fp16
    error compounds over sequential UNet calls)."""
    cfg, ora, ora_dev, ora_h, prod = models
    x, txt, neg = _loop_case(steps)
    t0 = time.time()
    inv_ref, out_ref = _oracle_loops(ora_dev, x, txt, neg, steps, cache_key=('loops', steps))
    t_ref = time.time() - t0
    t0 = time.time()
    inv_h, out_h = _oracle_loops(ora_h, x, txt, neg, steps, cache_key=('loops', steps))
    t_h = time.time() - t0
    _product_loops(prod, x, txt, neg, 1)          # first-call costs (weight packing, text K/V) outside the timing
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.time()
    e0.record()
    inv_p, out_p = _product_loops(prod, x, txt, neg, steps)
    e1.record()
    torch.cuda.synchronize()
    t_p, t_p_dev = time.time() - t0, e0.elapsed_time(e1) / 1e3
    _check_loops(f'loops_{steps}+{steps}_T16_64x64', inv_p, out_p, inv_ref, out_ref, inv_h, out_h,
                 extra=dict(wall_s_product=t_p, device_s_product=t_p_dev, wall_s_fp32_torch_oracle=t_ref,
                            wall_s_fp16_torch_oracle=t_h))


def test_clips_denoised_together_match_the_oracle_clip_by_clip(models):
    """(5c) bench.py's default workload denoises FOUR clips together (latents [4,4,T,h,w]: UNet batch 4 in the inversion, 8
    under CFG — the reference's own batch axis, pipeline_videoswap.py:478-550).  Every clip of the batch must obey the loop
    rule against the oracle run on that clip ALONE: nothing may leak between the clips of a batch (GroupNorm pools per batch
    item, the time-embedding row is shared, text rows are per item), and the larger M picks other kernels."""
    cfg, ora, ora_dev, ora_h, prod = models
    xs, txts, negs = [], [], []
    NC = 4
    for i in range(NC):
        x, txt = _inputs(1, 8, 64, 64, seed=151 + i)
        xs.append(x); txts.append(txt)
        negs.append(torch.randn(1, 77, 768, generator=torch.Generator().manual_seed(17 + i)))
    inv_p, out_p = _product_loops(prod, torch.cat(xs), torch.cat(txts), torch.cat(negs), 3)
    assert inv_p.shape[0] == NC and out_p.shape[0] == NC
    for i in range(NC):
        inv_ref, out_ref = _oracle_loops(ora_dev, xs[i], txts[i], negs[i], 3)
        inv_h, out_h = _oracle_loops(ora_h, xs[i], txts[i], negs[i], 3)
        _check_loops(f'loops_3+3_T8_64x64_clip{i}_of_a_batch_of_{NC}', inv_p[i:i + 1], out_p[i:i + 1], inv_ref, out_ref, inv_h, out_h)
    assert not torch.equal(out_p[0], out_p[1])


def test_four_clips_batched_at_depth_50_steps_T16(models):
    """(5d) bench.py's throughput leg at its real depth (the batched loop was pinned at 3 + 3 steps, T = 8):
    FOUR clips denoised together through 50 inversion steps (UNet batch 4) + 50 CFG-7.5 sampling steps (batch 8) at T = 16, 64x64
    (pipeline_videoswap.py:677-710, :555-601 with the batch axis of :478-550).  Clip 0 is the clip of the 50 + 50 single-clip case
    above: it goes against the fp32 oracle loop and the fp16-storage yardstick under `_check_loops` (the oracle loops are shared
    with that test: 45 s not spent twice).  Clips 1 - 3 go against the product's own SINGLE-clip runs of the same inputs — 200
    sequential UNet calls each way — within 1.5 x the yardstick's error (two fp16 evaluations of one network: other tiles,
    another fp32 summation order; 135 s of fp32 oracle not spent), and nothing may leak between the clips of a batch."""
    cfg, ora, ora_dev, ora_h, prod = models
    steps, NC = 50, 4
    x0, txt0, neg0 = _loop_case(steps)
    xs, txts, negs = [x0], [txt0], [neg0]
    for i in range(1, NC):
        x, txt = _inputs(1, 16, 64, 64, seed=171 + i)
        xs.append(x); txts.append(txt)
        negs.append(torch.randn(1, 77, 768, generator=torch.Generator().manual_seed(27 + i)))
    t0 = time.time()
    inv_p, out_p = _product_loops(prod, torch.cat(xs), torch.cat(txts), torch.cat(negs), steps)
    t_batch = time.time() - t0
    assert inv_p.shape[0] == NC and out_p.shape[0] == NC
    inv_ref, out_ref = _oracle_loops(ora_dev, x0, txt0, neg0, steps, cache_key=('loops', steps))
    inv_h, out_h = _oracle_loops(ora_h, x0, txt0, neg0, steps, cache_key=('loops', steps))
    _check_loops(f'loops_{steps}+{steps}_T16_64x64_clip0_of_a_batch_of_{NC}', inv_p[:1], out_p[:1], inv_ref, out_ref, inv_h, out_h,
                 extra=dict(wall_s_product_batch=t_batch))
    e16_inv, e16_out = rel_l2(inv_h, inv_ref), rel_l2(out_h, out_ref)
    for i in range(1, NC):
        inv_1, out_1 = _product_loops(prod, xs[i], txts[i], negs[i], steps)
        e_inv, e_out = rel_l2(inv_p[i:i + 1], inv_1), rel_l2(out_p[i:i + 1], out_1)
        _record(f'loops_{steps}+{steps}_T16_64x64_clip{i}_of_a_batch_of_{NC}_vs_single_clip_run', inversion_rel_l2=e_inv,
                final_rel_l2=e_out, yardstick_inversion=e16_inv, yardstick_final=e16_out, final_cosine=cosine(out_p[i:i + 1], out_1))
        assert e_inv <= 1.5 * e16_inv + 1e-4 and e_out <= 1.5 * e16_out + 1e-4, (i, e_inv, e_out, e16_inv, e16_out)
        assert e_out <= CAP_FINAL_LATENTS
    assert not torch.equal(out_p[0], out_p[1])


@torch.no_grad()
def heavy_tailed_weights_(model, seed):
    """A second synthetic weight family: the uniform fan-in family of `synth_weights_` gives well-behaved
    activations, real checkpoints do not.  Every matrix / convolution weight is drawn from a Student-t with 3 degrees of freedom
    (variance 3: scaled to the fan-in standard deviation of the uniform family; the tails put single weights 10 - 40 sigma out),
    every normalisation gain log-uniformly in [0.05, 8] per channel, so that fp16 GEMM epilogues, the folded LayerNorm and the
    fp16 residual stream see two orders of magnitude of dynamic range across channels."""
    dev = next(model.parameters()).device
    g = torch.Generator(device=dev).manual_seed(seed)
    for name, p in model.named_parameters():
        if p.dim() >= 2:
            fan_in = p[0].numel()
            # t(3) = normal / sqrt(chi2_3 / 3), from the seeded generator
            z = torch.randn(p.shape, generator=g, device=dev, dtype=torch.float32)
            c = sum(torch.randn(p.shape, generator=g, device=dev, dtype=torch.float32) ** 2 for _ in range(3)) / 3.0
            # (single draws of a t(3) reach 1 000 sigma in 860 M samples: no checkpoint looks like that, and one such weight decides
            # the whole error — clamped at 40 sigma, which keeps the heavy tail: ~ 1e-4 of the weights lie beyond 10 sigma)
            w = (z / c.sqrt()).clamp_(-40.0 * 3.0 ** 0.5, 40.0 * 3.0 ** 0.5) * (1.0 / (3.0 * fan_in)) ** 0.5
            if 'temporal_transformer.proj_out' in name:
                w = w * 0.1
            p.copy_(w.to(p.dtype))
        elif 'norm' in name and name.endswith('weight'):
            lo, hi = torch.log(torch.tensor(0.05)), torch.log(torch.tensor(8.0))
            p.copy_(torch.exp(lo + (hi - lo) * torch.rand(p.shape, generator=g, device=dev)).to(p.dtype))
        else:
            p.copy_((torch.randn(p.shape, generator=g, device=dev) * 0.05).to(p.dtype))
    return model


def test_forward_heavy_tailed_weight_family():
    """(6) one forward at B = 2, T = 4, 64x64 with Student-t(3) weights and per-channel gains over 0.05 ... 8, on its own models
    (the shared fixture keeps the uniform family).  Same rule as every other forward except the absolute cap: this family's
    yardstick is its own."""
    from oracle import unet3d
    from videoswap_amd.unet import AnimateDiffUNet3DModel
    cfg = unet3d.full_config()
    with torch.device('cuda'):
        prod = AnimateDiffUNet3DModel(**cfg)
        ora_dev = unet3d.AnimateDiffUNet3DModel(**cfg).eval()
    prod = heavy_tailed_weights_(prod, seed=4321).half().eval()
    ora_dev.load_state_dict({k: v.float() for k, v in prod.state_dict().items()}, strict=True)
    ora_h = copy.deepcopy(ora_dev).half()
    x, txt = _inputs(2, 4, 64, 64, seed=141)
    ref = _fwd(ora_dev, x, 441, txt)
    out = _fwd(prod, x, 441, txt)
    half = _fwd(ora_h, x, 441, txt)
    gains = torch.cat([p.detach().float().flatten() for n, p in prod.named_parameters() if 'norm' in n and n.endswith('weight')])
    wmax = max(float(p.detach().float().abs().max() / p.detach().float().std()) for n, p in prod.named_parameters() if p.dim() >= 2)
    _check('unet_B2_T4_64x64_heavy_tailed_weights', out, ref, half, cap=None,
           extra=dict(out_absmax=float(out.abs().max()), ref_absmax=float(ref.abs().max()), gain_min=float(gains.min()),
                      gain_max=float(gains.max()), largest_weight_in_sigmas=wmax))
    del prod, ora_dev, ora_h
    torch.cuda.empty_cache()
