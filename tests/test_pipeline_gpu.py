"""GPU parity of the two hot loops (DDIM inversion, CFG sampling with adapter residuals) against the CPU oracle."""
import pytest
import torch

from util import DEV, oracle_unet, product_unet_from, rel_l2

pytestmark = pytest.mark.device


@pytest.fixture(scope='module')
def models():
    from oracle import unet3d
    cfg = unet3d.tiny_config()
    ora = oracle_unet(cfg)
    return cfg, ora, product_unet_from(ora, cfg)


def oracle_loops(ora, x, txt, neg, steps, guidance, residuals=None, t2i_end=1.0):
    from oracle.diffusers_restated import SD15_SCHEDULER_CONFIG, DDIMInverseScheduler, DDIMScheduler
    inv, sch = DDIMInverseScheduler(**SD15_SCHEDULER_CONFIG), DDIMScheduler(**SD15_SCHEDULER_CONFIG)
    inv.set_timesteps(steps)
    sch.set_timesteps(steps)
    lat = x.clone()
    with torch.no_grad():
        for t in inv.timesteps:
            lat = inv.step(ora(lat, t, txt).sample, t, lat).prev_sample
        inverted = lat.clone()
        emb = torch.cat([neg, txt])
        for i, t in enumerate(sch.timesteps):
            res = None
            if residuals is not None and i <= steps * t2i_end:
                res = [torch.cat([r] * 2) for r in residuals]
            e = ora(torch.cat([lat] * 2), t, emb, down_block_additional_residuals=res).sample
            e = e[:1] + guidance * (e[1:] - e[:1])
            lat = sch.step(e, t, lat).prev_sample
    return inverted, lat


def test_inversion_and_guided_sampling(models):
    from videoswap_amd.compat import SD15_SCHEDULER_CONFIG, DDIMScheduler
    from videoswap_amd.pipeline import VideoSwapPipeline
    cfg, ora, prod = models
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 4, 4, 16, 16, generator=g)
    txt, neg = torch.randn(1, 77, 64, generator=g), torch.randn(1, 77, 64, generator=g)
    steps = 4
    inv_ref, out_ref = oracle_loops(ora, x, txt, neg, steps, 7.5)
    pipe = VideoSwapPipeline(unet=prod, scheduler=DDIMScheduler(**SD15_SCHEDULER_CONFIG)).to(DEV)
    inv = pipe.invert(latents=x.half().to(DEV), prompt_embeds=txt.half().to(DEV), num_inference_steps=steps).latents
    out = pipe(prompt_embeds=txt.half().to(DEV), negative_prompt_embeds=neg.half().to(DEV), latents=inv,
               num_inference_steps=steps, guidance_scale=7.5, output_type='latent').videos
    e_inv, e_out = rel_l2(inv.float().cpu(), inv_ref), rel_l2(out.float().cpu(), out_ref)
    print(f'inversion rel-L2 {e_inv:.3e}; inversion+sampling rel-L2 {e_out:.3e}')
    assert e_inv < 5e-3
    assert e_out < 2e-2          # 2*steps sequential fp16 UNet calls compound (SURVEY.md §7 hard parts)


def test_adapter_matches_oracle_and_feeds_the_unet(models):
    """SparsePointAdapter (MLP + bilinear scatter kernel) against the reference algorithm restated in the oracle."""
    from oracle import adapter as oadapter
    from videoswap_amd.adapter import SparsePointAdapter
    from videoswap_amd.synthetic import synthetic_clip
    cfg, ora, prod = models
    chans = list(cfg['block_out_channels'])
    o = oadapter.SparsePointAdapter(embedding_channels=1280, channels=chans).eval()
    p = SparsePointAdapter(embedding_channels=1280, channels=chans).eval()
    p.load_state_dict(o.state_dict(), strict=True)
    p = p.to(DEV, torch.float16)
    data = synthetic_clip(seed=9, frames=3, height=16, width=24, text_dim=64, points=6, device=DEV)
    cond = data['conditions']
    tracks16 = cond['pred_tracks'].half().float()     # the reference holds the tracks in fp16
    with torch.no_grad():
        ref = o(tracks16, cond['img_size'], cond['point_embedding'], index_list=[0, 1, 2, 4])
    got = p(cond['pred_tracks'], cond['img_size'], cond['point_embedding'].half().to(DEV), index_list=[0, 1, 2, 4])
    for level, (r, gt) in enumerate(zip(ref, SparsePointAdapter.to_reference_layout(got))):
        assert gt.shape == r.shape
        assert rel_l2(gt.float().cpu(), r) < 5e-3, level


def test_foreign_processor_protocol(models):
    """A processor written against the diffusers protocol (here: the oracle's restated AttnProcessor, which calls
    to_q/to_k/to_v, head_to_batch_dim, get_attention_scores, torch.bmm, to_out) must run on the HIP-backed Attention
    and give the same UNet output as the native fused processor."""
    from oracle.diffusers_restated import AttnProcessor as ForeignProcessor
    cfg, ora, prod = models
    g = torch.Generator().manual_seed(11)
    x = torch.randn(1, 4, 2, 16, 16, generator=g).half().to(DEV)
    txt = torch.randn(1, 77, 64, generator=g).half().to(DEV)
    with torch.no_grad():
        base = prod(x, 301, txt).sample
        saved = {}
        for name, m in prod.named_modules():
            if m.__class__.__name__ == 'Attention' and ('attn1' in name or 'attn2' in name):
                saved[name] = m.processor
                m.set_processor(ForeignProcessor())
        try:
            foreign = prod(x, 301, txt).sample
        finally:
            for name, m in prod.named_modules():
                if name in saved:
                    m.set_processor(saved[name])
    assert rel_l2(foreign.float(), base.float()) < 3e-3


def test_timestep_work_of_a_loop_is_prepared_in_one_pass(models, monkeypatch):
    """`AnimateDiffUNet3DModel.prepare_timesteps` (round 6): the sinusoid, the two time-embedding Linears and the 22 per-resnet
    `time_emb_proj` rows (unet.py:391-397, resnet.py:172-176) of ALL timesteps of a loop in three launches, handed out through
    the caches the per-step path uses.  The loop must then find every row prepared (no M = 1 projection inside a step), and its
    result must agree with the per-step path (VSX_PREPARE_TIMESTEPS=0) to the rounding of one GEMM in another tile shape."""
    from videoswap_amd import ops
    from videoswap_amd.compat import SD15_SCHEDULER_CONFIG, DDIMScheduler
    from videoswap_amd.pipeline import VideoSwapPipeline
    from videoswap_amd.unet import ResnetBlock3D
    cfg, ora, prod = models
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 4, 4, 16, 16, generator=g).half().to(DEV)
    txt = torch.randn(1, 77, 64, generator=g).half().to(DEV)
    pipe = VideoSwapPipeline(unet=prod, scheduler=DDIMScheduler(**SD15_SCHEDULER_CONFIG)).to(DEV)
    steps = 5
    monkeypatch.setenv('VSX_PREPARE_TIMESTEPS', '0')
    prod.clear_step_caches()
    per_step = pipe.invert(latents=x, prompt_embeds=txt, num_inference_steps=steps).latents.float().cpu()
    monkeypatch.setenv('VSX_PREPARE_TIMESTEPS', '1')
    prod.clear_step_caches()
    calls = []
    real = ops.linear

    def counting(xin, *a, **k):
        calls.append(tuple(xin.shape))
        return real(xin, *a, **k)
    monkeypatch.setattr(ops, 'linear', counting)
    prepared = pipe.invert(latents=x, prompt_embeds=txt, num_inference_steps=steps).latents.float().cpu()
    monkeypatch.setattr(ops, 'linear', real)
    resnets = [m for m in prod.modules() if isinstance(m, ResnetBlock3D)]
    assert len(prod._semb_cache) == steps and all(len(r.__dict__['_tproj'].entries) == steps for r in resnets)
    # the only launches with one row per TIMESTEP are the three of the preparation; no single-row GEMM inside the steps
    assert sum(1 for s in calls if s[0] == steps and len(s) == 2) == 3, [s for s in calls if len(s) == 2 and s[0] <= steps]
    assert not any(len(s) == 2 and s[0] == 1 for s in calls), 'a time-embedding row was projected inside a step'
    assert rel_l2(prepared, per_step) < 2e-3
    prod.clear_step_caches()
