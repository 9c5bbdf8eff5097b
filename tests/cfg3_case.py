"""BASELINE.json configs[2] at the SD-1.5 width, shared by the golden generator (tests/golden/make_golden_cfg3.py, build
container: oracle UNet on the CPU + the REFERENCE's own Prompt-to-Prompt controllers imported verbatim) and the GPU test
(tests/test_cfg3_fullwidth_gpu.py: the product's `VideoSwapPipeline.validation` on the HIP kernels).

The case: T = 4 frames, 64x64 latents, 4 inversion steps with the AttentionStore (`use_blend: true`) + 4 CFG-7.5 sampling
steps with ED-LoRA per-layer embeddings [2,16,77,768], merged rank-4 LoRA weights, point-adapter residuals inside the t2i
window, AttentionRefine + latent / self-attention SpatialBlenders (pipeline_videoswap.py:272-423).  Every number both
sides need is device-independent: weights and latents come from the integer hash of videoswap_amd.synthetic
(`portable_weights_`, `portable_randn`), everything else from CPU generators.

TEST INFRASTRUCTURE: the oracle flow below is the checker's side only."""
import copy

import torch

SOURCE = 'a silver jeep driving down a curvy road in the countryside'
REPLACE = 'silver jeep -> <porsche1> <porsche2>'
FRAMES, HW, STEPS, POINTS = 4, 64, 4, 8
SEED_W, SEED_A, SEED_X = 1234, 77, 9001
LORA_ALPHA = 0.7
BLEND = dict(cross_replace_steps=0.5, self_replace_steps=0.5, blend_th=0.3)
T2I = dict(t2i_guidance_scale=0.5, t2i_start=0.0, t2i_end=0.5)


class Size:
    """One size of the case.  SMALL is the round-3 golden (T = 4, 4 + 4 steps: every controller branch fires on one or two
    steps); BENCH is the size bench.py --config 3 times in frames (T = 16) with 10 + 10 steps and the replace fractions of
    the reference's option files (`cross_replace_steps` / `self_replace_steps` 0.3: three steps with and seven without the
    replacement, attention_util.py:28-138; the adapter window [0, 0.5] covers five steps, pipeline_videoswap.py:560-567;
    both blenders run on every step, spatial_blend.py:25-145)."""

    def __init__(self, name, frames, steps, blend, seed_x):
        self.name, self.frames, self.steps, self.blend, self.seed_x = name, frames, steps, dict(blend), seed_x

    @property
    def golden(self):
        return 'cfg3_fullwidth.pt' if self.name == 'small' else f'cfg3_fullwidth_{self.name}.pt'


SMALL = Size('small', FRAMES, STEPS, BLEND, SEED_X)
BENCH = Size('T16_10+10', 16, 10, dict(cross_replace_steps=0.3, self_replace_steps=0.3, blend_th=0.3), 9016)


def synthetic_lora(state_dict, seed=4, rank=4, text_dim=768):
    """rank-4 factors on the keys convert_edlora_to_diffusers.py:46-53 merges (spatial transformers only).  Every key
    draws from its own generator (seeded by a hash of its name): the result does not depend on the order in which a
    model lists its parameters, so the oracle and the product get the same LoRA from their own state dicts."""
    import hashlib
    lora = {}
    for k in sorted(state_dict):
        w = state_dict[k]
        hit = any(k.endswith(s) for s in ('to_q.weight', 'to_k.weight', 'to_v.weight', 'to_out.0.weight',
                                          'ff.net.0.proj.weight', 'ff.net.2.weight', 'proj_in.weight',
                                          'proj_out.weight'))
        if hit and 'motion_modules' not in k and 'attentions' in k:
            g = torch.Generator().manual_seed(seed * 1000003 + int(hashlib.sha1(k.encode()).hexdigest()[:7], 16))
            base = k[:-len('weight')]
            down = torch.randn(rank, w.shape[1], generator=g) * 0.02
            up = torch.randn(w.shape[0], rank, generator=g) * 0.02
            if w.dim() == 4:
                down, up = down[:, :, None, None], up[:, :, None, None]
            lora[base + 'lora_down.weight'], lora[base + 'lora_up.weight'] = down, up
    emb = {'<porsche1>': torch.zeros(16, text_dim), '<porsche2>': torch.zeros(16, text_dim)}
    return {'params': {'new_concept_embedding': emb, 'unet': lora}}


def inputs(size=SMALL):
    from videoswap_amd.synthetic import portable_randn, synthetic_clip
    data = synthetic_clip(seed=21, frames=size.frames, height=HW, width=HW, text_dim=768, points=POINTS, device='cpu',
                          dtype=torch.float32)
    conditions = data['conditions']
    conditions['pred_tracks'] = conditions['pred_tracks'].half().float()     # the reference holds tracks in fp16
    # fp16-exact: both sides start from the same bits
    latents = portable_randn((1, 4, size.frames, HW, HW), size.seed_x).half().float()
    return latents, conditions


def editing_config(steps=None, size=SMALL):
    steps = size.steps if steps is None else steps
    return dict(use_invertion_latents=True, use_blend=True, num_inference_steps=steps, guidance_scale=7.5, **T2I,
                editing_prompts={'0': dict(replace=REPLACE, lora_path=f'synthetic_edlora.pth---{LORA_ALPHA}',
                                           blend_cfg=dict(size.blend))})


class PipeShim:
    """What encode_edlora_prompt / convert_edlora need from a pipeline object, on the oracle side."""

    def __init__(self, unet, tokenizer, text_encoder):
        self.unet, self.tokenizer, self.text_encoder = unet, tokenizer, text_encoder
        self.new_concept_cfg = None


@torch.no_grad()
def oracle_flow(ora, oad, store_cls, make_controller, steps=None, log=None, size=SMALL):
    """The swap flow assembled from the oracle UNet / adapter / loops on whatever device and dtype `ora` lives on, with
    the controller classes handed in (the reference's own, or videoswap_amd.control which is pinned against them).
    Returns (inverted latents, final latents), fp32 on the CPU."""
    from oracle import pipeline as opipe
    from videoswap_amd.edlora import convert_edlora, encode_edlora_prompt
    from videoswap_amd.synthetic import SyntheticTextEncoder, WhitespaceTokenizer
    p0 = next(ora.parameters())
    dev, dt = p0.device, p0.dtype
    steps = size.steps if steps is None else steps
    blend = size.blend
    latents, conditions = inputs(size)
    tok = WhitespaceTokenizer()
    enc = SyntheticTextEncoder(dim=768, dtype=dt, device=dev)
    store = store_cls()
    store.LOW_RESOURCE = True
    opipe.register_control(ora, store)
    src_emb = enc(tok(SOURCE).input_ids)[0]
    inv = opipe.invert(ora, latents.to(dev, dt), src_emb, steps, controller=store)
    if log:
        log('inversion done')
    store.LOW_RESOURCE = False
    shim = PipeShim(ora, tok, enc)
    snapshot = copy.deepcopy(ora.state_dict())
    try:
        _, concept_cfg = convert_edlora(shim, synthetic_lora(snapshot), enable_edlora=True, alpha=LORA_ALPHA)
        tok.new_concept_cfg = concept_cfg
        src_subject, tgt_subject = [s.strip() for s in REPLACE.split('->')]
        target = SOURCE.replace(src_subject, tgt_subject)
        edit = make_controller(tok, [SOURCE, target], False, cross_replace_steps=blend['cross_replace_steps'],
                               self_replace_steps=blend['self_replace_steps'],
                               blend_words=[src_subject.split(' '), tgt_subject.split(' ')],
                               additional_attention_store=store, blend_th=(blend['blend_th'], blend['blend_th']),
                               NUM_DDIM_STEPS=steps, blend_latents=True, blend_self_attention=True,
                               image_height=HW * 8, image_width=HW * 8)
        opipe.register_control(ora, edit, edlora=True)
        emb = encode_edlora_prompt(shim, target, concept_cfg, dev, 1, True, None)      # [2,16,77,768]: [uncond; cond]
        # the oracle adapter is a CPU loop (oracle/adapter.py); its maps then move to the UNet's device / dtype
        state = oad(conditions['pred_tracks'], conditions['img_size'], conditions['point_embedding'])
        state = [(s * T2I['t2i_guidance_scale']).to(dev, dt) for s in state]
        out = opipe.sample(ora, inv, emb[1:], emb[:1], steps, guidance=7.5, controller=edit, adapter_state=state,
                           t2i_start=T2I['t2i_start'], t2i_end=T2I['t2i_end'])
    finally:
        ora.load_state_dict(snapshot)
        opipe.reset_processors(ora)
    return inv.float().cpu(), out.float().cpu()
