"""GPU parity of the full VideoSwap swap path (BASELINE.json configs[2] in miniature): DDIM inversion with the
attention store -> ED-LoRA merge + per-layer prompt embeddings -> point-adapter residuals (t2i window) -> AttentionRefine
with latent + self-attention SpatialBlenders -> weights restored; product `VideoSwapPipeline.validation` on the GPU
against the same flow assembled from the CPU oracle (oracle UNet/adapter/loops; controllers are the shared host logic
pinned against the reference in tests/test_control.py)."""
import copy

import pytest
import torch

from util import DEV, oracle_unet, product_unet_from, rel_l2

pytestmark = pytest.mark.device

SOURCE = 'a silver jeep driving down a curvy road in the countryside'
STEPS, FRAMES = 4, 2
HW = 32     # (the real 64x64 / SD-1.5-width case, with the reference's own controllers on the oracle side, is
            #  tests/test_cfg3_fullwidth_gpu.py; this one checks the orchestration of `validation` at tiny width)


def synthetic_lora(state_dict, seed=4, rank=4):
    g = torch.Generator().manual_seed(seed)
    lora = {}
    for k, w in state_dict.items():
        hit = any(k.endswith(s) for s in ('to_q.weight', 'to_k.weight', 'to_v.weight', 'to_out.0.weight',
                                          'ff.net.0.proj.weight', 'ff.net.2.weight', 'proj_in.weight',
                                          'proj_out.weight'))
        if hit and 'motion_modules' not in k and 'attentions' in k:
            base = k[:-len('weight')]
            down = torch.randn(rank, w.shape[1], generator=g) * 0.02
            up = torch.randn(w.shape[0], rank, generator=g) * 0.02
            if w.dim() == 4:
                down, up = down[:, :, None, None], up[:, :, None, None]
            lora[base + 'lora_down.weight'], lora[base + 'lora_up.weight'] = down, up
    emb = {'<porsche1>': torch.zeros(16, 64), '<porsche2>': torch.zeros(16, 64)}
    return {'params': {'new_concept_embedding': emb, 'unet': lora}}


class CpuPipeShim:
    """What encode_edlora_prompt / convert_edlora need from a pipeline object, on the CPU oracle side."""

    def __init__(self, unet, tokenizer, text_encoder):
        self.unet, self.tokenizer, self.text_encoder = unet, tokenizer, text_encoder
        self.new_concept_cfg = None


def test_full_swap_flow_matches_oracle():
    from oracle import adapter as oadapter
    from oracle import pipeline as opipe
    from oracle import unet3d
    from videoswap_amd import control
    from videoswap_amd.adapter import SparsePointAdapter
    from videoswap_amd.compat import SD15_SCHEDULER_CONFIG, DDIMScheduler
    from videoswap_amd.edlora import convert_edlora, encode_edlora_prompt
    from videoswap_amd.pipeline import VideoSwapPipeline
    from videoswap_amd.synthetic import SyntheticTextEncoder, WhitespaceTokenizer, synthetic_clip

    cfg = unet3d.tiny_config()
    ora = oracle_unet(cfg)
    ora.dtype_ = torch.float32
    prod = product_unet_from(ora, cfg)
    chans = list(cfg['block_out_channels'])
    oad = oadapter.SparsePointAdapter(1280, chans).eval()
    pad = SparsePointAdapter(embedding_channels=1280, channels=chans).eval()
    pad.load_state_dict(oad.state_dict(), strict=True)
    pad = pad.to(DEV, torch.float16)

    data = synthetic_clip(seed=21, frames=FRAMES, height=HW, width=HW, text_dim=64, points=5, device='cpu',
                          dtype=torch.float32)
    latents = data['latents']                     # [1,4,F,h,w]
    conditions = data['conditions']
    conditions['pred_tracks'] = conditions['pred_tracks'].half().float()   # the reference holds tracks in fp16
    lora = synthetic_lora(ora.state_dict())
    editing_config = dict(use_invertion_latents=True, use_blend=True, num_inference_steps=STEPS, guidance_scale=7.5,
                          t2i_guidance_scale=0.5, t2i_start=0.0, t2i_end=0.5,
                          editing_prompts={'0': dict(replace='silver jeep -> <porsche1> <porsche2>',
                                                     lora_path='synthetic_edlora.pth---0.7',
                                                     blend_cfg=dict(cross_replace_steps=0.5, self_replace_steps=0.5,
                                                                    blend_th=0.3))})

    # ---------------- product on the GPU ----------------
    tok = WhitespaceTokenizer()
    pipe = VideoSwapPipeline(unet=prod, adapter=pad, tokenizer=tok, scheduler=DDIMScheduler(**SD15_SCHEDULER_CONFIG),
                             text_encoder=SyntheticTextEncoder(dim=64, dtype=torch.float16, device=DEV)).to(DEV)
    before = copy.deepcopy(prod.state_dict())
    video = latents[0].permute(1, 0, 2, 3).contiguous().half().to(DEV)          # [F,4,h,w] "video" of latents
    edited = pipe.validation(video, conditions, SOURCE, editing_config, lora_loader=lambda path: lora)
    got = edited['0'].float().cpu()
    # weights restored bit-exactly, processors reset to the ED-LoRA-free state is not required by the reference
    after = prod.state_dict()
    assert all(torch.equal(before[k], after[k]) for k in before)

    # ---------------- oracle on the CPU ----------------
    otok = WhitespaceTokenizer()
    oenc = SyntheticTextEncoder(dim=64, dtype=torch.float32, device='cpu')
    store = control.AttentionStore()
    store.LOW_RESOURCE = True
    opipe.register_control(ora, store)
    src_emb = oenc(otok(SOURCE).input_ids)[0]
    inv = opipe.invert(ora, latents, src_emb, STEPS, controller=store)
    store.LOW_RESOURCE = False
    shim = CpuPipeShim(ora, otok, oenc)
    snapshot = copy.deepcopy(ora.state_dict())
    _, concept_cfg = convert_edlora(shim, lora, enable_edlora=True, alpha=0.7)
    otok.new_concept_cfg = concept_cfg
    target = SOURCE.replace('silver jeep', '<porsche1> <porsche2>')
    edit = control.make_controller(otok, [SOURCE, target], False, cross_replace_steps=0.5, self_replace_steps=0.5,
                                   blend_words=[['silver', 'jeep'], ['<porsche1>', '<porsche2>']],
                                   additional_attention_store=store, blend_th=(0.3, 0.3), NUM_DDIM_STEPS=STEPS,
                                   blend_latents=True, blend_self_attention=True, image_height=HW * 8,
                                   image_width=HW * 8)
    opipe.register_control(ora, edit, edlora=True)
    emb = encode_edlora_prompt(shim, target, concept_cfg, 'cpu', 1, True, None)     # [2,16,77,64]: [uncond; cond]
    state = oad(conditions['pred_tracks'], conditions['img_size'], conditions['point_embedding'])
    state = [s * 0.5 for s in state]
    ref = opipe.sample(ora, inv, emb[1:], emb[:1], STEPS, guidance=7.5, controller=edit, adapter_state=state,
                       t2i_start=0.0, t2i_end=0.5)
    ora.load_state_dict(snapshot)

    err = rel_l2(got, ref)
    print(f'full swap flow ({STEPS}+{STEPS} steps, ED-LoRA + adapter + P2P blend): rel-L2 {err:.3e}')
    assert got.shape == ref.shape and torch.isfinite(got).all()
    assert err < 5e-2, err
