"""TEST INFRASTRUCTURE — CPU restatement of VideoSwapPipeline's two loops and the processors that ride on them.

Follows pipeline_videoswap.py:552-601 (guided sampling), :677-710 (inversion), attention_register.py:15-173 (control
processors, materialised probabilities through the controller for layers with < 32^2 tokens) and
edlora_util.py:13-82 (per-layer text embedding).  The Prompt-to-Prompt controllers themselves are host logic shared
with the product (videoswap_amd/control.py, device-agnostic torch code pinned against the reference's p2p modules in
tests/test_control.py); here they are only CALLED with CPU tensors.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import anything under oracle/."""
import torch

from .diffusers_restated import SD15_SCHEDULER_CONFIG, DDIMInverseScheduler, DDIMScheduler


class ControlProcessor:
    """attention_register.py:96-173 / :15-93 restated on the oracle Attention: layers with < 32^2 query tokens
    materialise their probabilities and hand them to the controller, larger ones take the fused branch."""

    def __init__(self, place_in_unet, controller, cross_attention_idx=None):
        self.place_in_unet, self.controller, self.cross_attention_idx = place_in_unet, controller, cross_attention_idx

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None):
        is_cross = encoder_hidden_states is not None
        ctx = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        if ctx.dim() == 4:
            ctx = ctx[:, self.cross_attention_idx, ...]
        q = attn.head_to_batch_dim(attn.to_q(hidden_states))
        k = attn.head_to_batch_dim(attn.to_k(ctx))
        v = attn.head_to_batch_dim(attn.to_v(ctx))
        if q.shape[-2] >= 32 ** 2:    # the xformers branch of the reference: fused, controller not called
            out = torch.nn.functional.scaled_dot_product_attention(q, k, v)
        else:
            probs = attn.get_attention_scores(q, k, None)
            bh, s, t = probs.shape
            probs = probs.reshape(bh // attn.heads, attn.heads, s, t)
            probs = self.controller(probs, is_cross, self.place_in_unet)
            out = torch.bmm(probs.reshape(bh, s, t), v)
        out = attn.batch_to_head_dim(out)
        return attn.to_out[1](attn.to_out[0](out))


class EDLoRAProcessor:
    """edlora_util.py:13-82"""

    def __init__(self, cross_attention_idx):
        self.cross_attention_idx = cross_attention_idx

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None):
        ctx = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        if ctx.dim() == 4:
            ctx = ctx[:, self.cross_attention_idx, ...]
        q = attn.head_to_batch_dim(attn.to_q(hidden_states))
        k = attn.head_to_batch_dim(attn.to_k(ctx))
        v = attn.head_to_batch_dim(attn.to_v(ctx))
        out = attn.batch_to_head_dim(torch.bmm(attn.get_attention_scores(q, k, None), v))
        return attn.to_out[1](attn.to_out[0](out))


def _walk(unet, fn):
    """Visit attn1/attn2 in the reference's traversal order (down, mid, up; edlora_util.py:85-99)."""
    counts = [0, 0]

    def visit(module, place):
        for name, layer in module.named_children():
            if layer.__class__.__name__ == 'Attention' and ('attn1' in name or 'attn2' in name):
                fn(layer, name, place, counts)
                counts[0 if 'attn1' in name else 1] += 1
            else:
                visit(layer, place)
    visit(unet.down_blocks, 'down')
    visit(unet.mid_block, 'mid')
    visit(unet.up_blocks, 'up')
    return counts


def use_edlora(unet):
    def fn(layer, name, place, counts):
        if 'attn2' in name:
            layer.set_processor(EDLoRAProcessor(counts[1]))
    _walk(unet, fn)


def register_control(unet, controller, edlora=False):
    def fn(layer, name, place, counts):
        layer.set_processor(ControlProcessor(place, controller, counts[1] if edlora else None))
    counts = _walk(unet, fn)
    controller.num_att_layers = counts[0] + counts[1]


def reset_processors(unet, edlora=False):
    from .diffusers_restated import AttnProcessor2_0

    def fn(layer, name, place, counts):
        layer.set_processor(EDLoRAProcessor(counts[1]) if (edlora and 'attn2' in name) else AttnProcessor2_0())
    _walk(unet, fn)


@torch.no_grad()
def invert(unet, latents, text, steps, controller=None):
    """pipeline_videoswap.py:677-710 (guidance 1: no CFG)"""
    sch = DDIMInverseScheduler(**SD15_SCHEDULER_CONFIG)
    sch.set_timesteps(steps)
    for t in sch.timesteps:
        eps = unet(latents, t, text).sample
        latents = sch.step(eps, t, latents).prev_sample
        if controller is not None:
            latents = controller.step_callback(latents)
    return latents


@torch.no_grad()
def sample(unet, latents, text, negative, steps, guidance=7.5, controller=None, adapter_state=None, t2i_start=0.0,
           t2i_end=1.0):
    """pipeline_videoswap.py:552-601; `text`/`negative` [1,77,D] or [1,16,77,D] (negative is repeated over layers)."""
    sch = DDIMScheduler(**SD15_SCHEDULER_CONFIG)
    sch.set_timesteps(steps)
    if text.dim() == 4 and negative.dim() == 3:
        negative = negative[:, None].repeat(1, text.shape[1], 1, 1)
    emb = torch.cat([negative, text])
    n = len(sch.timesteps)
    for i, t in enumerate(sch.timesteps):
        res = None
        if adapter_state is not None and n * t2i_start <= i <= n * t2i_end:
            res = [torch.cat([s] * 2).clone() for s in adapter_state]
        eps = unet(torch.cat([latents] * 2), t, emb, down_block_additional_residuals=res).sample
        eps = eps[:1] + guidance * (eps[1:] - eps[:1])
        latents = sch.step(eps, t, latents).prev_sample
        if controller is not None:
            latents = controller.step_callback(latents)
    return latents
