"""Shared helpers of the processor / ED-LoRA golden vectors: seeded toy inputs and a deterministic controller that
implements the controller protocol (`controller(probs[b,h,s,t], is_cross, place) -> probs`, attention_register.py:70-76)
without any Prompt-to-Prompt logic — the P2P controllers themselves are pinned in tests/test_control.py."""
import torch


class ToyController:
    """probs -> a * probs + (1 - a) * roll(probs, 1 key); `a` depends on (is_cross, place); logs every call."""
    LOW_RESOURCE = False
    num_att_layers = 0

    def __init__(self):
        self.calls = []

    def __call__(self, probs, is_cross, place_in_unet):
        self.calls.append((bool(is_cross), place_in_unet, tuple(probs.shape)))
        a = {'down': 0.75, 'mid': 0.5, 'up': 0.25}[place_in_unet] + (0.125 if is_cross else 0.0)
        return a * probs + (1.0 - a) * torch.roll(probs, 1, dims=-1)

    def step_callback(self, x):
        return x


def attention_inputs(seed=7, nb=2, tokens=64, dim=320, text_dim=768):
    g = torch.Generator().manual_seed(seed)
    return dict(hidden=torch.randn(nb, tokens, dim, generator=g),
                text=torch.randn(nb, 77, text_dim, generator=g),
                text_layers=torch.randn(nb, 16, 77, text_dim, generator=g))


def attention_weights(seed=8, dim=320, text_dim=768):
    """state dicts of a self- and a cross-attention `Attention(query_dim=dim, heads=8, dim_head=dim//8)`"""
    g = torch.Generator().manual_seed(seed)

    def lin(o, i):
        return torch.randn(o, i, generator=g) * i ** -0.5
    self_sd = {'to_q.weight': lin(dim, dim), 'to_k.weight': lin(dim, dim), 'to_v.weight': lin(dim, dim),
               'to_out.0.weight': lin(dim, dim), 'to_out.0.bias': torch.randn(dim, generator=g) * 0.1}
    cross_sd = {'to_q.weight': lin(dim, dim), 'to_k.weight': lin(dim, text_dim), 'to_v.weight': lin(dim, text_dim),
                'to_out.0.weight': lin(dim, dim), 'to_out.0.bias': torch.randn(dim, generator=g) * 0.1}
    return self_sd, cross_sd


def lora_case(seed=11):
    """A small UNet-like and text-encoder-like state dict plus an ED-LoRA checkpoint in the on-disk layout
    (convert_edlora_to_diffusers.py:82-105: {'params': {'new_concept_embedding', 'unet', 'text_encoder'}})."""
    g = torch.Generator().manual_seed(seed)

    def t(*s):
        return torch.randn(*s, generator=g)
    unet = {
        'down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q.weight': t(32, 32),
        'down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_k.weight': t(32, 32),
        'down_blocks.0.attentions.0.transformer_blocks.0.attn2.to_v.weight': t(32, 48),
        'down_blocks.0.attentions.0.transformer_blocks.0.attn2.to_out.0.weight': t(32, 32),
        'down_blocks.0.attentions.0.transformer_blocks.0.attn2.to_out.0.bias': t(32),
        'down_blocks.0.attentions.0.transformer_blocks.0.ff.net.0.proj.weight': t(256, 32),
        'down_blocks.0.attentions.0.transformer_blocks.0.ff.net.2.weight': t(32, 128),
        'down_blocks.0.attentions.0.proj_in.weight': t(32, 32, 1, 1),
        'down_blocks.0.attentions.0.proj_out.weight': t(32, 32, 1, 1),
        'down_blocks.0.resnets.0.conv1.weight': t(32, 32, 3, 3),          # no LoRA on convs
        'down_blocks.0.motion_modules.0.temporal_transformer.proj_out.weight': t(32, 32),   # Linear named proj_out
    }
    text = {
        'text_model.encoder.layers.0.self_attn.q_proj.weight': t(24, 24),
        'text_model.encoder.layers.0.self_attn.out_proj.weight': t(24, 24),
        'text_model.encoder.layers.0.mlp.fc1.weight': t(96, 24),
        'text_model.encoder.layers.0.mlp.fc2.weight': t(24, 96),
        'text_model.embeddings.token_embedding.weight': t(50, 24),
    }

    def factors(sd, names, rank=4):
        out = {}
        for k in names:
            w = sd[k]
            o, i = w.shape[0], w.shape[1]
            down, up = t(rank, i) * 0.1, t(o, rank) * 0.1
            if w.dim() == 4:
                down, up = down[..., None, None], up[..., None, None]
            out[k.replace('.weight', '.lora_down.weight')] = down
            out[k.replace('.weight', '.lora_up.weight')] = up
        return out
    unet_lora = factors(unet, [k for k in unet if k.endswith('weight') and 'conv1' not in k])
    text_lora = factors(text, [k for k in text if 'token_embedding' not in k])
    ckpt = {'params': {'new_concept_embedding': {'<catA1>': t(16, 24), '<catA2>': t(16, 24)},
                       'unet': unet_lora, 'text_encoder': text_lora}}
    return unet, text, ckpt
