"""CLIP text encoder on the libvsx kernels — SURVEY.md §8 f2: the step before the loops (the source prompt for the
inversion, the 16 per-layer ED-LoRA prompts + the negative prompt for the sampling: edlora_util.py:116-196,
pipeline_videoswap.py:491,658).

`CLIPTextModel` keeps the transformers 4.25 module tree and state-dict keys the reference's checkpoints and LoRA
merges address (`text_model.embeddings.token_embedding.weight`, `text_model.encoder.layers.{i}.self_attn.
{q,k,v,out}_proj`, `mlp.fc1/fc2`, `layer_norm1/2`, `text_model.final_layer_norm`: convert_edlora_to_diffusers.py:37-44,
pipeline_videoswap.py:304-305) — the transformers release installed here (5.x) dropped the `text_model.` prefix, so
its own class could not load them.  Weights without the prefix are accepted too.  Forward: embedding lookup (torch
indexing: plumbing) -> 12 x {LayerNorm -> q|k|v GEMM -> causal softmax(q k^T / 8) v -> out GEMM (+residual) ->
LayerNorm -> fc1 GEMM -> quick_gelu -> fc2 GEMM (+residual)} -> final LayerNorm; returns (last_hidden_state,).
The tokenizer is transformers' CLIPTokenizer (host-side BPE, no arithmetic)."""
import json
import os

import torch
from torch import nn

from . import formats, ops
from .compat import ModelMixin
from .layers import LayerNorm, Linear


class CLIPTextConfig:
    def __init__(self, vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12,
                 num_attention_heads=12, max_position_embeddings=77, hidden_act='quick_gelu', layer_norm_eps=1e-5,
                 **unused):
        self.vocab_size, self.hidden_size, self.intermediate_size = vocab_size, hidden_size, intermediate_size
        self.num_hidden_layers, self.num_attention_heads = num_hidden_layers, num_attention_heads
        self.max_position_embeddings, self.hidden_act, self.layer_norm_eps = max_position_embeddings, hidden_act, layer_norm_eps
        if hidden_act != 'quick_gelu':
            raise NotImplementedError(f'CLIP text encoder activation {hidden_act} (SD-1.x uses quick_gelu)')

    def to_dict(self):
        return dict(self.__dict__)


class CLIPAttention(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        d = cfg.hidden_size
        self.heads, self.scale = cfg.num_attention_heads, (d // cfg.num_attention_heads) ** -0.5
        self.k_proj, self.v_proj, self.q_proj, self.out_proj = Linear(d, d), Linear(d, d), Linear(d, d), Linear(d, d)

    def forward(self, x, residual):
        n, t, d = x.shape
        q, k = self.q_proj(x), self.k_proj(x)
        vt = ops.linear_vt(x.reshape(n * t, d), self.v_proj.weight, self.v_proj.bias, t)
        probs = ops.attention_scores(q, k, self.heads, self.scale, causal=True)
        return self.out_proj(ops.attention_pv(probs, vt), residual=residual)


class CLIPMLP(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.fc1, self.fc2 = Linear(cfg.hidden_size, cfg.intermediate_size), Linear(cfg.intermediate_size, cfg.hidden_size)

    def forward(self, x, residual):
        return self.fc2(ops.quick_gelu(self.fc1(x)), residual=residual)


class CLIPEncoderLayer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.self_attn = CLIPAttention(cfg)
        self.layer_norm1 = LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)
        self.mlp = CLIPMLP(cfg)
        self.layer_norm2 = LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)

    def forward(self, x):
        x = self.self_attn(self.layer_norm1(x), residual=x)
        return self.mlp(self.layer_norm2(x), residual=x)


class CLIPTextEmbeddings(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.token_embedding = nn.Embedding(cfg.vocab_size, cfg.hidden_size)
        self.position_embedding = nn.Embedding(cfg.max_position_embeddings, cfg.hidden_size)

    def forward(self, input_ids):
        t = input_ids.shape[1]
        return self.token_embedding(input_ids) + self.position_embedding.weight[:t][None]


class CLIPEncoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.layers = nn.ModuleList([CLIPEncoderLayer(cfg) for _ in range(cfg.num_hidden_layers)])


class CLIPTextTransformer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.embeddings = CLIPTextEmbeddings(cfg)
        self.encoder = CLIPEncoder(cfg)
        self.final_layer_norm = LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)


class CLIPTextModel(ModelMixin):
    def __init__(self, config):
        super().__init__()
        self.config = config if isinstance(config, CLIPTextConfig) else CLIPTextConfig(**dict(config))
        self.text_model = CLIPTextTransformer(self.config)

    @torch.no_grad()
    def forward(self, input_ids, attention_mask=None, **unused):
        if attention_mask is not None:
            raise NotImplementedError('padding masks are not used by the SD text-encoder call (max_length padding)')
        ids = input_ids.to(self.device)
        x = self.text_model.embeddings(ids).to(self.dtype).contiguous()
        for layer in self.text_model.encoder.layers:
            x = layer(x)
        return (self.text_model.final_layer_norm(x),)

    # ---- the small transformers surface convert_edlora_to_diffusers.py:4-33 touches ----
    def get_input_embeddings(self):
        return self.text_model.embeddings.token_embedding

    def resize_token_embeddings(self, new_num_tokens):
        old = self.text_model.embeddings.token_embedding
        if new_num_tokens == old.num_embeddings:
            return old
        new = nn.Embedding(new_num_tokens, old.embedding_dim, device=old.weight.device, dtype=old.weight.dtype)
        with torch.no_grad():
            new.weight.zero_()
            n = min(new_num_tokens, old.num_embeddings)
            new.weight[:n] = old.weight[:n]
        self.text_model.embeddings.token_embedding = new
        self.config.vocab_size = new_num_tokens
        return new

    def load_state_dict(self, state_dict, strict=True):
        sd = {}
        for k, v in state_dict.items():
            if k.endswith('position_ids'):
                continue                                    # buffer of older transformers checkpoints
            sd[k if k.startswith('text_model.') else 'text_model.' + k] = v
        table = sd.get('text_model.embeddings.token_embedding.weight')
        if table is not None and table.shape[0] != self.text_model.embeddings.token_embedding.num_embeddings:
            self.resize_token_embeddings(table.shape[0])
        return super().load_state_dict(sd, strict=strict)

    @classmethod
    def from_pretrained(cls, pretrained_model_path, subfolder=None, torch_dtype=None, **unused):
        path = os.path.join(pretrained_model_path, subfolder) if subfolder else pretrained_model_path
        with open(os.path.join(path, 'config.json')) as f:
            raw = json.load(f)
        raw = raw.get('text_config', raw) if 'hidden_size' not in raw else raw
        model = cls(CLIPTextConfig(**raw))
        for name in ('model.safetensors', 'pytorch_model.bin'):
            file = os.path.join(path, name)
            if os.path.isfile(file):
                if name.endswith('.safetensors'):
                    from safetensors.torch import load_file
                    state = load_file(file)
                else:
                    state = formats.load_checkpoint(file)
                break
        else:
            raise RuntimeError(f'no text-encoder weights under {path}')
        model.load_state_dict(state, strict=True)
        return model.to(dtype=torch_dtype) if torch_dtype is not None else model


def load_tokenizer(pretrained_model_path, subfolder='tokenizer'):
    """transformers' CLIPTokenizer from `<path>/tokenizer` (vocab.json + merges.txt, or a saved tokenizer.json);
    model_max_length 77 as in every SD checkpoint."""
    from transformers import CLIPTokenizer
    path = os.path.join(pretrained_model_path, subfolder) if subfolder else pretrained_model_path
    vocab, merges = os.path.join(path, 'vocab.json'), os.path.join(path, 'merges.txt')
    if os.path.isfile(vocab) and os.path.isfile(merges):
        tok = CLIPTokenizer(vocab, merges, model_max_length=77)
    else:
        tok = CLIPTokenizer.from_pretrained(path)
    if tok.model_max_length is None or tok.model_max_length > 100000:
        tok.model_max_length = 77
    return tok
