"""TEST INFRASTRUCTURE — plain-PyTorch fp32 restatement of the CLIP text encoder (transformers `CLIPTextModel`, which
the reference instantiates through StableDiffusionPipeline.from_pretrained and calls at edlora_util.py:144,176 and
pipeline_videoswap.py:491,658).  PINNED: tests/test_clip.py checks it against the `transformers` CLIPTextModel
installed in this image (same weights, same ids -> same last_hidden_state) — the one third-party model of the path
that IS available here.  State-dict keys are the transformers 4.25 ones (`text_model.` prefix) the reference's
checkpoints use.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import oracle/."""
import torch
from torch import nn


class Layer(nn.Module):
    def __init__(self, d, heads, inter, eps):
        super().__init__()
        self.self_attn = nn.Module()
        for n in ('k_proj', 'v_proj', 'q_proj', 'out_proj'):
            setattr(self.self_attn, n, nn.Linear(d, d))
        self.layer_norm1 = nn.LayerNorm(d, eps=eps)
        self.mlp = nn.Module()
        self.mlp.fc1, self.mlp.fc2 = nn.Linear(d, inter), nn.Linear(inter, d)
        self.layer_norm2 = nn.LayerNorm(d, eps=eps)
        self.heads = heads

    def forward(self, x):
        n, t, d = x.shape
        h = self.layer_norm1(x)
        a = self.self_attn

        def split(y):
            return y.view(n, t, self.heads, d // self.heads).transpose(1, 2)
        q, k, v = split(a.q_proj(h)), split(a.k_proj(h)), split(a.v_proj(h))
        scores = q @ k.transpose(-1, -2) * (d // self.heads) ** -0.5
        mask = torch.full((t, t), float('-inf'), device=x.device).triu(1)
        o = (torch.softmax(scores + mask, -1) @ v).transpose(1, 2).reshape(n, t, d)
        x = x + a.out_proj(o)
        h = self.mlp.fc1(self.layer_norm2(x))
        return x + self.mlp.fc2(h * torch.sigmoid(1.702 * h))


class CLIPTextModel(nn.Module):
    def __init__(self, vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12,
                 num_attention_heads=12, max_position_embeddings=77, layer_norm_eps=1e-5, **unused):
        super().__init__()
        tm = self.text_model = nn.Module()
        tm.embeddings = nn.Module()
        tm.embeddings.token_embedding = nn.Embedding(vocab_size, hidden_size)
        tm.embeddings.position_embedding = nn.Embedding(max_position_embeddings, hidden_size)
        tm.encoder = nn.Module()
        tm.encoder.layers = nn.ModuleList([Layer(hidden_size, num_attention_heads, intermediate_size, layer_norm_eps)
                                           for _ in range(num_hidden_layers)])
        tm.final_layer_norm = nn.LayerNorm(hidden_size, eps=layer_norm_eps)

    # the transformers surface convert_edlora_to_diffusers.py:4-33 touches
    def get_input_embeddings(self):
        return self.text_model.embeddings.token_embedding

    def resize_token_embeddings(self, n):
        old = self.text_model.embeddings.token_embedding
        new = nn.Embedding(n, old.embedding_dim, device=old.weight.device, dtype=old.weight.dtype)
        with torch.no_grad():
            new.weight.zero_()
            k = min(n, old.num_embeddings)
            new.weight[:k] = old.weight[:k]
        self.text_model.embeddings.token_embedding = new
        return new

    def forward(self, input_ids):
        tm = self.text_model
        x = tm.embeddings.token_embedding(input_ids) + tm.embeddings.position_embedding.weight[:input_ids.shape[1]]
        for layer in tm.encoder.layers:
            x = layer(x)
        return (tm.final_layer_norm(x),)


def tiny_clip_config(vocab_size=300):
    return dict(vocab_size=vocab_size, hidden_size=64, intermediate_size=128, num_hidden_layers=2,
                num_attention_heads=4, max_position_embeddings=77, hidden_act='quick_gelu', layer_norm_eps=1e-5)


@torch.no_grad()
def synth_weights_(model, seed=55):
    g = torch.Generator().manual_seed(seed)
    for name, p in model.named_parameters():
        if 'layer_norm' in name and name.endswith('weight'):
            p.copy_(torch.rand(p.shape, generator=g) + 0.5)
        elif p.dim() > 1:
            p.copy_(torch.randn(p.shape, generator=g) * (0.02 if 'embedding' in name else p.shape[1] ** -0.5))
        else:
            p.copy_(torch.randn(p.shape, generator=g) * 0.05)
    return model
