"""Deterministic stand-ins for the components OUTSIDE the measured path: CLIP text encoder /
tokenizer and the VAE are third-party models whose weights are not available offline, so benchmarks and tests
feed the denoising loop synthetic tensors of the right shape.  Nothing here is on the hot path.
"""
import hashlib

import torch


class WhitespaceTokenizer:
    """Minimal tokenizer protocol used by the Prompt-to-Prompt word bookkeeping (ptp_utils.py:62-95,
    seq_aligner.py:170-191): whitespace words -> stable ids, BOS/EOS framing, max length 77."""
    model_max_length = 77
    bos_token_id = 49406
    eos_token_id = 49407

    def __init__(self):
        self._vocab = {}
        self._words = {}

    def _id(self, word):
        if word not in self._vocab:
            i = 1000 + int(hashlib.sha1(word.encode()).hexdigest()[:6], 16) % 40000
            while i in self._words and self._words[i] != word:
                i += 1
            self._vocab[word] = i
            self._words[i] = word
        return self._vocab[word]

    def add_tokens(self, names):
        new = [n for n in names if n not in self._vocab]
        for n in new:
            self._id(n)
        return len(new)

    def convert_tokens_to_ids(self, name):
        return self._id(name)

    def encode(self, text):
        return [self.bos_token_id] + [self._id(w) for w in text.split(' ') if w] + [self.eos_token_id]

    def decode(self, ids):
        return ' '.join(self._words.get(i, '') for i in ids if i not in (self.bos_token_id, self.eos_token_id))

    def __call__(self, prompts, padding='max_length', max_length=77, truncation=True, return_tensors='pt'):
        if isinstance(prompts, str):
            prompts = [prompts]
        rows = []
        for p in prompts:
            ids = self.encode(p)[:max_length]
            ids = ids + [self.eos_token_id] * (max_length - len(ids))
            rows.append(ids)
        return type('Encoding', (), {'input_ids': torch.tensor(rows, dtype=torch.long)})()


class SyntheticTextEncoder:
    """ids [n, 77] -> embeddings [n, 77, dim]: a fixed random table indexed by token id plus a position term
    (deterministic, seed-free).  Stands in for CLIPTextModel(...)[0]."""

    def __init__(self, dim=768, dtype=torch.float16, device='cpu'):
        self.dim, self.dtype, self.device = dim, dtype, device

    def to(self, device=None, dtype=None):
        self.device = device or self.device
        self.dtype = dtype or self.dtype
        return self

    def _row(self, token_id):
        g = torch.Generator().manual_seed(int(token_id) * 2654435761 % (2 ** 31))
        return torch.randn(self.dim, generator=g)

    def __call__(self, input_ids):
        n, length = input_ids.shape
        g = torch.Generator().manual_seed(12345)
        pos = torch.randn(length, self.dim, generator=g) * 0.1
        out = torch.stack([torch.stack([self._row(t) for t in row]) + pos for row in input_ids.tolist()])
        return (out.to(device=self.device, dtype=self.dtype),)


def synthetic_clip(seed=0, frames=16, height=64, width=64, text_dim=768, points=8, batch=1, edlora=False,
                   device='cuda', dtype=torch.float16):
    """Seeded synthetic inputs of one clip: latents N(0,1), text N(0,1), point tracks with 10 %
    invisible, point embeddings N(0,1)."""
    g = torch.Generator().manual_seed(seed)
    latents = torch.randn(batch, 4, frames, height, width, generator=g)
    g1 = torch.Generator().manual_seed(seed + 1)
    shape = (1, 16, 77, text_dim) if edlora else (1, 77, text_dim)
    text = torch.randn(*shape, generator=g1)
    negative = torch.randn(1, 77, text_dim, generator=g1)
    g2 = torch.Generator().manual_seed(seed + 2)
    W, H = width * 8, height * 8
    tracks = torch.rand(frames, points, 2, generator=g2) * torch.tensor([float(W), float(H)])
    hidden = torch.rand(frames, points, generator=g2) < 0.1
    tracks[hidden] = -1.0
    emb = torch.randn(points, 1280, generator=g2)
    conditions = dict(pred_tracks=tracks[None], point_embedding=emb[None], img_size=(W, H), index_list=None,
                      point_name2id={f'p{i}': i for i in range(points)})
    return dict(latents=latents.to(device, dtype), text=text.to(device, dtype),
                negative=negative.to(device, dtype), conditions=conditions)


@torch.no_grad()
def synth_weights_(model, seed=1234):
    """Seeded synthetic weights for benchmarking (no pretrained checkpoints exist offline): uniform
    fan-in initialisation of every matrix/conv, the AnimateDiff zero-initialised motion-module proj_out re-drawn
    N(0, 0.02^2) (otherwise the temporal path contributes exactly 0), norm affine parameters randomised.  Runs on
    whatever device the model lives on."""
    dev = next(model.parameters()).device
    g = torch.Generator(device=dev).manual_seed(seed)
    for name, p in model.named_parameters():
        if p.dim() > 1:
            bound = 1.0 / (p[0].numel() ** 0.5)
            p.copy_(((torch.rand(p.shape, generator=g, device=dev) * 2 - 1) * bound).to(p.dtype))
        elif 'norm' in name and name.endswith('weight'):
            p.copy_((torch.rand(p.shape, generator=g, device=dev) + 0.5).to(p.dtype))
        elif 'norm' in name and name.endswith('bias'):
            p.copy_((torch.randn(p.shape, generator=g, device=dev) * 0.1).to(p.dtype))
        else:
            p.copy_(((torch.rand(p.shape, generator=g, device=dev) * 2 - 1) * 0.05).to(p.dtype))
    for name, p in model.named_parameters():
        if 'temporal_transformer.proj_out' in name:
            p.copy_((torch.randn(p.shape, generator=g, device=dev) * 0.02).to(p.dtype))
    return model


def _hash_uniform(n, key, device):
    """n uniform numbers in [0, 1) from a 32-bit integer hash of (index, key): integer arithmetic and one exact
    int -> float conversion only, so the CPU and a GPU produce the SAME bits (torch.Generator streams differ per device)."""
    m = 0xFFFFFFFF
    h = (torch.arange(n, dtype=torch.int64, device=device) * 2654435761 + (int(key) & m)) & m
    for _ in range(2):
        h = h ^ (h >> 16)
        h = (h * 0x45D9F3B) & m
    h = h ^ (h >> 16)
    return (h >> 8).to(torch.float32) * (1.0 / (1 << 24))


@torch.no_grad()
def portable_weights_(model, seed=1234):
    """The same recipe as `synth_weights_` with device-independent numbers: a golden vector recorded on the build
    container's CPU (tests/golden/make_golden_cfg3.py) and the model on the GPU box get bit-identical parameters.
    Normal draws are the sum of four uniforms (Irwin-Hall, variance 1/3), which is all a synthetic weight needs."""
    dev = next(model.parameters()).device

    def uni(p, key):
        return _hash_uniform(p.numel(), key, dev).view(p.shape)

    def nrm(p, key):
        u = uni(p, key) + uni(p, key + 1) + uni(p, key + 2) + uni(p, key + 3)
        return (u - 2.0) * 1.7320508
    for name, p in model.named_parameters():
        key = seed * 1000003 + int(hashlib.sha1(name.encode()).hexdigest()[:7], 16)
        if 'temporal_transformer.proj_out' in name:
            v = nrm(p, key) * 0.02
        elif p.dim() > 1:
            v = (uni(p, key) * 2 - 1) * (1.0 / (p[0].numel() ** 0.5))
        elif 'norm' in name and name.endswith('weight'):
            v = uni(p, key) + 0.5
        elif 'norm' in name and name.endswith('bias'):
            v = nrm(p, key) * 0.1
        else:
            v = (uni(p, key) * 2 - 1) * 0.05
        p.copy_(v.to(p.dtype))
    return model


def portable_randn(shape, key, device='cpu', dtype=torch.float32):
    """Device-independent N(0, 1)-like tensor (Irwin-Hall of the integer hash above) for golden inputs."""
    n = 1
    for d in shape:
        n *= int(d)
    u = sum(_hash_uniform(n, int(key) * 7919 + j, device) for j in range(4))
    return ((u - 2.0) * 1.7320508).view(*shape).to(dtype)
