"""AutoencoderKL (the SD-1.5 VAE) on the libvsx kernels — SURVEY.md §8 f1: the steps immediately either side of the
denoising loops (`vae.encode(frames).latent_dist.sample() * scaling_factor` before the inversion,
`vae.decode(latents / scaling_factor)` after the sampling: pipeline_videoswap.py:204-233, 603-610).

Same module tree / state-dict keys as diffusers 0.19.3 `AutoencoderKL` (encoder / decoder / quant_conv /
post_quant_conv; `mid_block.attentions.0.{group_norm,to_q,to_k,to_v,to_out.0}`; checkpoints that still use the
pre-0.18 attention names `query/key/value/proj_attn` are renamed on load), so `vae/diffusion_pytorch_model.bin` of an
SD-1.5 checkpoint loads unchanged.  Activations are channels-last fp16 [N, H, W, C]; every conv is the implicit-GEMM
kernel (the encoder's `F.pad(x, (0, 1, 0, 1))` + stride-2 conv is its asymmetric-padding mode), GroupNorm+SiLU the
fused norm kernels, the single-head d = 512 mid-block attention runs as GEMM + row softmax + GEMM.
diffusers is not installed here and the reference has no VAE code of its own: the arithmetic follows the published
0.19.3 source (restated in oracle/vae.py) — parity unpinned, like the other diffusers pieces.
"""
import json
import os
from dataclasses import dataclass

import torch
from torch import nn

from . import formats, ops
from .compat import BaseOutput, ConfigMixin, ModelMixin, register_to_config
from .layers import GroupNorm, InflatedConv3d as Conv2d, Linear

SD15_VAE_CONFIG = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512),
                       layers_per_block=2, norm_num_groups=32, sample_size=512, scaling_factor=0.18215,
                       act_fn='silu', down_block_types=('DownEncoderBlock2D',) * 4,
                       up_block_types=('UpDecoderBlock2D',) * 4)


class ResnetBlock2D(nn.Module):
    """diffusers ResnetBlock2D(temb_channels=None, eps=1e-6): GN -> SiLU -> conv -> GN -> SiLU -> conv (+ 1x1 shortcut)."""

    def __init__(self, in_channels, out_channels, groups=32, eps=1e-6):
        super().__init__()
        self.norm1 = GroupNorm(groups, in_channels, eps=eps)
        self.conv1 = Conv2d(in_channels, out_channels, 3, padding=1)
        self.norm2 = GroupNorm(groups, out_channels, eps=eps)
        self.conv2 = Conv2d(out_channels, out_channels, 3, padding=1)
        self.conv_shortcut = Conv2d(in_channels, out_channels, 1, padding=0) if in_channels != out_channels else None

    def forward(self, x):
        n = x.shape[0]
        h = self.conv1(self.norm1(x, n, silu=True))
        h = self.norm2(h, n, silu=True)
        shortcut = x if self.conv_shortcut is None else self.conv_shortcut(x)
        return self.conv2(h, residual=shortcut)


class VaeAttention(nn.Module):
    """diffusers Attention(channels, heads=1, dim_head=channels, bias=True, residual_connection=True,
    norm_num_groups=32, eps=1e-6) as the VAE mid block builds it: GroupNorm -> q, k, v -> softmax(q k^T / sqrt(C)) v
    -> to_out -> + input."""

    def __init__(self, channels, groups=32, eps=1e-6):
        super().__init__()
        self.group_norm = GroupNorm(groups, channels, eps=eps)
        self.to_q = Linear(channels, channels)
        self.to_k = Linear(channels, channels)
        self.to_v = Linear(channels, channels)
        self.to_out = nn.ModuleList([Linear(channels, channels), nn.Identity()])
        self.scale = channels ** -0.5

    def forward(self, x):
        n, h, w, c = x.shape
        tokens = self.group_norm(x, n).view(n, h * w, c)
        q, k = self.to_q(tokens), self.to_k(tokens)
        vt = ops.linear_vt(tokens.view(n * h * w, c), self.to_v.weight, self.to_v.bias, h * w)
        probs = ops.attention_scores(q, k, 1, self.scale)          # [n, 1, hw, hw]: materialised (d = 512)
        o = ops.attention_pv(probs, vt)
        return self.to_out[0](o, residual=x.view(n, h * w, c)).view(n, h, w, c)


class Downsample2D(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.conv = Conv2d(channels, channels, 3, stride=2, padding=1)      # parameters only; see forward

    def forward(self, x):
        # diffusers Downsample2D(padding=0): F.pad(x, (0, 1, 0, 1)) then a stride-2 conv without padding
        return ops.conv2d(x, self.conv.ohwi(), self.conv.bias, stride=2, padding=(0, 1))


class Upsample2D(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.conv = Conv2d(channels, channels, 3, padding=1)

    def forward(self, x):
        return self.conv(x, upsample=True)       # nearest-2x folded into the conv's loader


class _Block(nn.Module):
    def __init__(self, resnets, sampler_name=None, sampler=None):
        super().__init__()
        self.resnets = nn.ModuleList(resnets)
        if sampler is not None:
            setattr(self, sampler_name, nn.ModuleList([sampler]))
        self._sampler_name = sampler_name if sampler is not None else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self._sampler_name:
            x = getattr(self, self._sampler_name)[0](x)
        return x


class MidBlock(nn.Module):
    def __init__(self, channels, groups):
        super().__init__()
        self.attentions = nn.ModuleList([VaeAttention(channels, groups)])
        self.resnets = nn.ModuleList([ResnetBlock2D(channels, channels, groups), ResnetBlock2D(channels, channels, groups)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


def _pad_channels(x, to):
    if x.shape[-1] == to:
        return x
    out = x.new_zeros(*x.shape[:-1], to)
    out[..., :x.shape[-1]] = x
    return out


class Encoder(nn.Module):
    def __init__(self, in_channels, out_channels, boc, layers, groups, double_z=True):
        super().__init__()
        self.conv_in = Conv2d(in_channels, boc[0], 3, padding=1)
        blocks, ch = [], boc[0]
        for i, out_ch in enumerate(boc):
            res = [ResnetBlock2D(ch if j == 0 else out_ch, out_ch, groups) for j in range(layers)]
            blocks.append(_Block(res, 'downsamplers', Downsample2D(out_ch) if i < len(boc) - 1 else None))
            ch = out_ch
        self.down_blocks = nn.ModuleList(blocks)
        self.mid_block = MidBlock(boc[-1], groups)
        self.conv_norm_out = GroupNorm(groups, boc[-1], eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = Conv2d(boc[-1], 2 * out_channels if double_z else out_channels, 3, padding=1)

    def forward(self, x):                       # x [N, H, W, 8] (RGB zero-padded to 8 channels)
        x = self.conv_in(x)
        for b in self.down_blocks:
            x = b(x)
        x = self.mid_block(x)
        return self.conv_out(self.conv_norm_out(x, x.shape[0], silu=True))


class Decoder(nn.Module):
    def __init__(self, in_channels, out_channels, boc, layers, groups):
        super().__init__()
        rev = list(reversed(boc))
        self.conv_in = Conv2d(in_channels, rev[0], 3, padding=1)
        self.mid_block = MidBlock(rev[0], groups)
        blocks, ch = [], rev[0]
        for i, out_ch in enumerate(rev):
            res = [ResnetBlock2D(ch if j == 0 else out_ch, out_ch, groups) for j in range(layers + 1)]
            blocks.append(_Block(res, 'upsamplers', Upsample2D(out_ch) if i < len(rev) - 1 else None))
            ch = out_ch
        self.up_blocks = nn.ModuleList(blocks)
        self.conv_norm_out = GroupNorm(groups, rev[-1], eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = Conv2d(rev[-1], out_channels, 3, padding=1)

    def forward(self, z):                       # z [N, h, w, 8] (latent channels zero-padded)
        x = self.mid_block(self.conv_in(z))
        for b in self.up_blocks:
            x = b(x)
        x = self.conv_norm_out(x, x.shape[0], silu=True)
        # N = 3 output channels: run the GEMM with the weight rows padded to 8 (16-byte rows), slice afterwards
        w = self.conv_out.ohwi()
        pad_rows = (-w.shape[0]) % 8
        if pad_rows:
            w = torch.cat([w, w.new_zeros(pad_rows, *w.shape[1:])]).contiguous()
            b = torch.cat([self.conv_out.bias, self.conv_out.bias.new_zeros(pad_rows)])
        else:
            b = self.conv_out.bias
        return ops.conv2d(x, w, b)[..., :self.conv_out.out_channels]


class DiagonalGaussianDistribution:
    """diffusers.models.vae.DiagonalGaussianDistribution on [N, 2C, h, w] moments."""

    def __init__(self, parameters):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)

    def sample(self, generator=None):
        dev = generator.device if generator is not None else self.mean.device
        noise = torch.randn(self.mean.shape, generator=generator, device=dev, dtype=torch.float32)
        return self.mean + self.std * noise.to(device=self.mean.device, dtype=self.mean.dtype)

    def mode(self):
        return self.mean


@dataclass
class AutoencoderKLOutput(BaseOutput):
    latent_dist: DiagonalGaussianDistribution = None


@dataclass
class DecoderOutput(BaseOutput):
    sample: torch.Tensor = None


class AutoencoderKL(ModelMixin, ConfigMixin):
    @register_to_config
    def __init__(self, in_channels=3, out_channels=3, down_block_types=('DownEncoderBlock2D',),
                 up_block_types=('UpDecoderBlock2D',), block_out_channels=(64,), layers_per_block=1, act_fn='silu',
                 latent_channels=4, norm_num_groups=32, sample_size=32, scaling_factor=0.18215, **ignored):
        super().__init__()
        if act_fn != 'silu' or any(t != 'DownEncoderBlock2D' for t in down_block_types) \
                or any(t != 'UpDecoderBlock2D' for t in up_block_types):
            raise NotImplementedError('AutoencoderKL: only the SD VAE layout (DownEncoderBlock2D / UpDecoderBlock2D, silu)')
        boc = tuple(block_out_channels)
        self.encoder = Encoder(in_channels, latent_channels, boc, layers_per_block, norm_num_groups)
        self.decoder = Decoder(latent_channels, out_channels, boc, layers_per_block, norm_num_groups)
        self.quant_conv = Conv2d(2 * latent_channels, 2 * latent_channels, 1, padding=0)
        self.post_quant_conv = Conv2d(latent_channels, latent_channels, 1, padding=0)
        self.use_slicing = False

    def enable_slicing(self):
        self.use_slicing = True

    def disable_slicing(self):
        self.use_slicing = False

    def _chunks(self, x):
        return x.split(4) if self.use_slicing and x.shape[0] > 4 else (x,)

    @torch.no_grad()
    def encode(self, x, return_dict=True):
        """x [N, 3, H, W] in [-1, 1] -> latent_dist over [N, 4, H/8, W/8] (moments in the reference's layout)."""
        outs = []
        for part in self._chunks(x):
            nhwc = _pad_channels(part.to(self.dtype).permute(0, 2, 3, 1), 8).contiguous()
            h = self.encoder(nhwc)                                              # [n, h, w, 8]
            outs.append(self.quant_conv(h).permute(0, 3, 1, 2))
        dist = DiagonalGaussianDistribution(torch.cat(outs).contiguous())
        return AutoencoderKLOutput(latent_dist=dist) if return_dict else (dist,)

    @torch.no_grad()
    def decode(self, z, return_dict=True):
        """z [N, 4, h, w] -> images [N, 3, 8h, 8w]"""
        outs = []
        for part in self._chunks(z):
            zc = _pad_channels(part.to(self.dtype).permute(0, 2, 3, 1), 8).contiguous()
            zc = self.post_quant_conv(zc)                                       # 4 -> 4 (input channels padded to 8)
            img = self.decoder(_pad_channels(zc, 8).contiguous())
            outs.append(img.permute(0, 3, 1, 2))
        out = torch.cat(outs).contiguous()
        return DecoderOutput(sample=out) if return_dict else (out,)

    def forward(self, sample, sample_posterior=False, generator=None):
        dist = self.encode(sample).latent_dist
        return self.decode(dist.sample(generator) if sample_posterior else dist.mode())

    # ---- checkpoints ----
    _LEGACY = {'query': 'to_q', 'key': 'to_k', 'value': 'to_v', 'proj_attn': 'to_out.0'}

    def load_state_dict(self, state_dict, strict=True):
        renamed = {}
        for k, v in state_dict.items():
            if '.attentions.' in k:
                for old, new in self._LEGACY.items():
                    k = k.replace(f'.{old}.', f'.{new}.')
                if v.dim() == 4 and ('to_' in k):           # 1x1 conv attention weights of very old checkpoints
                    v = v[:, :, 0, 0]
            renamed[k] = v
        return super().load_state_dict(renamed, strict=strict)

    @classmethod
    def from_pretrained(cls, pretrained_model_path, subfolder=None, torch_dtype=None, **unused):
        path = os.path.join(pretrained_model_path, subfolder) if subfolder else pretrained_model_path
        with open(os.path.join(path, 'config.json')) as f:
            config = {k: v for k, v in json.load(f).items() if not k.startswith('_')}
        model = cls(**config)
        bin_file = os.path.join(path, 'diffusion_pytorch_model.bin')
        st_file = os.path.join(path, 'diffusion_pytorch_model.safetensors')
        if os.path.isfile(st_file):
            from safetensors.torch import load_file
            state = load_file(st_file)
        elif os.path.isfile(bin_file):
            state = formats.load_checkpoint(bin_file)
        else:
            raise RuntimeError(f'no VAE weights under {path}')
        model.load_state_dict(state, strict=True)
        return model.to(dtype=torch_dtype) if torch_dtype is not None else model


class VaeImageProcessor:
    """diffusers VaeImageProcessor as the pipeline uses it (pipeline_videoswap.py:165,609,651): PIL list -> [-1, 1]
    tensor [N, 3, H, W] (sides rounded down to multiples of the VAE scale factor) and back."""

    def __init__(self, vae_scale_factor=8, do_resize=True, do_normalize=True):
        self.vae_scale_factor, self.do_resize, self.do_normalize = vae_scale_factor, do_resize, do_normalize

    def preprocess(self, image):
        import numpy as np
        from PIL import Image
        if torch.is_tensor(image):
            return image
        if isinstance(image, Image.Image):
            image = [image]
        frames = []
        for img in image:
            if self.do_resize:
                w, h = (x - x % self.vae_scale_factor for x in img.size)
                if (w, h) != img.size:
                    img = img.resize((w, h), Image.LANCZOS)
            frames.append(torch.from_numpy(np.asarray(img.convert('RGB'), dtype=np.float32) / 255.0))
        x = torch.stack(frames).permute(0, 3, 1, 2)
        return 2.0 * x - 1.0 if self.do_normalize else x

    def postprocess(self, image, output_type='pil'):
        if output_type == 'latent' or not torch.is_tensor(image):
            return image
        x = (image.float() / 2 + 0.5).clamp(0, 1) if self.do_normalize else image.float()
        if output_type == 'pt':
            return x
        arr = x.cpu().permute(0, 2, 3, 1).numpy()
        if output_type == 'np':
            return arr
        from PIL import Image
        return [Image.fromarray((a * 255).round().astype('uint8')) for a in arr]
