"""TEST INFRASTRUCTURE — plain-PyTorch fp32 restatement of diffusers 0.19.3 `AutoencoderKL` (models/autoencoder_kl.py,
models/vae.py Encoder / Decoder / DiagonalGaussianDistribution, models/unet_2d_blocks.py DownEncoderBlock2D /
UpDecoderBlock2D / UNetMidBlock2D, models/resnet.py ResnetBlock2D / Downsample2D / Upsample2D) as the reference uses
it (pipeline_videoswap.py:204-233 encode + sample + scaling, :603-610 decode).

diffusers is neither vendored in /root/reference nor installable here, and the reference has no VAE code or tests of
its own: PARITY UNPINNED (recalled from the published 0.19.3 source; state-dict keys follow that release, so a real
`vae/diffusion_pytorch_model.bin` loads).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import anything under oracle/."""
import torch
import torch.nn.functional as F
from torch import nn


class ResnetBlock2D(nn.Module):
    def __init__(self, in_channels, out_channels, groups=32, eps=1e-6):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=eps)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1) if in_channels != out_channels else None

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        return (x if self.conv_shortcut is None else self.conv_shortcut(x)) + h


class Attention(nn.Module):
    """diffusers Attention(heads=1, dim_head=channels, residual_connection=True, bias=True, norm_num_groups, eps) on
    a [N, C, H, W] input, default processor (softmax upcast to fp32)."""

    def __init__(self, channels, groups=32, eps=1e-6):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, channels, eps=eps)
        self.to_q = nn.Linear(channels, channels)
        self.to_k = nn.Linear(channels, channels)
        self.to_v = nn.Linear(channels, channels)
        self.to_out = nn.ModuleList([nn.Linear(channels, channels), nn.Dropout(0.0)])
        self.scale = channels ** -0.5

    def forward(self, x):
        n, c, h, w = x.shape
        t = x.view(n, c, h * w).transpose(1, 2)
        t = self.group_norm(t.transpose(1, 2)).transpose(1, 2)
        q, k, v = self.to_q(t), self.to_k(t), self.to_v(t)
        probs = torch.softmax((q @ k.transpose(1, 2) * self.scale).float(), dim=-1).to(q.dtype)
        o = self.to_out[1](self.to_out[0](probs @ v))
        return o.transpose(1, 2).reshape(n, c, h, w) + x


class Downsample2D(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, stride=2, padding=0)

    def forward(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1), mode='constant', value=0))


class Upsample2D(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode='nearest'))


class Block(nn.Module):
    def __init__(self, resnets, name=None, sampler=None):
        super().__init__()
        self.resnets = nn.ModuleList(resnets)
        self._name = name if sampler is not None else None
        if sampler is not None:
            setattr(self, name, nn.ModuleList([sampler]))

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        return getattr(self, self._name)[0](x) if self._name else x


class MidBlock(nn.Module):
    def __init__(self, channels, groups):
        super().__init__()
        self.attentions = nn.ModuleList([Attention(channels, groups)])
        self.resnets = nn.ModuleList([ResnetBlock2D(channels, channels, groups), ResnetBlock2D(channels, channels, groups)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class Encoder(nn.Module):
    def __init__(self, in_channels, out_channels, boc, layers, groups):
        super().__init__()
        self.conv_in = nn.Conv2d(in_channels, boc[0], 3, padding=1)
        blocks, ch = [], boc[0]
        for i, out_ch in enumerate(boc):
            res = [ResnetBlock2D(ch if j == 0 else out_ch, out_ch, groups) for j in range(layers)]
            blocks.append(Block(res, 'downsamplers', Downsample2D(out_ch) if i < len(boc) - 1 else None))
            ch = out_ch
        self.down_blocks = nn.ModuleList(blocks)
        self.mid_block = MidBlock(boc[-1], groups)
        self.conv_norm_out = nn.GroupNorm(groups, boc[-1], eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(boc[-1], 2 * out_channels, 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down_blocks:
            x = b(x)
        return self.conv_out(self.conv_act(self.conv_norm_out(self.mid_block(x))))


class Decoder(nn.Module):
    def __init__(self, in_channels, out_channels, boc, layers, groups):
        super().__init__()
        rev = list(reversed(boc))
        self.conv_in = nn.Conv2d(in_channels, rev[0], 3, padding=1)
        self.mid_block = MidBlock(rev[0], groups)
        blocks, ch = [], rev[0]
        for i, out_ch in enumerate(rev):
            res = [ResnetBlock2D(ch if j == 0 else out_ch, out_ch, groups) for j in range(layers + 1)]
            blocks.append(Block(res, 'upsamplers', Upsample2D(out_ch) if i < len(rev) - 1 else None))
            ch = out_ch
        self.up_blocks = nn.ModuleList(blocks)
        self.conv_norm_out = nn.GroupNorm(groups, rev[-1], eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(rev[-1], out_channels, 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for b in self.up_blocks:
            x = b(x)
        return self.conv_out(self.conv_act(self.conv_norm_out(x)))


class AutoencoderKL(nn.Module):
    def __init__(self, in_channels=3, out_channels=3, block_out_channels=(64,), layers_per_block=1, latent_channels=4,
                 norm_num_groups=32, scaling_factor=0.18215, **unused):
        super().__init__()
        boc = tuple(block_out_channels)
        self.encoder = Encoder(in_channels, latent_channels, boc, layers_per_block, norm_num_groups)
        self.decoder = Decoder(latent_channels, out_channels, boc, layers_per_block, norm_num_groups)
        self.quant_conv = nn.Conv2d(2 * latent_channels, 2 * latent_channels, 1)
        self.post_quant_conv = nn.Conv2d(latent_channels, latent_channels, 1)
        self.scaling_factor = scaling_factor

    def moments(self, x):
        return self.quant_conv(self.encoder(x))

    def encode_mode(self, x):
        return torch.chunk(self.moments(x), 2, dim=1)[0]

    def encode_sample(self, x, noise):
        mean, logvar = torch.chunk(self.moments(x), 2, dim=1)
        return mean + torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0)) * noise

    def decode(self, z):
        return self.decoder(self.post_quant_conv(z))


def tiny_vae_config():
    return dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(32, 64, 64, 64), layers_per_block=1,
                norm_num_groups=8, sample_size=64, scaling_factor=0.18215, act_fn='silu',
                down_block_types=('DownEncoderBlock2D',) * 4, up_block_types=('UpDecoderBlock2D',) * 4)


@torch.no_grad()
def synth_weights_(model, seed=77):
    import math
    g = torch.Generator().manual_seed(seed)
    for name, p in model.named_parameters():
        if p.dim() > 1:
            bound = 1.0 / math.sqrt(p[0].numel())
            p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * bound)
        elif 'norm' in name and name.endswith('weight'):
            p.copy_(torch.rand(p.shape, generator=g) + 0.5)
        else:
            p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * 0.05)
    return model
