"""TEST INFRASTRUCTURE — CPU fp32 restatement of VideoSwap's AnimateDiffUNet3DModel.

Plain PyTorch, reference tensor layout [B, C, F, H, W], parameter names identical to the reference's
state-dict keys (LoRA merging and checkpoint loading key on them: convert_edlora_to_diffusers.py:46-53,
test.py:63).  Every class cites the reference lines it follows (paths relative to
videoswap/models/animatediff_models/).  The third-party diffusers pieces come from
oracle/diffusers_restated.py.  tests/test_oracle_vs_reference.py checks this file against the reference's own
modules imported verbatim (with the restated diffusers stubs) and against tests/golden/.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import anything under oracle/.
"""
import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F
from torch import nn

from .diffusers_restated import (Attention, BaseOutput, ConfigMixin, FeedForward, ModelMixin, TimestepEmbedding,
                                 Timesteps, register_to_config)


def fold(x):
    """'b c f h w -> (b f) c h w' (resnet.py:14)"""
    b, c, f, h, w = x.shape
    return x.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)


def unfold(x, f):
    """'(b f) c h w -> b c f h w' (resnet.py:16)"""
    bf, c, h, w = x.shape
    return x.reshape(bf // f, f, c, h, w).permute(0, 2, 1, 3, 4)


class InflatedConv3d(nn.Conv2d):
    """resnet.py:9-18"""

    def forward(self, x):
        f = x.shape[2]
        return unfold(super().forward(fold(x)), f)


class Upsample3D(nn.Module):
    """resnet.py:21-69: nearest x[1,2,2] then 3x3 conv"""

    def __init__(self, channels, out_channels=None):
        super().__init__()
        self.conv = InflatedConv3d(channels, out_channels or channels, 3, padding=1)

    def forward(self, x, output_size=None):
        if output_size is None:
            x = F.interpolate(x, scale_factor=[1.0, 2.0, 2.0], mode='nearest')
        else:
            x = F.interpolate(x, size=output_size, mode='nearest')
        return self.conv(x)


class Downsample3D(nn.Module):
    """resnet.py:72-95: 3x3 stride-2 conv, padding 1"""

    def __init__(self, channels, out_channels=None, padding=1):
        super().__init__()
        self.conv = InflatedConv3d(channels, out_channels or channels, 3, stride=2, padding=padding)

    def forward(self, x):
        return self.conv(x)


class ResnetBlock3D(nn.Module):
    """resnet.py:98-193 (time_embedding_norm='default', swish).  GroupNorm is applied to the 5-D tensor, i.e.
    statistics pool over frames (resnet.py:166,177)."""

    def __init__(self, in_channels, out_channels, temb_channels, groups=32, eps=1e-5, output_scale_factor=1.0):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps, affine=True)
        self.conv1 = InflatedConv3d(in_channels, out_channels, 3, stride=1, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=eps, affine=True)
        self.conv2 = InflatedConv3d(out_channels, out_channels, 3, stride=1, padding=1)
        self.output_scale_factor = output_scale_factor
        self.conv_shortcut = None
        if in_channels != out_channels:
            self.conv_shortcut = InflatedConv3d(in_channels, out_channels, 1, stride=1, padding=0)

    def forward(self, x, temb):
        h = F.silu(self.norm1(x))
        h = self.conv1(h)
        h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None, None]
        h = F.silu(self.norm2(h))
        h = self.conv2(h)
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return (x + h) / self.output_scale_factor


class BasicTransformerBlock(nn.Module):
    """attention.py:148-256 with unet_use_cross_frame_attention = unet_use_temporal_attention = False"""

    def __init__(self, dim, heads, dim_head, cross_attention_dim):
        super().__init__()
        self.attn1 = Attention(query_dim=dim, heads=heads, dim_head=dim_head)
        self.norm1 = nn.LayerNorm(dim)
        self.attn2 = Attention(query_dim=dim, cross_attention_dim=cross_attention_dim, heads=heads, dim_head=dim_head)
        self.norm2 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim, activation_fn='geglu')
        self.norm3 = nn.LayerNorm(dim)

    def forward(self, x, encoder_hidden_states=None, video_length=None):
        x = self.attn1(self.norm1(x)) + x
        x = self.attn2(self.norm2(x), encoder_hidden_states=encoder_hidden_states) + x
        return self.ff(self.norm3(x)) + x


@dataclass
class Transformer3DModelOutput(BaseOutput):
    sample: torch.Tensor = None


class Transformer3DModel(nn.Module):
    """attention.py:31-145 (use_linear_projection False: 1x1 conv projections, per-frame GroupNorm eps 1e-6)"""

    def __init__(self, heads, dim_head, in_channels, cross_attention_dim, groups=32):
        super().__init__()
        inner = heads * dim_head
        self.norm = nn.GroupNorm(groups, in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Conv2d(in_channels, inner, 1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(inner, heads, dim_head, cross_attention_dim)])
        self.proj_out = nn.Conv2d(inner, in_channels, 1)

    def forward(self, x, encoder_hidden_states=None):
        f = x.shape[2]
        x = fold(x)
        if encoder_hidden_states.dim() == 3:   # 'b n c -> (b f) n c'
            ehs = encoder_hidden_states.repeat_interleave(f, dim=0)
        else:                                  # ED-LoRA 'b l n c -> (b f) l n c'
            ehs = encoder_hidden_states.repeat_interleave(f, dim=0)
        bf, c, h, w = x.shape
        res = x
        y = self.proj_in(self.norm(x))
        inner = y.shape[1]
        y = y.permute(0, 2, 3, 1).reshape(bf, h * w, inner)
        for blk in self.transformer_blocks:
            y = blk(y, encoder_hidden_states=ehs, video_length=f)
        y = y.reshape(bf, h, w, inner).permute(0, 3, 1, 2).contiguous()
        y = self.proj_out(y)
        return Transformer3DModelOutput(sample=unfold(y + res, f))


class PositionalEncoding(nn.Module):
    """motion_module.py:237-255"""

    def __init__(self, d_model, max_len=24):
        super().__init__()
        position = torch.arange(max_len).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
        pe = torch.zeros(1, max_len, d_model)
        pe[0, :, 0::2] = torch.sin(position * div_term)
        pe[0, :, 1::2] = torch.cos(position * div_term)
        self.register_buffer('pe', pe)

    def forward(self, x):
        return x + self.pe[:, :x.size(1)]


class VanillaAttentionProcessor(nn.Module):
    """motion_module.py:258-340: temporal self-attention with materialised probabilities"""

    def __init__(self, query_dim, temporal_position_encoding=True, max_len=24):
        super().__init__()
        self.is_cross_attention = False
        self.pos_encoder = PositionalEncoding(query_dim, max_len=max_len) if temporal_position_encoding else None

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None,
                 video_length=None):
        bf, d, c = hidden_states.shape
        b = bf // video_length
        x = hidden_states.reshape(b, video_length, d, c).permute(0, 2, 1, 3).reshape(b * d, video_length, c)
        if self.pos_encoder is not None:
            x = self.pos_encoder(x)
        q = attn.head_to_batch_dim(attn.to_q(x))
        k = attn.head_to_batch_dim(attn.to_k(x))
        v = attn.head_to_batch_dim(attn.to_v(x))
        probs = attn.get_attention_scores(q, k, None)
        x = attn.batch_to_head_dim(torch.bmm(probs, v))
        x = attn.to_out[1](attn.to_out[0](x))
        return x.reshape(b, d, video_length, c).permute(0, 2, 1, 3).reshape(bf, d, c)


class TemporalTransformerBlock(nn.Module):
    """motion_module.py:165-234"""

    def __init__(self, dim, heads, dim_head, n_attn=2, temporal_position_encoding=True, max_len=24):
        super().__init__()
        self.attention_blocks = nn.ModuleList([
            Attention(query_dim=dim, heads=heads, dim_head=dim_head,
                      processor=VanillaAttentionProcessor(dim, temporal_position_encoding, max_len))
            for _ in range(n_attn)])
        self.norms = nn.ModuleList([nn.LayerNorm(dim) for _ in range(n_attn)])
        self.ff = FeedForward(dim, activation_fn='geglu')
        self.ff_norm = nn.LayerNorm(dim)

    def forward(self, x, video_length=None):
        for attn, norm in zip(self.attention_blocks, self.norms):
            x = attn(norm(x), encoder_hidden_states=None, video_length=video_length) + x
        return self.ff(self.ff_norm(x)) + x


class TemporalTransformer3DModel(nn.Module):
    """motion_module.py:87-162"""

    def __init__(self, in_channels, heads, dim_head, num_layers, n_attn, temporal_position_encoding, max_len,
                 groups=32):
        super().__init__()
        inner = heads * dim_head
        self.norm = nn.GroupNorm(groups, in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Linear(in_channels, inner)
        self.transformer_blocks = nn.ModuleList([
            TemporalTransformerBlock(inner, heads, dim_head, n_attn, temporal_position_encoding, max_len)
            for _ in range(num_layers)])
        self.proj_out = nn.Linear(inner, in_channels)

    def forward(self, x):
        f = x.shape[2]
        x = fold(x)
        bf, c, h, w = x.shape
        res = x
        y = self.norm(x).permute(0, 2, 3, 1).reshape(bf, h * w, c)
        y = self.proj_in(y)
        for blk in self.transformer_blocks:
            y = blk(y, video_length=f)
        y = self.proj_out(y)
        y = y.reshape(bf, h, w, c).permute(0, 3, 1, 2).contiguous()
        return unfold(y + res, f)


class VanillaTemporalModule(nn.Module):
    """motion_module.py:48-84 (proj_out is zero-initialised: motion_module.py:76-77)"""

    def __init__(self, in_channels, num_attention_heads=8, num_transformer_block=1,
                 attention_block_types=('Temporal_Self', 'Temporal_Self'), temporal_position_encoding=True,
                 temporal_position_encoding_max_len=24, temporal_attention_dim_div=1, zero_initialize=True, groups=32):
        super().__init__()
        self.temporal_transformer = TemporalTransformer3DModel(
            in_channels, num_attention_heads, in_channels // num_attention_heads // temporal_attention_dim_div,
            num_transformer_block, len(attention_block_types), temporal_position_encoding,
            temporal_position_encoding_max_len, groups=groups)
        if zero_initialize:
            for p in self.temporal_transformer.proj_out.parameters():
                p.detach().zero_()

    def forward(self, x, temb=None, encoder_hidden_states=None):
        return self.temporal_transformer(x)


def _mm(ch, use, kw, groups):
    # the motion module's GroupNorm always has 32 groups (motion_module.py:95), whatever resnet_groups is
    return VanillaTemporalModule(ch, groups=32, **kw) if use else None


class CrossAttnDownBlock3D(nn.Module):
    """unet_blocks.py:268-412"""
    has_cross_attention = True

    def __init__(self, cin, cout, temb, layers, heads, xdim, eps, groups, add_down, use_mm, mm_kw):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock3D(cin if i == 0 else cout, cout, temb, groups, eps)
                                      for i in range(layers)])
        self.attentions = nn.ModuleList([Transformer3DModel(heads, cout // heads, cout, xdim, groups)
                                         for _ in range(layers)])
        self.motion_modules = nn.ModuleList([_mm(cout, use_mm, mm_kw, groups) for _ in range(layers)])
        self.downsamplers = nn.ModuleList([Downsample3D(cout, cout)]) if add_down else None

    def forward(self, x, temb=None, encoder_hidden_states=None, additional_residuals=None):
        outs = ()
        n = len(self.resnets)
        for i, (res, attn, mm) in enumerate(zip(self.resnets, self.attentions, self.motion_modules)):
            x = res(x, temb)
            x = attn(x, encoder_hidden_states=encoder_hidden_states).sample
            if mm is not None:
                x = mm(x, temb, encoder_hidden_states)
            if i == n - 1 and additional_residuals is not None:
                x = x + unfold(additional_residuals, x.shape[2])   # '(b f) c h w -> b c f h w' (:399-402)
            outs += (x,)
        if self.downsamplers is not None:
            for d in self.downsamplers:
                x = d(x)
            outs += (x,)
        return x, outs


class DownBlock3D(nn.Module):
    """unet_blocks.py:415-508"""
    has_cross_attention = False

    def __init__(self, cin, cout, temb, layers, eps, groups, add_down, use_mm, mm_kw):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock3D(cin if i == 0 else cout, cout, temb, groups, eps)
                                      for i in range(layers)])
        self.motion_modules = nn.ModuleList([_mm(cout, use_mm, mm_kw, groups) for _ in range(layers)])
        self.downsamplers = nn.ModuleList([Downsample3D(cout, cout)]) if add_down else None

    def forward(self, x, temb=None, encoder_hidden_states=None):
        outs = ()
        for res, mm in zip(self.resnets, self.motion_modules):
            x = res(x, temb)
            if mm is not None:
                x = mm(x, temb, encoder_hidden_states)
            outs += (x,)
        if self.downsamplers is not None:
            for d in self.downsamplers:
                x = d(x)
            outs += (x,)
        return x, outs


class UNetMidBlock3DCrossAttn(nn.Module):
    """unet_blocks.py:163-265"""
    has_cross_attention = True

    def __init__(self, ch, temb, heads, xdim, eps, groups, use_mm, mm_kw):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock3D(ch, ch, temb, groups, eps), ResnetBlock3D(ch, ch, temb, groups, eps)])
        self.attentions = nn.ModuleList([Transformer3DModel(heads, ch // heads, ch, xdim, groups)])
        self.motion_modules = nn.ModuleList([_mm(ch, use_mm, mm_kw, groups)])

    def forward(self, x, temb=None, encoder_hidden_states=None):
        x = self.resnets[0](x, temb)
        for attn, res, mm in zip(self.attentions, self.resnets[1:], self.motion_modules):
            x = attn(x, encoder_hidden_states=encoder_hidden_states).sample
            if mm is not None:
                x = mm(x, temb, encoder_hidden_states)
            x = res(x, temb)
        return x


class CrossAttnUpBlock3D(nn.Module):
    """unet_blocks.py:511-651"""
    has_cross_attention = True

    def __init__(self, cin, cout, prev, temb, layers, heads, xdim, eps, groups, add_up, use_mm, mm_kw):
        super().__init__()
        rs = []
        for i in range(layers):
            skip = cin if i == layers - 1 else cout
            rin = prev if i == 0 else cout
            rs.append(ResnetBlock3D(rin + skip, cout, temb, groups, eps))
        self.resnets = nn.ModuleList(rs)
        self.attentions = nn.ModuleList([Transformer3DModel(heads, cout // heads, cout, xdim, groups)
                                         for _ in range(layers)])
        self.motion_modules = nn.ModuleList([_mm(cout, use_mm, mm_kw, groups) for _ in range(layers)])
        self.upsamplers = nn.ModuleList([Upsample3D(cout, cout)]) if add_up else None

    def forward(self, x, res_hidden_states_tuple, temb=None, encoder_hidden_states=None, upsample_size=None):
        for res, attn, mm in zip(self.resnets, self.attentions, self.motion_modules):
            skip = res_hidden_states_tuple[-1]
            res_hidden_states_tuple = res_hidden_states_tuple[:-1]
            x = torch.cat([x, skip], dim=1)
            x = res(x, temb)
            x = attn(x, encoder_hidden_states=encoder_hidden_states).sample
            if mm is not None:
                x = mm(x, temb, encoder_hidden_states)
        if self.upsamplers is not None:
            for u in self.upsamplers:
                x = u(x, upsample_size)
        return x


class UpBlock3D(nn.Module):
    """unet_blocks.py:654-740"""
    has_cross_attention = False

    def __init__(self, cin, cout, prev, temb, layers, eps, groups, add_up, use_mm, mm_kw):
        super().__init__()
        rs = []
        for i in range(layers):
            skip = cin if i == layers - 1 else cout
            rin = prev if i == 0 else cout
            rs.append(ResnetBlock3D(rin + skip, cout, temb, groups, eps))
        self.resnets = nn.ModuleList(rs)
        self.motion_modules = nn.ModuleList([_mm(cout, use_mm, mm_kw, groups) for _ in range(layers)])
        self.upsamplers = nn.ModuleList([Upsample3D(cout, cout)]) if add_up else None

    def forward(self, x, res_hidden_states_tuple, temb=None, encoder_hidden_states=None, upsample_size=None):
        for res, mm in zip(self.resnets, self.motion_modules):
            skip = res_hidden_states_tuple[-1]
            res_hidden_states_tuple = res_hidden_states_tuple[:-1]
            x = torch.cat([x, skip], dim=1)
            x = res(x, temb)
            if mm is not None:
                x = mm(x, temb, encoder_hidden_states)
        if self.upsamplers is not None:
            for u in self.upsamplers:
                x = u(x, upsample_size)
        return x


@dataclass
class UNet3DConditionOutput(BaseOutput):
    sample: torch.Tensor = None


SD15_UNET_CONFIG = dict(sample_size=64, in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280),
                        layers_per_block=2, attention_head_dim=8, cross_attention_dim=768, norm_num_groups=32,
                        norm_eps=1e-5, flip_sin_to_cos=True, freq_shift=0)
# options/model_cfg/inference.yml:1-21
INFERENCE_KWARGS = dict(use_motion_module=True, motion_module_resolutions=(1, 2, 4, 8), motion_module_mid_block=False,
                        motion_module_type='Vanilla',
                        motion_module_kwargs=dict(num_attention_heads=8, num_transformer_block=1,
                                                  attention_block_types=('Temporal_Self', 'Temporal_Self'),
                                                  temporal_position_encoding=True,
                                                  temporal_position_encoding_max_len=24,
                                                  temporal_attention_dim_div=1),
                        unet_use_cross_frame_attention=False, unet_use_temporal_attention=False)


class AnimateDiffUNet3DModel(ModelMixin, ConfigMixin):
    """unet.py:32-481 restricted to the options the VideoSwap configs use."""

    @register_to_config
    def __init__(self, sample_size=None, in_channels=4, out_channels=4, flip_sin_to_cos=True, freq_shift=0,
                 block_out_channels=(320, 640, 1280, 1280), layers_per_block=2, norm_num_groups=32, norm_eps=1e-5,
                 cross_attention_dim=1280, attention_head_dim=8, use_motion_module=False,
                 motion_module_resolutions=(1, 2, 4, 8), motion_module_mid_block=False,
                 motion_module_decoder_only=False, motion_module_type=None, motion_module_kwargs={},
                 unet_use_cross_frame_attention=None, unet_use_temporal_attention=None):
        super().__init__()
        assert not unet_use_cross_frame_attention and not unet_use_temporal_attention
        boc = tuple(block_out_channels)
        temb = boc[0] * 4
        g, eps, heads, xdim = norm_num_groups, norm_eps, attention_head_dim, cross_attention_dim
        kw = dict(motion_module_kwargs)
        self.conv_in = InflatedConv3d(in_channels, boc[0], kernel_size=3, padding=(1, 1))
        self.time_proj = Timesteps(boc[0], flip_sin_to_cos, freq_shift)
        self.time_embedding = TimestepEmbedding(boc[0], temb)
        self.down_blocks = nn.ModuleList()
        out_ch = boc[0]
        for i in range(len(boc)):
            in_ch, out_ch = out_ch, boc[i]
            final = i == len(boc) - 1
            use_mm = use_motion_module and (2 ** i in motion_module_resolutions) and not motion_module_decoder_only
            if not final:   # 'CrossAttnDownBlock3D' x3 then 'DownBlock3D' (unet.py:490-495)
                blk = CrossAttnDownBlock3D(in_ch, out_ch, temb, layers_per_block, heads, xdim, eps, g, True, use_mm, kw)
            else:
                blk = DownBlock3D(in_ch, out_ch, temb, layers_per_block, eps, g, False, use_mm, kw)
            self.down_blocks.append(blk)
        self.mid_block = UNetMidBlock3DCrossAttn(boc[-1], temb, heads, xdim, eps, g,
                                                 use_motion_module and motion_module_mid_block, kw)
        self.num_upsamplers = 0
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(boc))
        out_ch = rev[0]
        for i in range(len(boc)):
            final = i == len(boc) - 1
            prev, out_ch = out_ch, rev[i]
            in_ch = rev[min(i + 1, len(boc) - 1)]
            use_mm = use_motion_module and (2 ** (3 - i) in motion_module_resolutions)
            if not final:
                self.num_upsamplers += 1
            if i == 0:     # 'UpBlock3D' then 'CrossAttnUpBlock3D' x3 (unet.py:496-501)
                blk = UpBlock3D(in_ch, out_ch, prev, temb, layers_per_block + 1, eps, g, not final, use_mm, kw)
            else:
                blk = CrossAttnUpBlock3D(in_ch, out_ch, prev, temb, layers_per_block + 1, heads, xdim, eps, g,
                                         not final, use_mm, kw)
            self.up_blocks.append(blk)
        self.conv_norm_out = nn.GroupNorm(num_channels=boc[0], num_groups=g, eps=eps)
        self.conv_act = nn.SiLU()
        self.conv_out = InflatedConv3d(boc[0], out_channels, kernel_size=3, padding=1)

    def forward(self, sample, timestep, encoder_hidden_states, class_labels=None, attention_mask=None,
                cross_attention_kwargs=None, down_block_additional_residuals=None, return_dict=True):
        # unet.py:376-397
        timesteps = timestep
        if not torch.is_tensor(timesteps):
            dtype = torch.float64 if isinstance(timestep, float) else torch.int64
            timesteps = torch.tensor([timesteps], dtype=dtype, device=sample.device)
        elif timesteps.dim() == 0:
            timesteps = timesteps[None].to(sample.device)
        timesteps = timesteps.expand(sample.shape[0])
        emb = self.time_embedding(self.time_proj(timesteps).to(dtype=self.dtype))
        sample = self.conv_in(sample)
        is_adapter = down_block_additional_residuals is not None
        # unet.py:413-440
        skips = (sample,)
        for blk in self.down_blocks:
            if blk.has_cross_attention:
                extra = {}
                if is_adapter and len(down_block_additional_residuals) > 0:
                    extra['additional_residuals'] = down_block_additional_residuals.pop(0)
                sample, res = blk(sample, temb=emb, encoder_hidden_states=encoder_hidden_states, **extra)
            else:
                sample, res = blk(sample, temb=emb, encoder_hidden_states=encoder_hidden_states)
                if is_adapter and len(down_block_additional_residuals) > 0:
                    add = down_block_additional_residuals.pop(0)
                    sample = sample + unfold(add, sample.shape[2])
            skips += res
        sample = self.mid_block(sample, emb, encoder_hidden_states=encoder_hidden_states)
        # unet.py:446-471
        for blk in self.up_blocks:
            n = len(blk.resnets)
            res, skips = skips[-n:], skips[:-n]
            sample = blk(sample, res, temb=emb, encoder_hidden_states=encoder_hidden_states)
        sample = self.conv_out(self.conv_act(self.conv_norm_out(sample)))
        if not return_dict:
            return (sample,)
        return UNet3DConditionOutput(sample=sample)


def tiny_config(channels=(64, 128, 256, 256), cross_attention_dim=64, max_len=24):
    """Small instance of the same architecture (8 heads, 32 groups, 2 layers/block) used by parity tests."""
    cfg = dict(SD15_UNET_CONFIG)
    cfg.update(block_out_channels=tuple(channels), cross_attention_dim=cross_attention_dim, sample_size=16)
    cfg.update(INFERENCE_KWARGS)
    cfg['motion_module_kwargs'] = dict(cfg['motion_module_kwargs'], temporal_position_encoding_max_len=max_len)
    return cfg


def full_config(max_len=24):
    cfg = dict(SD15_UNET_CONFIG)
    cfg.update(INFERENCE_KWARGS)
    cfg['motion_module_kwargs'] = dict(cfg['motion_module_kwargs'], temporal_position_encoding_max_len=max_len)
    return cfg


@torch.no_grad()
def synth_weights_(model, seed=1234):
    """Seeded synthetic weights: PyTorch default initialisers under `seed`, then the zero-initialised
    motion-module proj_out re-drawn N(0, 0.02^2) (otherwise temporal attention contributes exactly 0,
    motion_module.py:76-77) and every norm's affine parameters randomised."""
    g = torch.Generator().manual_seed(seed)
    for name, p in model.named_parameters():
        if p.dim() > 1:
            fan_in = p[0].numel()
            bound = 1.0 / math.sqrt(fan_in)
            p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * bound)
        elif 'norm' in name and name.endswith('weight'):
            p.copy_(torch.rand(p.shape, generator=g) + 0.5)
        elif 'norm' in name and name.endswith('bias'):
            p.copy_(torch.randn(p.shape, generator=g) * 0.1)
        else:
            p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * 0.05)
    for name, p in model.named_parameters():
        if 'temporal_transformer.proj_out' in name:
            p.copy_(torch.randn(p.shape, generator=g) * 0.02)
    return model
