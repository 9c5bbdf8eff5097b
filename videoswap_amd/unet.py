"""AnimateDiffUNet3DModel (SD-1.5 UNet3D + AnimateDiff motion modules) on the libvsx kernels.

Drop-in for the reference class of the same name (videoswap/models/animatediff_models/unet.py:32-523): same
constructor config, same `forward` signature and return types, same state-dict keys, same module tree names
(`down_blocks / mid_block / up_blocks`, `attn1 / attn2`, `attention_blocks`, ...) so `test.py`'s weight loading,
the ED-LoRA merge and the attention-processor registration walk it unchanged.

What differs is everything underneath: activations stay channels-last fp16 [B*F, H, W, C] end to end (the
reference's [B, C, F, H, W] is converted once at the entry and once at the exit), so none of the ~150 einops
rearrange copies, torch.cat skip concats or F.interpolate tensors of the reference exist — they are folded
into the loaders of the implicit-GEMM conv kernel — and bias / time-embedding / residual / GEGLU are GEMM
epilogues.
"""
import json
import math
import os
from dataclasses import dataclass

import torch
from torch import nn

from . import formats, ops
from .attention import Attention, VanillaAttentionProcessor
from .compat import MODEL_REGISTRY, BaseOutput, ConfigMixin, ModelMixin, register_to_config
from .layers import (FeedForward, GroupNorm, InflatedConv3d, LayerNorm, Linear, PointwiseConv, StepInvariantCache,
                     param_key)


class Geometry:
    """Shape of the channels-last activation [B*F, H, W, C] travelling through the blocks."""
    __slots__ = ('B', 'F', 'gn_hook', 'gn_frames', 'site_shard')

    def __init__(self, B, F, gn_hook=None, gn_frames=None, site_shard=None):
        self.B, self.F = B, F
        self.site_shard = site_shard  # frame-sharded mode with exchange='sites': frames <-> sites around motion modules
        self.gn_hook = gn_hook        # frame-sharded mode: all-reduce of 5-D GroupNorm partial sums
        self.gn_frames = gn_frames    # global frame count behind those statistics


def _native(attn):
    return getattr(attn.processor, 'vsx_native', False)


def _shareable(attn):
    """May this attention run once for two identical batch items?  Only the plain fused processors: they look at nothing but
    their arguments (the controller processors are `vsx_native` as well, but hand per-batch-half probabilities to host state)."""
    return getattr(attn.processor, 'vsx_shareable', False)


# ------------------------------------------------------------------------------------------------
# resnet.py
# ------------------------------------------------------------------------------------------------
class Upsample3D(nn.Module):
    """resnet.py:21-69; the nearest-2x upsample is an index transform inside the conv's A-operand loader."""

    def __init__(self, channels, use_conv=True, use_conv_transpose=False, out_channels=None, name='conv'):
        super().__init__()
        self.channels = channels
        self.out_channels = out_channels or channels
        self.conv = InflatedConv3d(channels, self.out_channels, 3, padding=1)

    def forward(self, x, geo, output_size=None):
        if output_size is not None:
            raise NotImplementedError('forced upsample size (latent sides must be multiples of 8)')
        return self.conv(x, upsample=True)


class Downsample3D(nn.Module):
    """resnet.py:72-95"""

    def __init__(self, channels, use_conv=True, out_channels=None, padding=1, name='conv'):
        super().__init__()
        self.channels = channels
        self.out_channels = out_channels or channels
        self.conv = InflatedConv3d(channels, self.out_channels, 3, stride=2, padding=padding)

    def forward(self, x, geo):
        return self.conv(x)


class ResnetBlock3D(nn.Module):
    """resnet.py:98-193: GN5D -> SiLU -> conv3x3 (+bias +time-emb, epilogue) -> GN5D -> SiLU -> conv3x3
    (+bias +shortcut, epilogue).  `x2` is the skip tensor the reference concatenates on C before the block
    (unet_blocks.py:618,720): the GroupNorm kernels, conv1 and the 1x1 shortcut read both sources directly."""

    def __init__(self, *, in_channels, out_channels=None, temb_channels=512, groups=32, eps=1e-6,
                 output_scale_factor=1.0, **unused):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels = in_channels, out_channels
        if output_scale_factor != 1.0:
            raise NotImplementedError('output_scale_factor != 1')
        self.norm1 = GroupNorm(groups, in_channels, eps=eps, affine=True)
        self.conv1 = InflatedConv3d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.time_emb_proj = Linear(temb_channels, out_channels)
        self.norm2 = GroupNorm(groups, out_channels, eps=eps, affine=True)
        self.conv2 = InflatedConv3d(out_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.conv_shortcut = None
        if in_channels != out_channels:
            self.conv_shortcut = InflatedConv3d(in_channels, out_channels, kernel_size=1, stride=1, padding=0)

    def _gn(self, norm, x, geo, x2=None):
        rows = None if geo.gn_frames is None else geo.gn_frames * x.shape[1] * x.shape[2]
        return norm(x, geo.B, silu=True, x2=x2, partial_hook=geo.gn_hook, count_rows=rows)

    def forward(self, x, silu_temb, geo, x2=None):
        bf, h, w, _ = x.shape
        hidden = self._gn(self.norm1, x, geo, x2)
        # [B, Cout]; step-invariant per timestep: the UNet hands out one cached silu_temb tensor per timestep value
        cache = self.__dict__.get('_tproj')
        if cache is None:
            cache = self.__dict__['_tproj'] = StepInvariantCache(limit=512)
        tproj = cache.get(silu_temb, None, (self.time_emb_proj.weight, self.time_emb_proj.bias),
                          lambda: self.time_emb_proj(silu_temb))
        # one embedding row per batch item, or a single row shared by the whole batch (scalar timestep)
        hidden = self.conv1(hidden, rowvec=tproj, rows_per_vec=geo.F * h * w if tproj.shape[0] > 1 else bf * h * w)
        hidden = self._gn(self.norm2, hidden, geo)
        if self.conv_shortcut is not None:
            shortcut = self.conv_shortcut(x, x2=x2)
        else:
            shortcut = x
        return self.conv2(hidden, residual=shortcut)


# ------------------------------------------------------------------------------------------------
# attention.py
# ------------------------------------------------------------------------------------------------
class BasicTransformerBlock(nn.Module):
    """attention.py:148-256 (unet_use_cross_frame_attention = unet_use_temporal_attention = False)."""

    def __init__(self, dim, num_attention_heads, attention_head_dim, cross_attention_dim=None, **unused):
        super().__init__()
        self.attn1 = Attention(query_dim=dim, heads=num_attention_heads, dim_head=attention_head_dim)
        self.norm1 = LayerNorm(dim)
        self.attn2 = Attention(query_dim=dim, cross_attention_dim=cross_attention_dim, heads=num_attention_heads,
                               dim_head=attention_head_dim)
        self.norm2 = LayerNorm(dim)
        self.ff = FeedForward(dim, activation_fn='geglu')
        self.norm3 = LayerNorm(dim)

    @staticmethod
    def _attend(attn, normed, x, text, frames):
        if _native(attn):
            return attn(normed, encoder_hidden_states=text, video_length=frames, residual=x)
        if text is not None and text.shape[0] != normed.shape[0]:   # foreign processor: repeat like the reference
            text = text.repeat_interleave(frames, dim=0)
        return ops.axpy(attn(normed, encoder_hidden_states=text), x)

    def forward(self, x, encoder_hidden_states=None, timestep=None, attention_mask=None, video_length=None,
                split_after_self=False):
        # the LayerNorms are folded into the projections that consume them when the processor is one of this package's
        x = self._attend(self.attn1, self.norm1(x, defer=_native(self.attn1)), x, None, video_length)
        if split_after_self:        # shared CFG prefix (AnimateDiffUNet3DModel._forward_body): the text makes the halves differ
            x = torch.cat([x, x])
        x = self._attend(self.attn2, self.norm2(x, defer=_native(self.attn2)), x, encoder_hidden_states, video_length)
        return self.ff(self.norm3(x, defer=True), residual=x)


@dataclass
class Transformer3DModelOutput(BaseOutput):
    sample: torch.Tensor = None


class Transformer3DModel(nn.Module):
    """attention.py:31-145: per-frame GroupNorm(eps 1e-6) -> 1x1 conv -> transformer block -> 1x1 conv -> +residual."""

    def __init__(self, num_attention_heads=16, attention_head_dim=88, in_channels=None, num_layers=1,
                 norm_num_groups=32, cross_attention_dim=None, use_linear_projection=False, **unused):
        super().__init__()
        if use_linear_projection or num_layers != 1:
            raise NotImplementedError('Transformer3DModel: SD-1.5 uses conv projections and one block')
        inner = num_attention_heads * attention_head_dim
        self.in_channels = in_channels
        self.norm = GroupNorm(norm_num_groups, in_channels, eps=1e-6, affine=True)
        self.proj_in = PointwiseConv(in_channels, inner, kernel_size=1)
        self.transformer_blocks = nn.ModuleList([
            BasicTransformerBlock(inner, num_attention_heads, attention_head_dim,
                                  cross_attention_dim=cross_attention_dim)])
        self.proj_out = PointwiseConv(inner, in_channels, kernel_size=1)

    def forward(self, x, geo, encoder_hidden_states=None, split_after_self=False):
        """`split_after_self`: x holds ONE copy of a classifier-free-guidance batch whose halves are identical so far; the
        result holds both halves (they part at the cross-attention)."""
        bf, h, w, c = x.shape
        y = self.norm(x, bf)
        y = self.proj_in(y.view(bf, h * w, c), row_stats=True)       # norm1 of the block follows
        for block in self.transformer_blocks:
            y = block(y, encoder_hidden_states=encoder_hidden_states, video_length=geo.F,
                      split_after_self=split_after_self)
        if split_after_self:
            x = torch.cat([x, x])
            bf *= 2
        y = self.proj_out(y, residual=x.view(bf, h * w, c))
        return y.view(bf, h, w, c)


# ------------------------------------------------------------------------------------------------
# motion_module.py
# ------------------------------------------------------------------------------------------------
class TemporalTransformerBlock(nn.Module):
    """motion_module.py:165-234"""

    def __init__(self, dim, num_attention_heads, attention_head_dim,
                 attention_block_types=('Temporal_Self', 'Temporal_Self'), cross_attention_dim=768,
                 temporal_position_encoding=False, temporal_position_encoding_max_len=24, **unused):
        super().__init__()
        blocks, norms = [], []
        for name in attention_block_types:
            if name != 'Temporal_Self':
                raise NotImplementedError(name)
            proc = VanillaAttentionProcessor(attention_mode='Temporal', temporal_position_encoding=temporal_position_encoding,
                                             temporal_position_encoding_max_len=temporal_position_encoding_max_len,
                                             cross_attention_dim=None, query_dim=dim)
            blocks.append(Attention(query_dim=dim, heads=num_attention_heads, dim_head=attention_head_dim,
                                    processor=proc))
            norms.append(LayerNorm(dim))
        self.attention_blocks = nn.ModuleList(blocks)
        self.norms = nn.ModuleList(norms)
        self.ff = FeedForward(dim, activation_fn='geglu')
        self.ff_norm = LayerNorm(dim)

    def forward(self, x, encoder_hidden_states=None, attention_mask=None, video_length=None):
        bf, hw, c = x.shape
        for attn, norm in zip(self.attention_blocks, self.norms):
            proc = attn.processor
            if _native(attn):
                pe = proc.pe_table() if hasattr(proc, 'pe_table') else None
                normed = norm(x, pe=pe, rows_per_frame=hw, frames=video_length,
                              frame_offset=getattr(proc, 'frame_offset', 0), defer=True)
                x = attn(normed, encoder_hidden_states=None, video_length=video_length, residual=x,
                         pe_applied=pe is not None)
            else:
                x = ops.axpy(attn(norm(x), encoder_hidden_states=None, video_length=video_length), x)
        return self.ff(self.ff_norm(x, defer=True), residual=x)


class TemporalTransformer3DModel(nn.Module):
    """motion_module.py:87-162"""

    def __init__(self, in_channels, num_attention_heads, attention_head_dim, num_layers,
                 attention_block_types=('Temporal_Self', 'Temporal_Self'), norm_num_groups=32,
                 cross_attention_dim=768, temporal_position_encoding=False, temporal_position_encoding_max_len=24,
                 **unused):
        super().__init__()
        inner = num_attention_heads * attention_head_dim
        self.norm = GroupNorm(norm_num_groups, in_channels, eps=1e-6, affine=True)
        self.proj_in = Linear(in_channels, inner)
        self.transformer_blocks = nn.ModuleList([
            TemporalTransformerBlock(inner, num_attention_heads, attention_head_dim,
                                     attention_block_types=attention_block_types,
                                     cross_attention_dim=cross_attention_dim,
                                     temporal_position_encoding=temporal_position_encoding,
                                     temporal_position_encoding_max_len=temporal_position_encoding_max_len)
            for _ in range(num_layers)])
        self.proj_out = Linear(inner, in_channels)

    def forward(self, x, geo):
        bf, h, w, c = x.shape
        y = self.norm(x, bf)
        y = self.proj_in(y.view(bf, h * w, c), row_stats=True)       # the first LayerNorm of the block follows
        shard = geo.site_shard
        if shard is None or not shard.use_sites(h * w):
            for block in self.transformer_blocks:
                y = block(y, video_length=geo.F)
        else:
            # long clip, frames sharded over the ranks: everything between proj_in and proj_out is local to a SITE, so
            # re-shard frames -> sites once, run the blocks on all F_total frames of this rank's sites, re-shard back
            inner = y.shape[-1]
            ys = shard.to_sites(y.view(-1, inner), geo.B, h * w)
            ys = ys.view(geo.B * shard.total_frames, -1, inner)
            shard.kv_active = False              # all frames of these sites are local: no K|V gather in the blocks
            try:
                for block in self.transformer_blocks:
                    ys = block(ys, video_length=shard.total_frames)
            finally:
                shard.kv_active = True
            y = shard.to_frames(ys.view(-1, inner), geo.B, h * w).view(bf, h * w, inner)
        y = self.proj_out(y, residual=x.view(bf, h * w, c))
        return y.view(bf, h, w, c)


class VanillaTemporalModule(nn.Module):
    """motion_module.py:48-84 (proj_out zero-initialised, as in AnimateDiff)."""

    def __init__(self, in_channels, num_attention_heads=8, num_transformer_block=2,
                 attention_block_types=('Temporal_Self', 'Temporal_Self'), cross_frame_attention_mode=None,
                 temporal_position_encoding=False, temporal_position_encoding_max_len=24,
                 temporal_attention_dim_div=1, zero_initialize=True, long_video_config=None):
        super().__init__()
        self.temporal_transformer = TemporalTransformer3DModel(
            in_channels=in_channels, num_attention_heads=num_attention_heads,
            attention_head_dim=in_channels // num_attention_heads // temporal_attention_dim_div,
            num_layers=num_transformer_block, attention_block_types=tuple(attention_block_types),
            temporal_position_encoding=temporal_position_encoding,
            temporal_position_encoding_max_len=temporal_position_encoding_max_len)
        if zero_initialize:
            for p in self.temporal_transformer.proj_out.parameters():
                p.detach().zero_()

    def forward(self, x, geo):
        return self.temporal_transformer(x, geo)


def get_motion_module(in_channels, motion_module_type, motion_module_kwargs):
    if motion_module_type != 'Vanilla':
        raise ValueError(motion_module_type)
    return VanillaTemporalModule(in_channels=in_channels, **motion_module_kwargs)


# ------------------------------------------------------------------------------------------------
# unet_blocks.py
# ------------------------------------------------------------------------------------------------
def _res(cin, cout, temb, eps, groups):
    return ResnetBlock3D(in_channels=cin, out_channels=cout, temb_channels=temb, eps=eps, groups=groups)


def _maybe_mm(ch, use, mtype, mkw):
    return get_motion_module(ch, mtype, mkw) if use else None


class CrossAttnDownBlock3D(nn.Module):
    """unet_blocks.py:268-412"""
    has_cross_attention = True

    def __init__(self, in_channels, out_channels, temb_channels, num_layers, resnet_eps, resnet_groups,
                 attn_num_head_channels, cross_attention_dim, add_downsample, use_motion_module, motion_module_type,
                 motion_module_kwargs):
        super().__init__()
        self.attn_num_head_channels = attn_num_head_channels
        self.resnets = nn.ModuleList([_res(in_channels if i == 0 else out_channels, out_channels, temb_channels,
                                           resnet_eps, resnet_groups) for i in range(num_layers)])
        self.attentions = nn.ModuleList([
            Transformer3DModel(attn_num_head_channels, out_channels // attn_num_head_channels,
                               in_channels=out_channels, cross_attention_dim=cross_attention_dim,
                               norm_num_groups=resnet_groups) for _ in range(num_layers)])
        self.motion_modules = nn.ModuleList([_maybe_mm(out_channels, use_motion_module, motion_module_type,
                                                       motion_module_kwargs) for _ in range(num_layers)])
        self.downsamplers = nn.ModuleList([Downsample3D(out_channels, out_channels=out_channels)]) \
            if add_downsample else None

    def forward(self, x, silu_temb, geo, encoder_hidden_states=None, additional_residuals=None, shared_geo=None):
        """`shared_geo` (first down block only): x is the single copy of two identical CFG halves, with its geometry; the
        first resnet and the first self-attention run once, the halves part at that transformer's cross-attention."""
        outs = ()
        last = len(self.resnets) - 1
        for i, (res, attn, mm) in enumerate(zip(self.resnets, self.attentions, self.motion_modules)):
            if i == 0 and shared_geo is not None:
                x = res(x, silu_temb, shared_geo)
                x = attn(x, shared_geo, encoder_hidden_states=encoder_hidden_states, split_after_self=True)
            else:
                x = res(x, silu_temb, geo)
                x = attn(x, geo, encoder_hidden_states=encoder_hidden_states)
            if mm is not None:
                x = mm(x, geo)
            if i == last and additional_residuals is not None:     # unet_blocks.py:399-402
                x = ops.axpy(x, additional_residuals)
            outs += (x,)
        if self.downsamplers is not None:
            for d in self.downsamplers:
                x = d(x, geo)
            outs += (x,)
        return x, outs


class DownBlock3D(nn.Module):
    """unet_blocks.py:415-508"""
    has_cross_attention = False

    def __init__(self, in_channels, out_channels, temb_channels, num_layers, resnet_eps, resnet_groups,
                 add_downsample, use_motion_module, motion_module_type, motion_module_kwargs):
        super().__init__()
        self.resnets = nn.ModuleList([_res(in_channels if i == 0 else out_channels, out_channels, temb_channels,
                                           resnet_eps, resnet_groups) for i in range(num_layers)])
        self.motion_modules = nn.ModuleList([_maybe_mm(out_channels, use_motion_module, motion_module_type,
                                                       motion_module_kwargs) for _ in range(num_layers)])
        self.downsamplers = nn.ModuleList([Downsample3D(out_channels, out_channels=out_channels)]) \
            if add_downsample else None

    def forward(self, x, silu_temb, geo, encoder_hidden_states=None):
        outs = ()
        for res, mm in zip(self.resnets, self.motion_modules):
            x = res(x, silu_temb, geo)
            if mm is not None:
                x = mm(x, geo)
            outs += (x,)
        if self.downsamplers is not None:
            for d in self.downsamplers:
                x = d(x, geo)
            outs += (x,)
        return x, outs


class UNetMidBlock3DCrossAttn(nn.Module):
    """unet_blocks.py:163-265"""
    has_cross_attention = True

    def __init__(self, in_channels, temb_channels, resnet_eps, resnet_groups, attn_num_head_channels,
                 cross_attention_dim, use_motion_module, motion_module_type, motion_module_kwargs):
        super().__init__()
        self.attn_num_head_channels = attn_num_head_channels
        self.resnets = nn.ModuleList([_res(in_channels, in_channels, temb_channels, resnet_eps, resnet_groups)
                                      for _ in range(2)])
        self.attentions = nn.ModuleList([
            Transformer3DModel(attn_num_head_channels, in_channels // attn_num_head_channels, in_channels=in_channels,
                               cross_attention_dim=cross_attention_dim, norm_num_groups=resnet_groups)])
        self.motion_modules = nn.ModuleList([_maybe_mm(in_channels, use_motion_module, motion_module_type,
                                                       motion_module_kwargs)])

    def forward(self, x, silu_temb, geo, encoder_hidden_states=None):
        x = self.resnets[0](x, silu_temb, geo)
        for attn, res, mm in zip(self.attentions, self.resnets[1:], self.motion_modules):
            x = attn(x, geo, encoder_hidden_states=encoder_hidden_states)
            if mm is not None:
                x = mm(x, geo)
            x = res(x, silu_temb, geo)
        return x


def _up_resnets(in_channels, out_channels, prev_output_channel, temb_channels, num_layers, eps, groups):
    out = []
    for i in range(num_layers):
        skip = in_channels if i == num_layers - 1 else out_channels
        rin = prev_output_channel if i == 0 else out_channels
        out.append(_res(rin + skip, out_channels, temb_channels, eps, groups))
    return nn.ModuleList(out)


class CrossAttnUpBlock3D(nn.Module):
    """unet_blocks.py:511-651"""
    has_cross_attention = True

    def __init__(self, in_channels, out_channels, prev_output_channel, temb_channels, num_layers, resnet_eps,
                 resnet_groups, attn_num_head_channels, cross_attention_dim, add_upsample, use_motion_module,
                 motion_module_type, motion_module_kwargs):
        super().__init__()
        self.attn_num_head_channels = attn_num_head_channels
        self.resnets = _up_resnets(in_channels, out_channels, prev_output_channel, temb_channels, num_layers,
                                   resnet_eps, resnet_groups)
        self.attentions = nn.ModuleList([
            Transformer3DModel(attn_num_head_channels, out_channels // attn_num_head_channels,
                               in_channels=out_channels, cross_attention_dim=cross_attention_dim,
                               norm_num_groups=resnet_groups) for _ in range(num_layers)])
        self.motion_modules = nn.ModuleList([_maybe_mm(out_channels, use_motion_module, motion_module_type,
                                                       motion_module_kwargs) for _ in range(num_layers)])
        self.upsamplers = nn.ModuleList([Upsample3D(out_channels, out_channels=out_channels)]) if add_upsample else None

    def forward(self, x, res_hidden_states_tuple, silu_temb, geo, encoder_hidden_states=None, upsample_size=None):
        for res, attn, mm in zip(self.resnets, self.attentions, self.motion_modules):
            skip = res_hidden_states_tuple[-1]
            res_hidden_states_tuple = res_hidden_states_tuple[:-1]
            x = res(x, silu_temb, geo, x2=skip)          # concat folded into the kernels
            x = attn(x, geo, encoder_hidden_states=encoder_hidden_states)
            if mm is not None:
                x = mm(x, geo)
        if self.upsamplers is not None:
            for u in self.upsamplers:
                x = u(x, geo, upsample_size)
        return x


class UpBlock3D(nn.Module):
    """unet_blocks.py:654-740"""
    has_cross_attention = False

    def __init__(self, in_channels, prev_output_channel, out_channels, temb_channels, num_layers, resnet_eps,
                 resnet_groups, add_upsample, use_motion_module, motion_module_type, motion_module_kwargs):
        super().__init__()
        self.resnets = _up_resnets(in_channels, out_channels, prev_output_channel, temb_channels, num_layers,
                                   resnet_eps, resnet_groups)
        self.motion_modules = nn.ModuleList([_maybe_mm(out_channels, use_motion_module, motion_module_type,
                                                       motion_module_kwargs) for _ in range(num_layers)])
        self.upsamplers = nn.ModuleList([Upsample3D(out_channels, out_channels=out_channels)]) if add_upsample else None

    def forward(self, x, res_hidden_states_tuple, silu_temb, geo, encoder_hidden_states=None, upsample_size=None):
        for res, mm in zip(self.resnets, self.motion_modules):
            skip = res_hidden_states_tuple[-1]
            res_hidden_states_tuple = res_hidden_states_tuple[:-1]
            x = res(x, silu_temb, geo, x2=skip)
            if mm is not None:
                x = mm(x, geo)
        if self.upsamplers is not None:
            for u in self.upsamplers:
                x = u(x, geo, upsample_size)
        return x


# ------------------------------------------------------------------------------------------------
# unet.py
# ------------------------------------------------------------------------------------------------
@dataclass
class UNet3DConditionOutput(BaseOutput):
    sample: torch.Tensor = None


class TimestepEmbedding(nn.Module):
    """diffusers TimestepEmbedding(320 -> 1280): linear_1, SiLU, linear_2 (unet.py:117)."""

    def __init__(self, in_channels, time_embed_dim):
        super().__init__()
        self.linear_1 = Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = Linear(time_embed_dim, time_embed_dim)

    def forward(self, sample):
        return self.linear_2(ops.silu(self.linear_1(sample)))


class Timesteps(nn.Module):
    """diffusers Timesteps / get_timestep_embedding (unet.py:114,391): [cos | sin] sinusoid in fp32.  `num_channels`
    scalars per timestep: computed on the host (the timestep is a host scalar in the loop) and uploaded."""

    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift):
        super().__init__()
        self.num_channels = num_channels
        self.flip_sin_to_cos = flip_sin_to_cos
        self.downscale_freq_shift = downscale_freq_shift

    def forward(self, timesteps):
        half = self.num_channels // 2
        exponent = -math.log(10000) * torch.arange(0, half, dtype=torch.float32) / (half - self.downscale_freq_shift)
        emb = timesteps.detach().cpu()[:, None].float() * torch.exp(exponent)[None, :]
        emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
        if self.flip_sin_to_cos:
            emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
        return emb


@MODEL_REGISTRY.register()
class AnimateDiffUNet3DModel(ModelMixin, ConfigMixin):
    _supports_gradient_checkpointing = False

    @register_to_config
    def __init__(self, sample_size=None, in_channels=4, out_channels=4, center_input_sample=False,
                 flip_sin_to_cos=True, freq_shift=0,
                 down_block_types=('CrossAttnDownBlock3D', 'CrossAttnDownBlock3D', 'CrossAttnDownBlock3D', 'DownBlock3D'),
                 mid_block_type='UNetMidBlock3DCrossAttn',
                 up_block_types=('UpBlock3D', 'CrossAttnUpBlock3D', 'CrossAttnUpBlock3D', 'CrossAttnUpBlock3D'),
                 only_cross_attention=False, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
                 downsample_padding=1, mid_block_scale_factor=1, act_fn='silu', norm_num_groups=32, norm_eps=1e-5,
                 cross_attention_dim=1280, attention_head_dim=8, dual_cross_attention=False,
                 use_linear_projection=False, class_embed_type=None, num_class_embeds=None, upcast_attention=False,
                 resnet_time_scale_shift='default',
                 use_motion_module=False, motion_module_resolutions=(1, 2, 4, 8), motion_module_mid_block=False,
                 motion_module_decoder_only=False, motion_module_type=None, motion_module_kwargs={},
                 unet_use_cross_frame_attention=None, unet_use_temporal_attention=None, **ignored):
        super().__init__()
        if (center_input_sample or only_cross_attention or dual_cross_attention or use_linear_projection
                or class_embed_type or num_class_embeds or upcast_attention or act_fn != 'silu'
                or resnet_time_scale_shift != 'default' or unet_use_cross_frame_attention
                or unet_use_temporal_attention or downsample_padding != 1 or mid_block_scale_factor != 1):
            raise NotImplementedError('AnimateDiffUNet3DModel: option outside the SD-1.5 / VideoSwap configuration')
        if mid_block_type != 'UNetMidBlock3DCrossAttn':
            raise ValueError(f'unknown mid_block_type : {mid_block_type}')
        boc = tuple(block_out_channels)
        self.sample_size = sample_size
        time_embed_dim = boc[0] * 4
        heads = attention_head_dim if isinstance(attention_head_dim, int) else attention_head_dim[0]
        mkw = dict(motion_module_kwargs)

        self.conv_in = InflatedConv3d(in_channels, boc[0], kernel_size=3, padding=(1, 1))
        self.time_proj = Timesteps(boc[0], flip_sin_to_cos, freq_shift)
        self.time_embedding = TimestepEmbedding(boc[0], time_embed_dim)

        self.down_blocks = nn.ModuleList()
        output_channel = boc[0]
        for i, kind in enumerate(down_block_types):
            input_channel, output_channel = output_channel, boc[i]
            final = i == len(boc) - 1
            use_mm = use_motion_module and (2 ** i in motion_module_resolutions) and not motion_module_decoder_only
            if kind == 'CrossAttnDownBlock3D':
                blk = CrossAttnDownBlock3D(input_channel, output_channel, time_embed_dim, layers_per_block, norm_eps,
                                           norm_num_groups, heads, cross_attention_dim, not final, use_mm,
                                           motion_module_type, mkw)
            elif kind == 'DownBlock3D':
                blk = DownBlock3D(input_channel, output_channel, time_embed_dim, layers_per_block, norm_eps,
                                  norm_num_groups, not final, use_mm, motion_module_type, mkw)
            else:
                raise ValueError(f'{kind} does not exist.')
            self.down_blocks.append(blk)

        self.mid_block = UNetMidBlock3DCrossAttn(boc[-1], time_embed_dim, norm_eps, norm_num_groups, heads,
                                                 cross_attention_dim, use_motion_module and motion_module_mid_block,
                                                 motion_module_type, mkw)

        self.num_upsamplers = 0
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(boc))
        output_channel = rev[0]
        for i, kind in enumerate(up_block_types):
            final = i == len(boc) - 1
            prev_output_channel, output_channel = output_channel, rev[i]
            input_channel = rev[min(i + 1, len(boc) - 1)]
            use_mm = use_motion_module and (2 ** (3 - i) in motion_module_resolutions)
            if not final:
                self.num_upsamplers += 1
            if kind == 'CrossAttnUpBlock3D':
                blk = CrossAttnUpBlock3D(input_channel, output_channel, prev_output_channel, time_embed_dim,
                                         layers_per_block + 1, norm_eps, norm_num_groups, heads, cross_attention_dim,
                                         not final, use_mm, motion_module_type, mkw)
            elif kind == 'UpBlock3D':
                blk = UpBlock3D(input_channel, prev_output_channel, output_channel, time_embed_dim,
                                layers_per_block + 1, norm_eps, norm_num_groups, not final, use_mm,
                                motion_module_type, mkw)
            else:
                raise ValueError(f'{kind} does not exist.')
            self.up_blocks.append(blk)

        self.conv_norm_out = GroupNorm(num_channels=boc[0], num_groups=norm_num_groups, eps=norm_eps)
        self.conv_act = nn.SiLU()
        self.conv_out = InflatedConv3d(boc[0], out_channels, kernel_size=3, padding=1)
        self._frame_shard = None      # videoswap_amd.distributed.FrameShard for the long-clip mode
        self._temb_cache = {}
        self._semb_cache = {}
        self._graphs = None           # graphs.GraphCache when HIP-graph replay is enabled
        self.vsx_cfg_keyword = True   # forward() accepts cfg_halves_equal (VideoSwapPipeline asks before passing it)
        self._weights_epoch = getattr(self, '_weights_epoch', 0)

    # ---------------------------------------------------------------------------------------------
    def forward(self, sample, timestep, encoder_hidden_states, class_labels=None, attention_mask=None,
                cross_attention_kwargs=None, down_block_additional_residuals=None, return_dict=True, cfg_halves_equal=False):
        """sample [B, C, F, H, W] fp16 on the GPU; timestep scalar / 0-dim / [B] tensor; encoder_hidden_states
        [B, 77, D] or [B, 16, 77, D]; down_block_additional_residuals: list the UNet pops from (unet.py:422,435),
        entries [(B F), C, h, w] (reference layout) or channels-last [(B F), h, w, C] tagged `.vsx_nhwc`.
        cfg_halves_equal (an extension of the reference signature): the caller's statement that
        sample[:B/2] and sample[B/2:] hold the same values (classifier-free guidance, pipeline_videoswap.py:556) — the UNet then
        computes everything in front of the first cross-attention once.  A stride-0 batch view proves the same by itself."""
        if attention_mask is not None or class_labels is not None:
            raise NotImplementedError('attention_mask / class_labels are never used on the VideoSwap path')
        if sample.dim() != 5:
            raise ValueError(f'expected [B, C, F, H, W], got {tuple(sample.shape)}')
        B, _, F, H, W = sample.shape
        up = 2 ** self.num_upsamplers
        if H % up or W % up:
            raise NotImplementedError(f'latent sides must be multiples of {up} (got {H}x{W})')
        # time embedding (unet.py:376-397)
        timesteps = timestep
        if not torch.is_tensor(timesteps):
            timesteps = torch.tensor([timesteps], dtype=torch.float64 if isinstance(timestep, float) else torch.int64)
        elif timesteps.dim() == 0:
            timesteps = timesteps[None]
        silu_emb = self._silu_time_embedding(timesteps, B, sample.device)

        residuals = down_block_additional_residuals
        if residuals is not None:
            # the caller's list is consumed (unet.py:422,435); layout conversion happens outside the captured body
            taken = [residuals.pop(0) for _ in range(len(residuals))]
            residuals = [r if getattr(r, 'vsx_nhwc', False) else r.permute(0, 2, 3, 1).contiguous() for r in taken]
        # decided HERE, on the caller's tensor and the caller's statement: a clone (graphs.py's static buffers) or any other op in
        # between would lose a stride-0 view, and nothing downstream re-derives it
        half = self._shared_cfg_prefix(sample, encoder_hidden_states, silu_emb, bool(cfg_halves_equal))
        if self._graphs is not None and self._graphable(sample, silu_emb, encoder_hidden_states):
            out = self._graphs.run(self, sample, silu_emb, encoder_hidden_states, residuals, half)
        else:
            out = self._forward_body(sample, silu_emb, encoder_hidden_states, residuals, half)
        if not return_dict:
            return (out,)
        return UNet3DConditionOutput(sample=out)

    def _forward_body(self, sample, silu_emb, encoder_hidden_states, residuals, half=0):
        """Everything between the time embedding and the output tensor: kernel launches on the current stream and
        device allocations only (no host synchronisation, no host-side data dependence), so that it can be captured
        into a HIP graph (`enable_hip_graphs`).  `residuals`: channels-last adapter maps or None.  `half` > 0: the batch is
        two equal CFG halves of `half` items (`_shared_cfg_prefix`, decided by `forward`): the prefix runs on one of them."""
        B, _, F, H, W = sample.shape
        shard = self._frame_shard
        geo = Geometry(B, F, None if shard is None else shard.gn_hook, None if shard is None else shard.total_frames,
                       shard if shard is not None and shard.exchange != 'kv' else None)

        # Classifier-free guidance hands the UNet the SAME latents twice (pipeline_videoswap.py:556: `torch.cat([latents] * 2)`);
        # the halves only part at the first cross-attention.  When the caller proves the duplication by passing a
        # stride-0 batch view (`latents.expand(2, ...)`: VideoSwapPipeline does), conv_in, the first resnet and the first
        # self-attention (N = H*W keys: the largest attention launch of the model) run once for both halves.
        shared_geo = None
        if half:
            shared_geo = Geometry(half, F)
            x = ops.pack_latents(sample[:half].contiguous(), 8)
        else:
            x = ops.pack_latents(sample.contiguous(), 8)        # [B*F, H, W, 8] (latent channels zero-padded)
        x = self.conv_in(x)

        is_adapter = residuals is not None
        residuals = list(residuals) if is_adapter else None

        def pop_residual():
            return residuals.pop(0)

        skips = (x if shared_geo is None else torch.cat([x, x]),)
        for blk in self.down_blocks:
            if blk.has_cross_attention:
                extra = pop_residual() if (is_adapter and len(residuals) > 0) else None
                kw = {}
                if shared_geo is not None:
                    kw['shared_geo'], shared_geo = shared_geo, None
                x, outs = blk(x, silu_emb, geo, encoder_hidden_states=encoder_hidden_states,
                              additional_residuals=extra, **kw)
            else:
                x, outs = blk(x, silu_emb, geo, encoder_hidden_states=encoder_hidden_states)
                if is_adapter and len(residuals) > 0:           # unet.py:434-438: after the skips are taken
                    x = ops.axpy(x, pop_residual())
            skips += outs

        x = self.mid_block(x, silu_emb, geo, encoder_hidden_states=encoder_hidden_states)

        for blk in self.up_blocks:
            n = len(blk.resnets)
            res, skips = skips[-n:], skips[:-n]
            x = blk(x, res, silu_emb, geo, encoder_hidden_states=encoder_hidden_states)

        rows = None if geo.gn_frames is None else geo.gn_frames * H * W
        x = self.conv_norm_out(x, B, silu=True, partial_hook=geo.gn_hook, count_rows=rows)
        x = self.conv_out(x)
        return ops.unpack_latents(x, B, self.config.out_channels)

    # ---------------------------------------------------------------------------------------------
    def enable_hip_graphs(self, enabled=True, eager_every=0):
        """Replay the forward as a captured HIP graph (one per input signature) instead of ~700 eager launches: the
        launch-bound 16x16 / 8x8 levels leave the GPU idle between kernels in eager mode (profiles/r01_kernel_stats_v7:
        8.3 % of the span).  `eager_every = k > 0` runs every k-th call eagerly (bench.py brackets the GEMM launches of
        those calls with hipEvents: events cannot bracket the kernels of a graph replay).  Graphs are dropped when the
        weights or the attention processors change (load_state_dict / .to() / set_processor); call
        `invalidate_graphs()` after editing parameters in place."""
        from .graphs import GraphCache
        self._graphs = GraphCache(eager_every) if enabled else None

    def enable_gradient_checkpointing(self):
        """train.py:85-87 calls it when the option file says `gradient_checkpointing: true`.  The reference needs it to fit
        its activations in 24-48 GB; one 288-GB GPU keeps the ~40 GB tape of a 16-frame 512x512 step resident, so nothing
        is recomputed here."""
        self.gradient_checkpointing = True

    def invalidate_graphs(self):
        self._weights_epoch += 1

    def load_state_dict(self, *args, **kwargs):
        self._weights_epoch += 1
        return super().load_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        self._weights_epoch = getattr(self, '_weights_epoch', 0) + 1
        return super()._apply(fn, *args, **kwargs)

    def _shared_cfg_prefix(self, sample, text, silu_emb, stated=False):
        """Can the two CFG halves of the batch share everything in front of the first cross-attention?  -> the number of batch
        items of ONE half (0: no).  Only when the caller PROVES that the halves are identical — a stride-0 batch dimension
        (`latents.expand(2, ...)`: one clip), or, for several clips denoised together, the keyword `cfg_halves_equal=True` that
        `VideoSwapPipeline.__call__` passes with its own `torch.cat([latents] * 2)` (`stated`) — AND they share the timestep (one time-
        embedding row: a [B] timestep tensor gives B rows, and the shared prefix would hand the first half's rows to both), the
        first block is a cross-attention block whose first self- and cross-attention run on this package's plain fused
        processors (a Prompt-to-Prompt controller hooked there is `vsx_native` too, but it expects both halves of the batch:
        below 32 x 32 latents it is called on that very layer), and the clip is not frame-sharded."""
        nb = sample.shape[0]
        proven = (nb == 2 and sample.stride(0) == 0) or (nb % 2 == 0 and stated)
        if not proven or self._frame_shard is not None:
            return 0
        if silu_emb.shape[0] != 1:
            return 0
        if text is None or text.shape[0] != nb or os.environ.get('VSX_CFG_SHARED_PREFIX', '1') == '0':
            return 0
        blk = self.down_blocks[0]
        if not getattr(blk, 'has_cross_attention', False) or len(blk.attentions) == 0:
            return 0
        tb = blk.attentions[0].transformer_blocks[0]
        return nb // 2 if (_shareable(tb.attn1) and _shareable(tb.attn2)) else 0

    def _graphable(self, sample, silu_emb, text):
        if self._frame_shard is not None or not sample.is_cuda or silu_emb.shape[0] != 1:
            return False
        from .attention import Attention
        if self.__dict__.get('_native_epoch') != Attention.processor_epoch:
            # every processor must be a fused native one: controllers keep host-side state per call
            self.__dict__['_all_native'] = all(
                getattr(m.processor, 'vsx_graph_safe', False) for m in self.modules() if isinstance(m, Attention))
            self.__dict__['_native_epoch'] = Attention.processor_epoch
        return self.__dict__['_all_native']

    @staticmethod
    def _device_key(device):
        """One spelling per device for the step caches: 'cuda' (what a caller's `pipe.to('cuda')` hands over) and 'cuda:0' (what a
        tensor on it reports) are the same device, and a row prepared under one must be found under the other."""
        d = torch.device(device)
        if d.type == 'cuda' and d.index is None:
            d = torch.device('cuda', torch.cuda.current_device())
        return str(d)

    def _silu_time_embedding(self, timesteps, B, device):
        """SiLU(time_embedding(time_proj(t))) (unet.py:376-397; every consumer applies SiLU first, resnet.py:172).
        A scalar timestep gives ONE row [1, 1280] shared by the batch (the resnets broadcast it through the conv
        epilogue's row-vector index), cached per timestep value: DDIM inversion and sampling of a clip visit the same
        grid 1, 21, ..., 961, and because the SAME tensor object comes back for the same timestep, the per-resnet
        `time_emb_proj` results are cached on it as well (`clear_step_caches()` drops all of it)."""
        te = self.time_embedding
        if timesteps.numel() == 1:
            key = (float(timesteps.reshape(-1)[0]), self._device_key(device), self.dtype)
            stamp = param_key(te.linear_1.weight, te.linear_1.bias, te.linear_2.weight, te.linear_2.bias)
            hit = self._semb_cache.get(key)
            if hit is not None and hit[0] == stamp:
                return hit[1]
            val = ops.silu(te(self._timestep_features(timesteps, 1, device)))
            if len(self._semb_cache) >= 1024:
                self._semb_cache.clear()
            self._semb_cache[key] = (stamp, val)
            return val
        return ops.silu(te(self._timestep_features(timesteps, B, device)))

    def prepare_timesteps(self, timesteps, device=None):
        """Everything that depends on the timestep alone, for ALL timesteps of a denoising loop in three launches instead of 26
        per step: the loops (pipeline_videoswap.py:555, :677) call the UNet with 50 scalar timesteps that are known when the
        scheduler has been set, and per timestep the reference computes the sinusoid, the two time-embedding Linears
        (unet.py:391-397) and one `time_emb_proj(SiLU(emb))` row per resnet (resnet.py:172-176: 22 of them) — M = 1 GEMMs of
        13 us each, 0.4 % of a one-clip step pair (profiles/r06_gemm_traffic_by_shape_b1.txt: `gemm M=1 1280->1280`).  Here: one
        [T, 320] upload, the two Linears at M = T, and ONE GEMM of the [T, 1280] SiLU rows against the concatenated projection
        weights [sum C_out, 1280]; the rows are then handed out through the same caches the per-step path fills (`_semb_cache`: one
        tensor OBJECT per timestep value, and on it the per-resnet `_tproj` entries), so the forward itself is unchanged and a
        timestep that was not prepared is still computed on demand.  Scalar timesteps only; parameters are versioned as
        everywhere (a LoRA merge / reload recomputes)."""
        if self._frame_shard is not None or self._graphs is not None or os.environ.get('VSX_PREPARE_TIMESTEPS', '1') == '0':
            return          # (the environment switch keeps the per-step path for A/B runs)
        device = device if device is not None else next(self.parameters()).device
        te = self.time_embedding
        stamp = param_key(te.linear_1.weight, te.linear_1.bias, te.linear_2.weight, te.linear_2.bias)
        vals = [float(t) for t in (timesteps.reshape(-1).tolist() if torch.is_tensor(timesteps) else timesteps)]
        todo = []
        for v in vals:
            hit = self._semb_cache.get((v, self._device_key(device), self.dtype))
            if (hit is None or hit[0] != stamp) and v not in todo:
                todo.append(v)
        resnets = [m for m in self.modules() if isinstance(m, ResnetBlock3D)]
        if len(todo) < 2 or not resnets:
            return
        tt = torch.tensor(todo, dtype=torch.float64 if any(v != int(v) for v in todo) else torch.int64)
        feats = self.time_proj(tt).to(device=device, dtype=self.dtype)                  # [T, 320]
        semb = ops.silu(te(feats))                                                      # [T, 1280]
        wkey = param_key(*[p for r in resnets for p in (r.time_emb_proj.weight, r.time_emb_proj.bias)])
        cat = self.__dict__.get('_tproj_cat')
        if cat is None or cat[0] != wkey:
            cat = (wkey, torch.cat([r.time_emb_proj.weight.detach() for r in resnets]).contiguous(),
                   torch.cat([r.time_emb_proj.bias.detach() for r in resnets]).contiguous())
            self.__dict__['_tproj_cat'] = cat
        proj = ops.linear(semb, cat[1], cat[2])                                         # [T, sum C_out]
        if len(self._semb_cache) + len(todo) >= 1024:
            self._semb_cache.clear()
        for i, v in enumerate(todo):
            row = semb[i:i + 1]
            self._semb_cache[(v, self._device_key(device), self.dtype)] = (stamp, row)
            o = 0
            for r in resnets:
                c = r.out_channels
                cache = r.__dict__.get('_tproj')
                if cache is None:
                    cache = r.__dict__['_tproj'] = StepInvariantCache(limit=512)
                cache.put(row, None, (r.time_emb_proj.weight, r.time_emb_proj.bias), proj[i:i + 1, o:o + c])
                o += c

    def clear_step_caches(self):
        """Drop the step-invariant host caches (time-embedding rows per timestep, per-resnet projections of them,
        text K/V per cross-attention layer).  They are keyed on parameter versions, so this is never needed for
        correctness; bench.py calls it at the start of every timed clip so that no clip profits from the previous."""
        self._semb_cache.clear()
        self._temb_cache.clear()
        for m in self.modules():
            m.__dict__.pop('_tproj', None)
            m.__dict__.pop('_text_kv', None)

    def _timestep_features(self, timesteps, B, device):
        """Sinusoidal features [B, 320] on the device.  The loops visit the same 50 (+50) timesteps for every clip,
        so the uploaded rows are cached per value: an H2D copy from pageable memory would otherwise drain the stream
        once per UNet call and keep the host from enqueueing ahead of the GPU."""
        if timesteps.numel() == 1:
            key = (float(timesteps.reshape(-1)[0]), self._device_key(device), self.dtype)
            row = self._temb_cache.get(key)
            if row is None:
                row = self.time_proj(timesteps.reshape(1)).to(device=device, dtype=self.dtype)
                if len(self._temb_cache) < 4096:
                    self._temb_cache[key] = row
            return row.expand(B, -1).contiguous() if B > 1 else row
        return self.time_proj(timesteps.expand(B)).to(device=device, dtype=self.dtype)

    # ---------------------------------------------------------------------------------------------
    @classmethod
    def from_pretrained_2d(cls, pretrained_model_path, subfolder=None, unet_additional_kwargs=None):
        """unet.py:483-523: build from a 2-D SD `unet/config.json` + `diffusion_pytorch_model.bin` (strict=False:
        the motion-module weights are loaded afterwards by the caller, test.py:60-64)."""
        if subfolder is not None:
            pretrained_model_path = os.path.join(pretrained_model_path, subfolder)
        config_file = os.path.join(pretrained_model_path, 'config.json')
        if not os.path.isfile(config_file):
            raise RuntimeError(f'{config_file} does not exist')
        with open(config_file, 'r') as f:
            config = json.load(f)
        config['_class_name'] = cls.__name__
        config['down_block_types'] = ['CrossAttnDownBlock3D'] * 3 + ['DownBlock3D']
        config['up_block_types'] = ['UpBlock3D'] + ['CrossAttnUpBlock3D'] * 3
        config['mid_block_type'] = 'UNetMidBlock3DCrossAttn'
        model = cls.from_config(config, **(unet_additional_kwargs or {}))
        model_file = os.path.join(pretrained_model_path, 'diffusion_pytorch_model.bin')
        if not os.path.isfile(model_file):
            raise RuntimeError(f'{model_file} does not exist')
        state_dict = formats.load_checkpoint(model_file)
        missing, unexpected = model.load_state_dict(state_dict, strict=False)
        print(f'### missing keys: {len(missing)}; \n### unexpected keys: {len(unexpected)};')
        params = [p.numel() if 'temporal' in n else 0 for n, p in model.named_parameters()]
        print(f'### Temporal Module Parameters: {sum(params) / 1e6} M')
        return model


# options/model_cfg/inference.yml:1-21 and the public SD-1.5 unet/config.json
SD15_UNET_CONFIG = dict(sample_size=64, in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280),
                        layers_per_block=2, attention_head_dim=8, cross_attention_dim=768, norm_num_groups=32,
                        norm_eps=1e-5, flip_sin_to_cos=True, freq_shift=0)


def inference_kwargs(max_len=24):
    return dict(use_motion_module=True, motion_module_resolutions=(1, 2, 4, 8), motion_module_mid_block=False,
                motion_module_type='Vanilla',
                motion_module_kwargs=dict(num_attention_heads=8, num_transformer_block=1,
                                          attention_block_types=('Temporal_Self', 'Temporal_Self'),
                                          temporal_position_encoding=True, temporal_position_encoding_max_len=max_len,
                                          temporal_attention_dim_div=1),
                unet_use_cross_frame_attention=False, unet_use_temporal_attention=False)
