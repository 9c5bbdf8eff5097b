"""`Attention` module + attention processors of the denoising path on the libvsx kernels.

Protocol parity with diffusers 0.19.3 `Attention` / `AttnProcessor*` as the reference uses them
(attention.py:174-194; motion_module.py:202-211,258-340; edlora_util.py:13-99; attention_register.py:15-211):

    processor(attn, hidden_states[B*F, N, C], encoder_hidden_states=None | [.., 77, 768] | [.., 16, 77, 768],
              attention_mask=None, temb=None, **kwargs) -> [B*F, N, C]

The module exposes to_q/to_k/to_v/to_out, heads, scale, head_to_batch_dim, batch_to_head_dim,
get_attention_scores, prepare_attention_mask, set_processor, ... so processors written against diffusers
(e.g. the reference's own files) keep working; they then run through the same HIP GEMM/softmax kernels.

Native processors (class attribute `vsx_native = True`) additionally accept
    video_length : frames per clip — the text embedding may then be passed UN-repeated ([B, 77, 768]): K/V are
                   projected once per clip and shared by its frames (the reference repeats the text over
                   frames and recomputes K/V per frame: attention.py:100-103);
    residual     : added in the to_out GEMM epilogue (saves one pass over the activation).
"""
import torch
from torch import nn

from . import ops
from .layers import Identity, Linear, StepInvariantCache


class Attention(nn.Module):
    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, dropout=0.0, bias=False,
                 upcast_attention=False, upcast_softmax=False, out_bias=True, scale_qk=True,
                 only_cross_attention=False, rescale_output_factor=1.0, residual_connection=False, processor=None,
                 **unsupported):
        super().__init__()
        if upcast_attention or upcast_softmax or dropout != 0.0 or unsupported:
            raise NotImplementedError(f'Attention: unsupported options {unsupported or "upcast/dropout"}')
        inner_dim = dim_head * heads
        cross_attention_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.upcast_attention = False
        self.upcast_softmax = False
        self.rescale_output_factor = rescale_output_factor
        self.residual_connection = residual_connection
        self.scale_qk = scale_qk
        self.scale = dim_head ** -0.5 if scale_qk else 1.0
        self.heads = heads
        self.sliceable_head_dim = heads
        self.added_kv_proj_dim = None
        self.only_cross_attention = only_cross_attention
        self.group_norm = None
        self.spatial_norm = None
        self.norm_cross = None
        self.to_q = Linear(query_dim, inner_dim, bias=bias)
        self.to_k = Linear(cross_attention_dim, inner_dim, bias=bias)
        self.to_v = Linear(cross_attention_dim, inner_dim, bias=bias)
        self.to_out = nn.ModuleList([Linear(inner_dim, query_dim, bias=out_bias), Identity()])
        self.set_processor(processor if processor is not None else AttnProcessor2_0())

    def fused_weight(self, names):
        """Row-concatenation of the projection weights in `names` (e.g. ('to_q', 'to_k')) so that projections of the
        same input run as ONE GEMM (the input panel is streamed once).  Cached; rebuilt when any of the parameters
        changes (LoRA merge / load_state_dict bump `_version`, .to()/.half() replace the storage)."""
        ws = [getattr(self, n).weight for n in names]
        key = tuple((w.data_ptr(), w._version, w.dtype) for w in ws)
        cache = self.__dict__.setdefault('_fused_cache', {})
        hit = cache.get(names)
        if hit is None or hit[0] != key:
            hit = (key, torch.cat([w.detach() for w in ws], dim=0).contiguous())
            cache[names] = hit
        return hit[1]

    processor_epoch = 0     # bumped by every set_processor (HIP-graph caches key on it)

    def set_processor(self, processor):
        Attention.processor_epoch += 1
        if (hasattr(self, 'processor') and isinstance(self.processor, nn.Module)
                and not isinstance(processor, nn.Module)):
            self._modules.pop('processor')
        self.processor = processor

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **cross_attention_kwargs):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask, **cross_attention_kwargs)

    # ---- diffusers helper surface used by foreign processors ----
    def head_to_batch_dim(self, tensor, out_dim=3):
        h = self.heads
        b, s, d = tensor.shape
        tensor = tensor.reshape(b, s, h, d // h).permute(0, 2, 1, 3)
        return tensor.reshape(b * h, s, d // h) if out_dim == 3 else tensor

    def batch_to_head_dim(self, tensor):
        h = self.heads
        b, s, d = tensor.shape
        return tensor.reshape(b // h, h, s, d).permute(0, 2, 1, 3).reshape(b // h, s, d * h)

    def get_attention_scores(self, query, key, attention_mask=None):
        """softmax(scale * q k^T) for head-batched [B*heads, N, d] operands (diffusers signature)."""
        if attention_mask is not None:
            raise NotImplementedError('attention masks are never used on the VideoSwap path')
        return ops.head_scores(query, key, self.scale)

    def prepare_attention_mask(self, attention_mask, target_length, batch_size=None, out_dim=3):
        if attention_mask is None:
            return None
        raise NotImplementedError('attention masks are never used on the VideoSwap path')


def _project_kv(attn, hidden_states, encoder_hidden_states, layer_idx=None):
    """K [nkvb, nk, C] and V^T [nkvb, C, ld] of the context (self: the tokens; cross: the text, with the ED-LoRA
    per-layer slice `[:, layer_idx]` of a [.., 16, 77, 768] embedding: edlora_util.py:39-43)."""
    def project(ctx):
        nkvb, nk, _ = ctx.shape
        k = attn.to_k(ctx)
        vt = ops.linear_vt(ctx.reshape(nkvb * nk, ctx.shape[-1]), attn.to_v.weight, attn.to_v.bias, nk)
        return k, vt, nk

    if encoder_hidden_states is None:
        return project(hidden_states)

    def from_text():
        ctx = encoder_hidden_states
        if ctx.dim() == 4:
            ctx = ctx[:, 0 if layer_idx is None else layer_idx]
        return project(ctx.contiguous())

    # the text embedding is the same tensor in every denoising step: its K / V^T are step-invariant
    cache = attn.__dict__.get('_text_kv')
    if cache is None:
        cache = attn.__dict__['_text_kv'] = StepInvariantCache(limit=8)
    return cache.get(encoder_hidden_states, layer_idx,
                     (attn.to_k.weight, attn.to_k.bias, attn.to_v.weight, attn.to_v.bias), from_text)


class _FusedProcessor:
    """Fused flash attention (never materialises the probabilities)."""
    vsx_native = True
    vsx_graph_safe = True       # launches only: may run inside a HIP-graph capture
    vsx_shareable = True        # stateless: identical batch items may share one call (unet._shared_cfg_prefix)

    def __init__(self, cross_attention_idx=None):
        self.cross_attention_idx = cross_attention_idx

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None,
                 video_length=None, residual=None):
        if attention_mask is not None:
            raise NotImplementedError('attention masks are never used on the VideoSwap path')
        nb = hidden_states.shape[0]
        if encoder_hidden_states is None and attn.to_q.bias is None and attn.to_k.bias is None:
            # self-attention: q and k in one GEMM ([.., 2C] buffer, consumed as column slices), V^T separately
            c = attn.to_q.out_features
            qk = ops.linear(hidden_states, attn.fused_weight(('to_q', 'to_k')))
            q, k = qk[..., :c], qk[..., c:]
            nk = hidden_states.shape[1]
            vt = ops.linear_vt(hidden_states.reshape(nb * nk, hidden_states.shape[-1]), attn.to_v.weight,
                               attn.to_v.bias, nk)
        else:
            q = attn.to_q(hidden_states)
            k, vt, nk = _project_kv(attn, hidden_states, encoder_hidden_states, self.cross_attention_idx)
        kv_div = nb // k.shape[0]
        o = ops.attention(q, k, vt, attn.heads, attn.scale, kv_div=kv_div, nk=nk)
        if attn.residual_connection:
            residual = hidden_states if residual is None else ops.axpy(residual, hidden_states)
        # every output of a fused attention feeds the block's next LayerNorm (attention.py:199,205): statistics ride along
        out = attn.to_out[0](o, residual=residual, row_stats=True)
        return out


class AttnProcessor2_0(_FusedProcessor):
    """Default processor of every spatial Attention (replaces F.scaled_dot_product_attention)."""


class AttnProcessor(_FusedProcessor):
    """diffusers' non-fused processor name; numerically the same attention here."""


class XFormersAttnProcessor(_FusedProcessor):
    def __init__(self, attention_op=None):
        super().__init__()
        self.attention_op = attention_op


class EDLoRA_AttnProcessor(_FusedProcessor):
    """edlora_util.py:13-82: cross-attention against layer `cross_attention_idx` of the [B, 16, 77, 768]
    multi-layer prompt embedding (single-layer embeddings pass through)."""

    def __init__(self, cross_attention_idx, attention_op=None):
        super().__init__(cross_attention_idx)
        self.attention_op = attention_op


class AttnControlProcessor:
    """attention_register.py:96-173: Prompt-to-Prompt hook.  Layers with fewer than 32^2 query tokens materialise
    their probabilities [b, heads, s, t], hand them to `controller(probs, is_cross, place_in_unet)` and multiply
    the (possibly edited) result with V; larger layers run the fused kernel, as the reference does with xformers."""
    vsx_native = True

    def __init__(self, place_in_unet, controller, attention_op=None, cross_attention_idx=None):
        self.place_in_unet = place_in_unet
        self.controller = controller
        self.attention_op = attention_op
        self.cross_attention_idx = cross_attention_idx

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None,
                 video_length=None, residual=None):
        nb, nq, _ = hidden_states.shape
        is_cross = encoder_hidden_states is not None
        q = attn.to_q(hidden_states)
        k, vt, nk = _project_kv(attn, hidden_states, encoder_hidden_states, self.cross_attention_idx)
        kv_div = nb // k.shape[0]
        if nq >= 32 ** 2:
            o = ops.attention(q, k, vt, attn.heads, attn.scale, kv_div=kv_div, nk=nk)
        else:
            probs = ops.attention_scores(q, k, attn.heads, attn.scale, kv_div=kv_div)
            probs = self.controller(probs, is_cross, self.place_in_unet)
            o = ops.attention_pv(probs, vt, kv_div=kv_div)
        if attn.residual_connection:
            residual = hidden_states if residual is None else ops.axpy(residual, hidden_states)
        return attn.to_out[0](o, residual=residual)


class EDLoRA_AttnControlProcessor(AttnControlProcessor):
    """attention_register.py:15-93"""

    def __init__(self, cross_attention_idx, place_in_unet, controller, attention_op=None):
        super().__init__(place_in_unet, controller, attention_op, cross_attention_idx)


class PositionalEncoding(nn.Module):
    """motion_module.py:237-255; the table is consumed by the LayerNorm kernel (fused add)."""

    def __init__(self, d_model, dropout=0.0, max_len=24):
        super().__init__()
        import math
        position = torch.arange(max_len).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
        pe = torch.zeros(1, max_len, d_model)
        pe[0, :, 0::2] = torch.sin(position * div_term)
        pe[0, :, 1::2] = torch.cos(position * div_term)
        self.register_buffer('pe', pe)

    def table(self):
        return self.pe[0]


class VanillaAttentionProcessor(nn.Module):
    """Temporal self-attention across frames (motion_module.py:258-340) on the native [B, F, HW, C] layout.

    The reference rearranges '(b f) d c -> (b d) f c', adds the positional encoding, projects q/k/v, materialises
    [b*d*heads, f, f] probabilities and rearranges back.  Here q/k/v are per-token GEMMs in place and the
    attention kernel gathers a site's F rows by stride; the positional encoding is added by the preceding
    LayerNorm kernel (`pe_applied=True`) or here.  In frame-sharded long-clip mode `kv_gather` all-gathers the
    K/V rows of the other ranks' frames over RCCL before the attention.
    """
    vsx_native = True

    @property
    def vsx_graph_safe(self):
        return self.kv_gather is None

    def __init__(self, attention_mode=None, cross_frame_attention_mode=None, temporal_position_encoding=False,
                 temporal_position_encoding_max_len=24, attention_op=None, *args, **kwargs):
        super().__init__()
        self.attention_op = attention_op
        self.attention_mode = attention_mode
        self.is_cross_attention = kwargs.get('cross_attention_dim') is not None
        if attention_mode != 'Temporal' or self.is_cross_attention:
            raise NotImplementedError('only Temporal_Self attention blocks exist in the VideoSwap configs')
        self.pos_encoder = PositionalEncoding(kwargs['query_dim'], dropout=0.0,
                                              max_len=temporal_position_encoding_max_len) \
            if temporal_position_encoding else None
        self.kv_gather = None      # set by videoswap_amd.distributed for frame sharding

    @property
    def frame_offset(self):
        """Global index of the first local frame (positional encoding) while the K|V all-gather form of frame sharding
        is active; 0 otherwise (unsharded, or the block runs on the site layout where every frame is local)."""
        g = self.kv_gather
        return g.frame_offset if g is not None and g.kv_active else 0

    def pe_table(self):
        return None if self.pos_encoder is None else self.pos_encoder.table()

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None,
                 video_length=None, residual=None, pe_applied=False):
        bf, hw, c = hidden_states.shape
        frames = video_length
        b = bf // frames
        x = hidden_states
        if self.pos_encoder is not None and not pe_applied:
            pe = self.pe_table()[self.frame_offset:self.frame_offset + frames].to(x.dtype)
            x = (x.view(b, frames, hw, c) + pe[None, :, None, :]).view(bf, hw, c)
        nobias = attn.to_q.bias is None and attn.to_k.bias is None and attn.to_v.bias is None
        fk = frames
        if self.kv_gather is not None and self.kv_gather.kv_active:
            # frame-sharded long clip: K|V of the local frames first (one GEMM, N = 2C), their all-gather over the
            # frame axis goes in flight, the q projection runs behind it
            if nobias:
                kv = ops.linear(x, attn.fused_weight(('to_k', 'to_v'))).view(-1, 2 * c)
            else:
                kv = torch.cat([attn.to_k(x).view(-1, c), attn.to_v(x).view(-1, c)], dim=1)
            handle = self.kv_gather.kv_gather_start(kv, b, frames, hw)
            q = attn.to_q(x).view(-1, c)
            kv_all, fk = self.kv_gather.kv_gather_finish(handle)
            k, v = kv_all[:, :c], kv_all[:, c:]
        elif nobias:
            qkv = ops.linear(x, attn.fused_weight(('to_q', 'to_k', 'to_v'))).view(-1, 3 * c)   # one GEMM, N = 3C
            q, k, v = qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:]
        else:
            q, k, v = attn.to_q(x).view(-1, c), attn.to_k(x).view(-1, c), attn.to_v(x).view(-1, c)
        o = ops.temporal_attention(q, k, v, b, frames, fk, hw, attn.heads, attn.scale).view(bf, hw, c)
        return attn.to_out[0](o, residual=residual, row_stats=True)      # the next LayerNorm of the block follows (motion_module.py:213-219)
