"""TEST INFRASTRUCTURE — CPU restatement of the `diffusers==0.19.3` pieces VideoSwap's denoising path uses.

The reference pins `diffusers==0.19.3` (requirements.txt:2, README.md:46) but does not vendor it and it
is not installable here (no network).  The classes below restate, from the published 0.19.x source, only
the arithmetic and the attribute/protocol surface the reference touches.  PARITY UNPINNED for this file:
it cannot be diffed against the real package in this container; the DDIMInverseScheduler variant in
particular is documented in DESIGN.md.  Call sites in the reference are cited per class.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import anything under oracle/.
"""
import math
from collections import OrderedDict
from dataclasses import dataclass, fields

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn


# ------------------------------------------------------------------------------------------------
# configuration / model mixins (no arithmetic) — used by unet.py:32-35, attention.py:31-33,
# adapter_model.py:50-54
# ------------------------------------------------------------------------------------------------
class FrozenDict(OrderedDict):
    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError as e:
            raise AttributeError(name) from e


def register_to_config(init):
    import functools
    import inspect

    @functools.wraps(init)
    def inner(self, *args, **kwargs):
        sig = inspect.signature(init)
        params = [p for n, p in sig.parameters.items() if n != 'self']
        cfg = {p.name: p.default for p in params if p.default is not inspect.Parameter.empty}
        for a, p in zip(args, params):
            cfg[p.name] = a
        cfg.update({k: v for k, v in kwargs.items() if not k.startswith('_')})
        self._internal_dict = FrozenDict(cfg)
        init(self, *args, **{k: v for k, v in kwargs.items() if not k.startswith('_')})
    return inner


class ConfigMixin:
    @property
    def config(self):
        return self._internal_dict

    @classmethod
    def from_config(cls, config, **kwargs):
        import inspect
        accepted = set(inspect.signature(cls.__init__).parameters)
        cfg = {k: v for k, v in dict(config).items() if k in accepted}
        cfg.update({k: v for k, v in kwargs.items() if k in accepted})
        return cls(**cfg)


class ModelMixin(nn.Module):
    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device


class BaseOutput(OrderedDict):
    """dataclass-style output that also indexes like a tuple (diffusers.utils.BaseOutput)."""

    def __post_init__(self):
        for f in fields(self):
            v = getattr(self, f.name)
            if v is not None:
                self[f.name] = v

    def __getitem__(self, k):
        if isinstance(k, str):
            return dict(self.items())[k]
        return tuple(self.values())[k]

    def to_tuple(self):
        return tuple(self.values())


# ------------------------------------------------------------------------------------------------
# embeddings — unet.py:114-117,391-397
# ------------------------------------------------------------------------------------------------
def get_timestep_embedding(timesteps, embedding_dim, flip_sin_to_cos=False, downscale_freq_shift=1.0,
                           scale=1.0, max_period=10000):
    half = embedding_dim // 2
    exponent = -math.log(max_period) * torch.arange(0, half, dtype=torch.float32, device=timesteps.device)
    exponent = exponent / (half - downscale_freq_shift)
    emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
    emb = scale * emb
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    if embedding_dim % 2 == 1:
        emb = F.pad(emb, (0, 1, 0, 0))
    return emb


class Timesteps(nn.Module):
    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift):
        super().__init__()
        self.num_channels = num_channels
        self.flip_sin_to_cos = flip_sin_to_cos
        self.downscale_freq_shift = downscale_freq_shift

    def forward(self, timesteps):
        return get_timestep_embedding(timesteps, self.num_channels, flip_sin_to_cos=self.flip_sin_to_cos,
                                      downscale_freq_shift=self.downscale_freq_shift)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim, act_fn='silu', out_dim=None):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, out_dim or time_embed_dim)

    def forward(self, sample, condition=None):
        return self.linear_2(self.act(self.linear_1(sample)))


# ------------------------------------------------------------------------------------------------
# attention — attention.py:174-194; motion_module.py:202-211; processors edlora_util.py:13-82,
# attention_register.py:15-173
# ------------------------------------------------------------------------------------------------
class AttnProcessor:
    """Materialised-probabilities processor (diffusers AttnProcessor)."""

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None):
        residual = hidden_states
        batch_size, sequence_length, _ = (hidden_states.shape if encoder_hidden_states is None
                                          else encoder_hidden_states.shape)
        attention_mask = attn.prepare_attention_mask(attention_mask, sequence_length, batch_size)
        query = attn.to_q(hidden_states)
        if encoder_hidden_states is None:
            encoder_hidden_states = hidden_states
        key = attn.to_k(encoder_hidden_states)
        value = attn.to_v(encoder_hidden_states)
        query = attn.head_to_batch_dim(query)
        key = attn.head_to_batch_dim(key)
        value = attn.head_to_batch_dim(value)
        probs = attn.get_attention_scores(query, key, attention_mask)
        hidden_states = torch.bmm(probs, value)
        hidden_states = attn.batch_to_head_dim(hidden_states)
        hidden_states = attn.to_out[0](hidden_states)
        hidden_states = attn.to_out[1](hidden_states)
        if attn.residual_connection:
            hidden_states = hidden_states + residual
        return hidden_states / attn.rescale_output_factor


class AttnProcessor2_0:
    """F.scaled_dot_product_attention processor: the default of every spatial Attention in the reference."""

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None):
        residual = hidden_states
        batch_size = hidden_states.shape[0]
        inner_dim = hidden_states.shape[-1]
        query = attn.to_q(hidden_states)
        if encoder_hidden_states is None:
            encoder_hidden_states = hidden_states
        key = attn.to_k(encoder_hidden_states)
        value = attn.to_v(encoder_hidden_states)
        head_dim = inner_dim // attn.heads
        query = query.view(batch_size, -1, attn.heads, head_dim).transpose(1, 2)
        key = key.view(batch_size, -1, attn.heads, head_dim).transpose(1, 2)
        value = value.view(batch_size, -1, attn.heads, head_dim).transpose(1, 2)
        hidden_states = F.scaled_dot_product_attention(query, key, value, attn_mask=attention_mask, dropout_p=0.0,
                                                       is_causal=False)
        hidden_states = hidden_states.transpose(1, 2).reshape(batch_size, -1, attn.heads * head_dim)
        hidden_states = hidden_states.to(query.dtype)
        hidden_states = attn.to_out[0](hidden_states)
        hidden_states = attn.to_out[1](hidden_states)
        if attn.residual_connection:
            hidden_states = hidden_states + residual
        return hidden_states / attn.rescale_output_factor


class XFormersAttnProcessor(AttnProcessor2_0):
    """xformers is absent here; numerically the same attention (kept for isinstance checks,
    attention_register.py:190)."""

    def __init__(self, attention_op=None):
        self.attention_op = attention_op


class Attention(nn.Module):
    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, dropout=0.0, bias=False,
                 upcast_attention=False, upcast_softmax=False, out_bias=True, scale_qk=True,
                 only_cross_attention=False, rescale_output_factor=1.0, residual_connection=False, processor=None):
        super().__init__()
        inner_dim = dim_head * heads
        cross_attention_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.upcast_attention = upcast_attention
        self.upcast_softmax = upcast_softmax
        self.rescale_output_factor = rescale_output_factor
        self.residual_connection = residual_connection
        self.dropout = dropout
        self.scale_qk = scale_qk
        self.scale = dim_head ** -0.5 if scale_qk else 1.0
        self.heads = heads
        self.sliceable_head_dim = heads
        self.added_kv_proj_dim = None
        self.only_cross_attention = only_cross_attention
        self.group_norm = None
        self.spatial_norm = None
        self.norm_cross = None
        self.to_q = nn.Linear(query_dim, inner_dim, bias=bias)
        self.to_k = nn.Linear(cross_attention_dim, inner_dim, bias=bias)
        self.to_v = nn.Linear(cross_attention_dim, inner_dim, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner_dim, query_dim, bias=out_bias), nn.Dropout(dropout)])
        if processor is None:
            processor = AttnProcessor2_0()
        self.set_processor(processor)

    def set_processor(self, processor):
        if (hasattr(self, 'processor') and isinstance(self.processor, nn.Module)
                and not isinstance(processor, nn.Module)):
            self._modules.pop('processor')
        self.processor = processor

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **cross_attention_kwargs):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask, **cross_attention_kwargs)

    def batch_to_head_dim(self, tensor):
        h = self.heads
        b, s, d = tensor.shape
        tensor = tensor.reshape(b // h, h, s, d)
        return tensor.permute(0, 2, 1, 3).reshape(b // h, s, d * h)

    def head_to_batch_dim(self, tensor, out_dim=3):
        h = self.heads
        b, s, d = tensor.shape
        tensor = tensor.reshape(b, s, h, d // h).permute(0, 2, 1, 3)
        if out_dim == 3:
            tensor = tensor.reshape(b * h, s, d // h)
        return tensor

    def get_attention_scores(self, query, key, attention_mask=None):
        dtype = query.dtype
        if self.upcast_attention:
            query, key = query.float(), key.float()
        if attention_mask is None:
            base = torch.empty(query.shape[0], query.shape[1], key.shape[1], dtype=query.dtype, device=query.device)
            beta = 0
        else:
            base, beta = attention_mask, 1
        scores = torch.baddbmm(base, query, key.transpose(-1, -2), beta=beta, alpha=self.scale)
        if self.upcast_softmax:
            scores = scores.float()
        return scores.softmax(dim=-1).to(dtype)

    def prepare_attention_mask(self, attention_mask, target_length, batch_size=None, out_dim=3):
        if attention_mask is None:
            return None
        raise NotImplementedError('attention masks are never used on the VideoSwap path')


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, hidden_states):
        hidden_states, gate = self.proj(hidden_states).chunk(2, dim=-1)
        return hidden_states * F.gelu(gate)


class FeedForward(nn.Module):
    """attention.py:204; motion_module.py:218 (activation_fn='geglu', mult 4)."""

    def __init__(self, dim, dim_out=None, mult=4, dropout=0.0, activation_fn='geglu', final_dropout=False):
        super().__init__()
        assert activation_fn == 'geglu'
        inner_dim = int(dim * mult)
        self.net = nn.ModuleList([GEGLU(dim, inner_dim), nn.Dropout(dropout), nn.Linear(inner_dim, dim_out or dim)])

    def forward(self, hidden_states):
        for m in self.net:
            hidden_states = m(hidden_states)
        return hidden_states


class AdaLayerNorm(nn.Module):  # imported by attention.py:11, never instantiated (num_embeds_ada_norm=None)
    def __init__(self, *a, **k):
        raise NotImplementedError


# ------------------------------------------------------------------------------------------------
# schedulers — test.py:77; pipeline_videoswap.py:163,503-504,559,587,667-668,680,696
# ------------------------------------------------------------------------------------------------
@dataclass
class SchedulerOutput(BaseOutput):
    prev_sample: torch.Tensor = None
    pred_original_sample: torch.Tensor = None


SD15_SCHEDULER_CONFIG = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
                             beta_schedule='scaled_linear', clip_sample=False, set_alpha_to_one=False,
                             steps_offset=1, prediction_type='epsilon', timestep_spacing='leading')


def _alphas_cumprod(num_train_timesteps, beta_start, beta_end, beta_schedule):
    if beta_schedule == 'scaled_linear':
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    elif beta_schedule == 'linear':
        betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
    else:
        raise NotImplementedError(beta_schedule)
    return torch.cumprod(1.0 - betas, dim=0)


class DDIMScheduler:
    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule='linear',
                 clip_sample=True, set_alpha_to_one=True, steps_offset=0, prediction_type='epsilon',
                 timestep_spacing='leading', **unused):
        self.config = FrozenDict(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                 beta_schedule=beta_schedule, clip_sample=clip_sample,
                                 set_alpha_to_one=set_alpha_to_one, steps_offset=steps_offset,
                                 prediction_type=prediction_type, timestep_spacing=timestep_spacing)
        assert prediction_type == 'epsilon' and timestep_spacing == 'leading'
        self.alphas_cumprod = _alphas_cumprod(num_train_timesteps, beta_start, beta_end, beta_schedule)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    @classmethod
    def from_config(cls, config, **kw):
        return cls(**{**dict(config), **kw})

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        step_ratio = self.config.num_train_timesteps // num_inference_steps
        t = (np.arange(0, num_inference_steps) * step_ratio).round()[::-1].copy().astype(np.int64)
        t += self.config.steps_offset
        self.timesteps = torch.from_numpy(t).to(device)

    def step(self, model_output, timestep, sample, eta=0.0, use_clipped_model_output=False, generator=None,
             variance_noise=None, return_dict=True):
        assert eta == 0.0, 'VideoSwap samples with eta = 0 (pipeline_videoswap.py:438)'
        timestep = int(timestep)
        prev_timestep = timestep - self.config.num_train_timesteps // self.num_inference_steps
        alpha_prod_t = self.alphas_cumprod[timestep]
        alpha_prod_t_prev = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        beta_prod_t = 1 - alpha_prod_t
        pred_original_sample = (sample - beta_prod_t ** 0.5 * model_output) / alpha_prod_t ** 0.5
        pred_epsilon = model_output
        pred_sample_direction = (1 - alpha_prod_t_prev) ** 0.5 * pred_epsilon
        prev_sample = alpha_prod_t_prev ** 0.5 * pred_original_sample + pred_sample_direction
        if not return_dict:
            return (prev_sample,)
        return SchedulerOutput(prev_sample=prev_sample, pred_original_sample=pred_original_sample)

    def coefficients(self, timestep):
        """(alpha_bar_t, alpha_bar_prev) of one step — what the fused GPU update consumes."""
        timestep = int(timestep)
        prev_timestep = timestep - self.config.num_train_timesteps // self.num_inference_steps
        a_prev = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        return float(self.alphas_cumprod[timestep]), float(a_prev)


class DDIMInverseScheduler:
    """diffusers 0.19.x DDIMInverseScheduler (UNPINNED — see module docstring and DESIGN.md).

    Chosen variant = the 0.17-0.19 one: `set_timesteps` builds the leading-spaced ascending grid
    1, 21, ..., 981 (steps_offset 1), rolls it by one and sets timesteps[0] = timesteps[1] - step_ratio, giving
    -19, 1, 21, ..., 961; `step` moves x_t -> x_{t + step_ratio} with alpha_bar_t = `initial_alpha_cumprod` for the
    negative first timestep (1 when `set_alpha_to_one`, else alphas_cumprod[0]; SD-1.5: 0.99915).  The last step therefore lands on t = 981, exactly the first timestep of
    the DDIM sampler that consumes the inverted latents (pipeline_videoswap.py:503-518).
    """
    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule='linear',
                 clip_sample=True, set_alpha_to_one=True, steps_offset=0, prediction_type='epsilon',
                 timestep_spacing='leading', **unused):
        self.config = FrozenDict(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                 beta_schedule=beta_schedule, clip_sample=clip_sample,
                                 set_alpha_to_one=set_alpha_to_one, steps_offset=steps_offset,
                                 prediction_type=prediction_type, timestep_spacing=timestep_spacing)
        assert prediction_type == 'epsilon' and timestep_spacing == 'leading'
        self.alphas_cumprod = _alphas_cumprod(num_train_timesteps, beta_start, beta_end, beta_schedule)
        # diffusers 0.18-0.19: `initial_alpha_cumprod = 1.0 if set_alpha_to_one else alphas_cumprod[0]`
        self.initial_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps).copy().astype(np.int64))

    @classmethod
    def from_config(cls, config, **kw):
        return cls(**{**dict(config), **kw})

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        step_ratio = self.config.num_train_timesteps // num_inference_steps
        t = (np.arange(0, num_inference_steps) * step_ratio).round().copy().astype(np.int64)
        t += self.config.steps_offset
        t = np.roll(t, 1)
        t[0] = int(t[1] - step_ratio)
        self.timesteps = torch.from_numpy(t).to(device)

    def coefficients(self, timestep):
        timestep = int(timestep)
        nxt = timestep + self.config.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[timestep] if timestep >= 0 else self.initial_alpha_cumprod
        return float(a_t), float(self.alphas_cumprod[nxt])

    def step(self, model_output, timestep, sample, eta=0.0, use_clipped_model_output=False, variance_noise=None,
             return_dict=True):
        timestep = int(timestep)
        prev_timestep = timestep + self.config.num_train_timesteps // self.num_inference_steps
        alpha_prod_t = self.alphas_cumprod[timestep] if timestep >= 0 else self.initial_alpha_cumprod
        alpha_prod_t_prev = self.alphas_cumprod[prev_timestep]
        beta_prod_t = 1 - alpha_prod_t
        pred_original_sample = (sample - beta_prod_t ** 0.5 * model_output) / alpha_prod_t ** 0.5
        pred_epsilon = model_output
        pred_sample_direction = (1 - alpha_prod_t_prev) ** 0.5 * pred_epsilon
        prev_sample = alpha_prod_t_prev ** 0.5 * pred_original_sample + pred_sample_direction
        if not return_dict:
            return (prev_sample, pred_original_sample)
        return SchedulerOutput(prev_sample=prev_sample, pred_original_sample=pred_original_sample)
