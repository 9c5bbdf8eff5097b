"""Thin in-repo facade for the third-party symbols the reference path imports from diffusers 0.19.3
(SURVEY.md §8b): config/model mixins, BaseOutput, DDIM schedulers, and the plugin registries of
videoswap/utils/registry.py:79-82.  No arithmetic lives here except the schedulers' scalar coefficient tables
(the per-element DDIM update itself runs in the vsx_cfg_ddim_step HIP kernel).
"""
import functools
import inspect
from collections import OrderedDict
from dataclasses import fields

import numpy as np
import torch
from torch import nn


class Registry:
    """name -> class lookup; same interface as videoswap/utils/registry.py:4-76."""

    def __init__(self, name):
        self._name = name
        self._obj_map = {}

    def register(self, obj=None):
        def add(o):
            if o.__name__ in self._obj_map:
                raise AssertionError(f"An object named '{o.__name__}' was already registered in '{self._name}' registry!")
            self._obj_map[o.__name__] = o
            return o
        if obj is None:
            return add
        add(obj)

    def get(self, name):
        if name not in self._obj_map:
            raise KeyError(f"No object named '{name}' found in '{self._name}' registry!")
        return self._obj_map[name]

    def __contains__(self, name):
        return name in self._obj_map

    def __iter__(self):
        return iter(self._obj_map.items())

    def keys(self):
        return self._obj_map.keys()


PIPELINE_REGISTRY = Registry('pipelines')
MODEL_REGISTRY = Registry('models')


class FrozenDict(OrderedDict):
    def __getattr__(self, name):
        if name in self:
            return self[name]
        raise AttributeError(name)


def register_to_config(init):
    """Record the constructor arguments (defaults included) as `self.config` (diffusers ConfigMixin protocol)."""
    sig = inspect.signature(init)
    names = [n for n in sig.parameters if n != 'self']

    @functools.wraps(init)
    def wrapped(self, *args, **kwargs):
        cfg = {n: sig.parameters[n].default for n in names
               if sig.parameters[n].default is not inspect.Parameter.empty}
        cfg.update(dict(zip(names, args)))
        cfg.update(kwargs)
        self._internal_dict = FrozenDict(cfg)
        init(self, *args, **kwargs)
    return wrapped


class ConfigMixin:
    @property
    def config(self):
        return self._internal_dict

    @classmethod
    def from_config(cls, config, **kwargs):
        accepted = set(inspect.signature(cls.__init__).parameters) - {'self'}
        merged = {k: v for k, v in dict(config).items() if k in accepted}
        merged.update({k: v for k, v in kwargs.items() if k in accepted})
        return cls(**merged)


class ModelMixin(nn.Module):
    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device


class BaseOutput(OrderedDict):
    """Dataclass-style output that also unpacks/indexes like a tuple."""

    def __post_init__(self):
        for f in fields(self):
            value = getattr(self, f.name)
            if value is not None:
                self[f.name] = value

    def __getitem__(self, key):
        if isinstance(key, str):
            return OrderedDict.__getitem__(self, key)
        return tuple(self.values())[key]

    def to_tuple(self):
        return tuple(self.values())


# ------------------------------------------------------------------------------------------------
# DDIM schedulers (diffusers 0.19.x semantics; see DESIGN.md for the inverse-scheduler variant)
# ------------------------------------------------------------------------------------------------
SD15_SCHEDULER_CONFIG = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
                             beta_schedule='scaled_linear', clip_sample=False, set_alpha_to_one=False, steps_offset=1,
                             prediction_type='epsilon', timestep_spacing='leading')


class _SchedulerOutput(tuple):
    @property
    def prev_sample(self):
        return self[0]


class _DDIMBase:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule='linear',
                 clip_sample=True, set_alpha_to_one=True, steps_offset=0, prediction_type='epsilon',
                 timestep_spacing='leading', **ignored):
        if prediction_type != 'epsilon' or timestep_spacing != 'leading' or clip_sample:
            raise NotImplementedError('only epsilon prediction / leading spacing / no clipping (the SD-1.5 config)')
        self.config = FrozenDict(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                 beta_schedule=beta_schedule, clip_sample=clip_sample,
                                 set_alpha_to_one=set_alpha_to_one, steps_offset=steps_offset,
                                 prediction_type=prediction_type, timestep_spacing=timestep_spacing)
        if beta_schedule == 'scaled_linear':
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        elif beta_schedule == 'linear':
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        else:
            raise NotImplementedError(beta_schedule)
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.num_inference_steps = None
        self.timesteps = None

    @classmethod
    def from_config(cls, config, **kwargs):
        return cls(**{**dict(config), **kwargs})

    @classmethod
    def from_pretrained(cls, pretrained_model_path, subfolder=None, **kwargs):
        """`<path>/<subfolder>/scheduler_config.json` (test.py:77)"""
        from .formats import scheduler_config_from_pretrained
        return cls(**{**scheduler_config_from_pretrained(pretrained_model_path, subfolder), **kwargs})

    def scale_model_input(self, sample, timestep=None):
        return sample

    def _ratio(self):
        return self.config.num_train_timesteps // self.num_inference_steps

    def step(self, model_output, timestep, sample, eta=0.0, generator=None, return_dict=True, **ignored):
        """x_t -> x_next with the fused HIP update (guidance already applied to model_output)."""
        if eta != 0.0:
            raise NotImplementedError('eta != 0 (VideoSwap samples deterministically)')
        from . import ops
        a_t, a_n = self.coefficients(timestep)
        out = ops.cfg_ddim_step(sample.contiguous(), model_output.contiguous(), None, 1.0, a_t, a_n)
        return _SchedulerOutput((out,))


class DDIMScheduler(_DDIMBase):
    """Sampling direction: timesteps 981, 961, ..., 1 for 50 steps."""

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        t = (np.arange(0, num_inference_steps) * self._ratio()).round()[::-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(t + self.config.steps_offset)   # kept on the host: scalars only

    def coefficients(self, timestep):
        t = int(timestep)
        prev = t - self._ratio()
        final = 1.0 if self.config.set_alpha_to_one else float(self.alphas_cumprod[0])
        return float(self.alphas_cumprod[t]), (float(self.alphas_cumprod[prev]) if prev >= 0 else final)


class DDIMInverseScheduler(_DDIMBase):
    """Inversion direction, diffusers 0.18-0.19 variant: timesteps -19, 1, 21, ..., 961; each step moves
    x_t -> x_{t+20}; alpha_bar of the negative first timestep is `initial_alpha_cumprod` = 1 when
    `set_alpha_to_one`, else alphas_cumprod[0] (the SD-1.5 config: 0.99915)."""

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        t = (np.arange(0, num_inference_steps) * self._ratio()).round().copy().astype(np.int64)
        t = np.roll(t + self.config.steps_offset, 1)
        t[0] = int(t[1] - self._ratio()) if num_inference_steps > 1 else int(t[0] - self._ratio())
        self.timesteps = torch.from_numpy(t)

    def coefficients(self, timestep):
        t = int(timestep)
        initial = 1.0 if self.config.set_alpha_to_one else float(self.alphas_cumprod[0])
        a_t = float(self.alphas_cumprod[t]) if t >= 0 else initial
        return a_t, float(self.alphas_cumprod[t + self._ratio()])


class DDPMScheduler:
    """What the training step uses of diffusers' DDPMScheduler (train.py:157-160; trainer_videoswap.py:57,82-90): the
    noise schedule, `add_noise` (the forward diffusion q(x_t | x_0)) and `get_velocity`."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule='linear',
                 prediction_type='epsilon', **ignored):
        self.config = FrozenDict(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                 beta_schedule=beta_schedule, prediction_type=prediction_type)
        if beta_schedule == 'scaled_linear':
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        elif beta_schedule == 'linear':
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        else:
            raise NotImplementedError(beta_schedule)
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)

    @classmethod
    def from_config(cls, config, **kwargs):
        return cls(**{**dict(config), **kwargs})

    @classmethod
    def from_pretrained(cls, pretrained_model_path, subfolder=None, **kwargs):
        from .formats import scheduler_config_from_pretrained
        return cls(**{**scheduler_config_from_pretrained(pretrained_model_path, subfolder), **kwargs})

    def _coefficients(self, like, timesteps):
        a = self.alphas_cumprod.to(device=like.device, dtype=like.dtype)[timesteps.to(like.device)]
        shape = (-1,) + (1,) * (like.dim() - 1)
        return (a ** 0.5).view(shape), ((1 - a) ** 0.5).view(shape)

    def add_noise(self, original_samples, noise, timesteps):
        sa, s1a = self._coefficients(original_samples, timesteps)
        return sa * original_samples + s1a * noise

    def get_velocity(self, sample, noise, timesteps):
        sa, s1a = self._coefficients(sample, timesteps)
        return sa * noise - s1a * sample
