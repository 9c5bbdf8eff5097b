"""HIP-backed building blocks with the state-dict layout of their torch.nn counterparts.

Parameters keep the PyTorch/diffusers names and shapes (checkpoints, LoRA merges and `load_state_dict` round
trips work unchanged: pipeline_videoswap.py:303-305,417-420; convert_edlora_to_diffusers.py:46-53); `forward`
hands them to the libvsx kernels.  Activations are channels-last fp16: images [B*F, H, W, C], tokens [.., C].
"""
import torch
from torch import nn

from . import ops


def param_key(*params):
    """Identity of a set of parameters for host-side caches: a LoRA merge / load_state_dict bumps `_version`,
    .to()/.half() replace the storage."""
    return tuple((p.data_ptr(), p._version, p.dtype) for p in params if p is not None)


class StepInvariantCache:
    """Results that depend only on an input TENSOR OBJECT (kept alive here, so its address cannot be recycled) and on
    parameters: the text K/V projections and the time-embedding projections are identical in every denoising step
    that passes the same embedding, so they are computed once per (input, parameters) instead of 100 x per clip —
    about 70 launch-latency-bound M <= 154 GEMMs per UNet call."""

    # graphs.GraphCache sets this while it warms up and captures a forward: the graph's static input buffers are long-lived
    # tensor OBJECTS whose contents change between replays, so a result cached on them would freeze the first call's
    # projections into every replay (the projections must be kernels of the graph instead)
    bypass = False

    def __init__(self, limit=256):
        self.limit = limit
        self.entries = {}

    def get(self, src, extra, params, fn):
        if StepInvariantCache.bypass:
            return fn()
        key = (id(src), extra)
        hit = self.entries.get(key)
        stamp = (src._version, param_key(*params))
        if hit is not None and hit[0] is src and hit[1] == stamp:
            return hit[2]
        val = fn()
        if len(self.entries) >= self.limit:
            self.entries.clear()
        self.entries[key] = (src, stamp, val)
        return val

    def put(self, src, extra, params, val):
        """hand in a result computed elsewhere (AnimateDiffUNet3DModel.prepare_timesteps: all timesteps of a loop in one GEMM)"""
        if len(self.entries) >= self.limit:
            self.entries.clear()
        self.entries[(id(src), extra)] = (src, (src._version, param_key(*params)), val)


class Linear(nn.Linear):
    """y = x W^T + b on the MFMA GEMM; optional fused residual add.  `row_stats`: a LayerNorm consumes the result — the
    GEMM's epilogue may then emit the row statistics with it (ops.linear)."""

    def forward(self, x, residual=None, row_stats=False):
        return ops.linear(x, self.weight, self.bias, residual=residual, row_stats=row_stats)


class PointwiseConv(nn.Conv2d):
    """1x1 nn.Conv2d parameters ([out, in, 1, 1]) applied to channels-last tokens (attention.py:65,93)."""

    def forward(self, x, residual=None, row_stats=False):
        return ops.linear(x, self.weight, self.bias, residual=residual, row_stats=row_stats)


class Identity(nn.Module):
    """nn.Dropout(p=0) stand-in (to_out[1], ff.net[1]); kept so module indices / state-dict keys match."""

    def forward(self, x):
        return x


class _PackedWeight:
    """OHWI fp16 copy of a conv weight, rebuilt when the parameter changes (LoRA merge / load_state_dict bump
    `_version`; .to()/.half() replace the storage)."""

    def __init__(self):
        self.key = None
        self.value = None

    def get(self, weight, cin_pad=0):
        key = (weight.data_ptr(), weight._version, weight.dtype, weight.device, cin_pad)
        if key != self.key:
            w = weight.detach().permute(0, 2, 3, 1)
            if cin_pad and cin_pad != w.shape[-1]:
                w = torch.nn.functional.pad(w, (0, cin_pad - w.shape[-1]))
            self.value = w.contiguous()
            self.key = key
        return self.value


class InflatedConv3d(nn.Conv2d):
    """Per-frame 2-D convolution of a video tensor (resnet.py:9-18), as an implicit GEMM on the channels-last
    [B*F, H, W, C] activation: no '(b f)' fold copies.  The weight parameter keeps nn.Conv2d's [O, I, kh, kw]
    shape but is stored in channels_last memory format, i.e. physically OHWI — exactly the K-major B operand the
    kernel streams — so no packed copy is needed in the common case."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.weight.data = self.weight.data.contiguous(memory_format=torch.channels_last)
        self._packed = _PackedWeight()
        ks = self.kernel_size[0]
        if self.kernel_size != (ks, ks) or ks not in (1, 3) or self.padding not in ((ks // 2, ks // 2),):
            raise NotImplementedError(f'InflatedConv3d: kernel {self.kernel_size} padding {self.padding}')
        if self.stride not in ((1, 1), (2, 2)):
            raise NotImplementedError(f'InflatedConv3d: stride {self.stride}')

    def ohwi(self, cin_pad=0):
        w = self.weight
        if not cin_pad and w.is_contiguous(memory_format=torch.channels_last):
            # zero-copy view, contiguous as [O, kh, kw, I].  The SAME view object is handed out while the parameter's storage
            # stays in place: caches keyed on the tensor object (the flipped dgrad copies of videoswap_amd/autograd.py) would
            # otherwise miss on every call — the round-3 training step rebuilt ~1 GB of them per step and kept the old ones
            key = (w.data_ptr(), w.dtype, w.device)
            hit = self.__dict__.get('_ohwi_view')
            if hit is None or hit[0] != key:
                hit = self.__dict__['_ohwi_view'] = (key, w.detach().permute(0, 2, 3, 1))
            return hit[1]
        return self._packed.get(w, cin_pad)

    def forward(self, x, x2=None, upsample=False, rowvec=None, rows_per_vec=0, residual=None):
        cin = x.shape[-1] + (x2.shape[-1] if x2 is not None else 0)
        w = self.ohwi(cin if cin != self.in_channels else 0)    # conv_in: 4 latent channels padded to 8
        return ops.conv2d(x, w, self.bias, x2=x2, stride=self.stride[0], upsample=upsample, rowvec=rowvec,
                          rows_per_vec=rows_per_vec, residual=residual)


class GroupNorm(nn.GroupNorm):
    """GroupNorm on channels-last data.  `nimg` selects the statistics scope: B (5-D GroupNorm of the reference,
    pooled over frames: resnet.py:166,177; unet.py:474) or B*F (per frame: attention.py:108; motion_module.py:146)."""

    def forward(self, x, nimg, silu=False, x2=None, partial_hook=None, count_rows=None):
        return ops.group_norm(x, self.weight, self.bias, self.num_groups, self.eps, nimg, silu=silu, x2=x2,
                              partial_hook=partial_hook, count_rows=count_rows)


class LayerNorm(nn.LayerNorm):
    def forward(self, x, pe=None, rows_per_frame=0, frames=0, frame_offset=0, defer=False):
        """`defer=True` (the caller knows that every consumer is a Linear of this package: native attention processors,
        FeedForward) returns an `ops.DeferredLN`: the normalisation is folded into the consuming GEMMs and the normalised
        tensor is never written.  Inference on the GPU only (the gradient path and the CPU test emulation take the
        kernel)."""
        if defer and ops.LN_FUSE and x.is_cuda and not (torch.is_grad_enabled() and x.requires_grad):
            if pe is not None and (pe.dim() != 2 or pe.shape[1] != x.shape[-1] or frames <= 0 or rows_per_frame <= 0
                                   or frame_offset < 0 or pe.shape[0] < frame_offset + frames):
                return ops.layer_norm(x, self.weight, self.bias, self.eps, pe=pe, rows_per_frame=rows_per_frame,
                                      frames=frames, frame_offset=frame_offset)        # raises the table-size error
            return ops.DeferredLN(x, self.weight, self.bias, self.eps, pe, rows_per_frame, frames, frame_offset)
        return ops.layer_norm(x, self.weight, self.bias, self.eps, pe=pe, rows_per_frame=rows_per_frame,
                              frames=frames, frame_offset=frame_offset)


class GEGLU(nn.Module):
    """diffusers GEGLU: proj to 2*dim_out, h * gelu(g) — fused into the GEMM epilogue."""

    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = Linear(dim_in, dim_out * 2)

    def forward(self, x):
        return ops.linear(x, self.proj.weight, self.proj.bias, geglu=True)


class FeedForward(nn.Module):
    """diffusers FeedForward(dim, activation_fn='geglu') (attention.py:204; motion_module.py:218):
    net = [GEGLU(dim, 4 dim), Dropout(0), Linear(4 dim, dim)]."""

    def __init__(self, dim, dim_out=None, mult=4, dropout=0.0, activation_fn='geglu', final_dropout=False):
        super().__init__()
        if activation_fn != 'geglu' or dropout != 0.0:
            raise NotImplementedError('FeedForward: only geglu / dropout 0 (the VideoSwap configuration)')
        inner = int(dim * mult)
        self.net = nn.ModuleList([GEGLU(dim, inner), Identity(), Linear(inner, dim_out or dim)])

    def forward(self, x, residual=None):
        return self.net[2](self.net[0](x), residual=residual)
