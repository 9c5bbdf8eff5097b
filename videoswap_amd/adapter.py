"""SparsePointAdapter (videoswap/models/adapter_model.py:50-136) on the libvsx kernels.

Per level l: feat = MLP_l(point_embedding) [P, C_l] (two GEMMs + SiLU), then every visible point of every
frame is splatted onto the 4 bilinear corners of the level's [F, h_l, w_l, C_l] map by ONE scatter kernel per
level (the reference runs levels x points x frames python iterations with 4 tiny index-adds each).  Maps are
returned channels-last and tagged `.vsx_nhwc = True` so the UNet adds them without a layout conversion;
`.to_reference_layout()` gives the reference's [F, C, h, w].
"""
from typing import List

import torch
from torch import nn

from . import ops
from .compat import MODEL_REGISTRY, ConfigMixin, ModelMixin, register_to_config
from .layers import Linear


class _SiLU(nn.Module):
    def forward(self, x):
        return ops.silu(x)


class MLP(nn.Module):
    """adapter_model.py:12-22 — nn.Sequential(Linear, SiLU, Linear) under the attribute name `mlp`."""

    def __init__(self, in_dim, out_dim, mid_dim=128):
        super().__init__()
        self.mlp = nn.Sequential(Linear(in_dim, mid_dim, bias=True), _SiLU(), Linear(mid_dim, out_dim, bias=True))

    def forward(self, x):
        return self.mlp(x)


def _tag_nhwc(t):
    t.vsx_nhwc = True
    return t


@MODEL_REGISTRY.register()
class SparsePointAdapter(ModelMixin, ConfigMixin):

    @register_to_config
    def __init__(self, embedding_channels=1280, channels=[320, 640, 1280, 1280], downsample_rate=[8, 16, 32, 64],
                 mid_dim=128):
        super().__init__()
        self.model_list = nn.ModuleList([MLP(embedding_channels, ch, mid_dim) for ch in channels])
        self.downsample_rate = list(downsample_rate)
        self.channels = list(channels)
        self.radius = 2

    def generate_loss_mask(self, point_index_list, point_tracker, num_frames, h, w, loss_type):
        """adapter_model.py:70-95, on the host (a [F, 4, h/8, w/8] map of zeros and ones).  As in the reference the
        box of a visible (point, frame) pair is written into EVERY frame of the mask and spans [y1, y2) x [x1, x2)."""
        rate = self.downsample_rate[0]
        if loss_type == 'global':
            return torch.ones(num_frames, 4, h // rate, w // rate)
        mask = torch.zeros(num_frames, 4, h // rate, w // rate)
        tracks = point_tracker.detach().float().cpu()
        for p in point_index_list:
            for f in range(num_frames):
                px, py = float(tracks[f, p, 0]), float(tracks[f, p, 1])
                if px < 0 or py < 0:
                    continue
                px, py = px / rate, py / rate
                x1, y1, x2, y2 = int(px) - self.radius, int(py) - self.radius, int(px) + self.radius, int(py) + self.radius
                x1, x2 = max(min(x1, mask.shape[3] - 1), 0), max(min(x2, mask.shape[3] - 1), 0)
                y1, y2 = max(min(y1, mask.shape[2] - 1), 0), max(min(y2, mask.shape[2] - 1), 0)
                mask[:, :, y1:y2, x1:x2] = 1.0
        return mask

    @staticmethod
    def _features(module, emb):
        """MLP(point_embedding) in the activation dtype.  In training the adapter keeps fp32 master weights (accelerate
        mixed precision, train.py:135-144): they are cast per call and the gradient flows back through the cast."""
        lin1, lin2 = module.mlp[0], module.mlp[2]
        dt = emb.dtype
        x = ops.linear(emb, lin1.weight.to(dt), lin1.bias.to(dt))
        return ops.linear(ops.silu(x), lin2.weight.to(dt), lin2.bias.to(dt))

    def forward(self, point_tracker, size, point_embedding, index_list=None, drop_rate=0.0,
                loss_type='global', scale=1.0):
        """point_tracker [1?, F, P, 2] pixel (x, y), negative = invisible; size = (W, H); point_embedding
        [1?, P, 1280]; index_list: point ids to keep (None = all).  Returns 4 channels-last maps [F, h, w, C]; in
        training mode (adapter_model.py:105-107,133-134) points are dropped at random and the loss mask is returned
        as well: `(maps, loss_mask)`."""
        import random
        tracks = point_tracker.squeeze(0) if point_tracker.dim() == 4 else point_tracker
        emb = point_embedding.squeeze(0) if point_embedding.dim() == 3 else point_embedding
        w, h = int(size[0]), int(size[1])            # a DataLoader hands over one-element tensors
        num_frames, num_points = tracks.shape[:2]
        loss_mask = None
        if self.training:
            keep = [p for p in range(num_points) if random.random() > drop_rate]
            loss_mask = self.generate_loss_mask(keep, tracks, num_frames, h, w, loss_type)
        else:
            keep = [p for p in range(num_points) if index_list is None or p in index_list]
        selected = torch.zeros(num_points, dtype=torch.int32)
        selected[keep] = 1
        with torch.set_grad_enabled(self.training):
            emb = emb.to(torch.float16).contiguous()                        # the path's activation dtype
            selected = selected.to(emb.device)
            # the reference holds the tracks in the latent dtype (fp16): quantise, then hand fp32 to the kernel
            tracks32 = tracks.to(device=emb.device).to(emb.dtype).float().contiguous()
            out = []
            for level, module in enumerate(self.model_list):
                rate = self.downsample_rate[level]
                feat = self._features(module, emb)                           # [P, C_l]
                state = ops.adapter_scatter(tracks32, selected, feat, h // rate, w // rate, float(rate), float(scale))
                out.append(_tag_nhwc(state))
        return (out, loss_mask) if self.training else out

    @staticmethod
    def to_reference_layout(states):
        return [s.permute(0, 3, 1, 2).contiguous() for s in states]
