"""TEST INFRASTRUCTURE — CPU restatement of SparsePointAdapter (videoswap/models/adapter_model.py:12-136): the inference
branch and the training branch (random point dropout + loss mask).  Checked against the reference class imported
verbatim in tests/test_oracle.py and tests/test_training.py.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import anything under oracle/."""
import torch
from torch import nn


class MLP(nn.Module):
    """adapter_model.py:12-22"""

    def __init__(self, in_dim, out_dim, mid_dim=128):
        super().__init__()
        self.mlp = nn.Sequential(nn.Linear(in_dim, mid_dim, bias=True), nn.SiLU(), nn.Linear(mid_dim, out_dim, bias=True))

    def forward(self, x):
        return self.mlp(x)


def bilinear_splat(state, x, y, frame, value):
    """adapter_model.py:25-47: 4-corner bilinear accumulation; corners clamp to the map edge (and may coincide)."""
    x1, y1 = int(x), int(y)
    x2, y2 = x1 + 1, y1 + 1
    xf, yf = x - x1, y - y1
    w, h = state.shape[3], state.shape[2]
    x1, x2 = max(min(x1, w - 1), 0), max(min(x2, w - 1), 0)
    y1, y2 = max(min(y1, h - 1), 0), max(min(y2, h - 1), 0)
    state[frame, :, y1, x1] += value * ((1 - xf) * (1 - yf))
    state[frame, :, y1, x2] += value * (xf * (1 - yf))
    state[frame, :, y2, x1] += value * ((1 - xf) * yf)
    state[frame, :, y2, x2] += value * (xf * yf)
    return state


class SparsePointAdapter(nn.Module):
    def __init__(self, embedding_channels=1280, channels=(320, 640, 1280, 1280), downsample_rate=(8, 16, 32, 64),
                 mid_dim=128):
        super().__init__()
        self.model_list = nn.ModuleList([MLP(embedding_channels, ch, mid_dim) for ch in channels])
        self.downsample_rate = list(downsample_rate)
        self.channels = list(channels)
        self.radius = 2

    def generate_loss_mask(self, keep, tracks, frames, h, w, loss_type):
        """adapter_model.py:70-95: note `loss_mask[:, :, y1:y2, x1:x2]` — every frame, half-open box"""
        rate = self.downsample_rate[0]
        if loss_type == 'global':
            return torch.ones(frames, 4, h // rate, w // rate)
        mask = torch.zeros(frames, 4, h // rate, w // rate)
        for p in keep:
            for f in range(frames):
                px, py = tracks[f, p]
                if px < 0 or py < 0:
                    continue
                px, py = px / rate, py / rate
                x1, y1, x2, y2 = int(px) - self.radius, int(py) - self.radius, int(px) + self.radius, int(py) + self.radius
                x1, x2 = max(min(x1, mask.shape[3] - 1), 0), max(min(x2, mask.shape[3] - 1), 0)
                y1, y2 = max(min(y1, mask.shape[2] - 1), 0), max(min(y2, mask.shape[2] - 1), 0)
                mask[:, :, y1:y2, x1:x2] = 1.0
        return mask

    def forward(self, point_tracker, size, point_embedding, index_list=None, drop_rate=0.0, loss_type='global'):
        """adapter_model.py:97-136 -> 4 maps [F, C_l, h_l, w_l] (training: `(maps, loss_mask)`)"""
        import random
        tracks = point_tracker.squeeze(0)
        emb = point_embedding.squeeze(0)
        w, h = size
        frames, points = tracks.shape[:2]
        if self.training:
            keep = [p for p in range(points) if random.random() > drop_rate]
            loss_mask = self.generate_loss_mask(keep, tracks, frames, h, w, loss_type)
        else:
            keep = [p for p in range(points) if index_list is None or p in index_list]
        out = []
        for level, module in enumerate(self.model_list):
            rate = self.downsample_rate[level]
            feat = module(emb)
            state = torch.zeros(frames, self.channels[level], h // rate, w // rate, dtype=feat.dtype)
            for p in keep:
                for f in range(frames):
                    px, py = tracks[f, p]
                    if px < 0 or py < 0:
                        continue
                    state = bilinear_splat(state, px / rate, py / rate, f, feat[p])
            out.append(state)
        if self.training:
            return out, loss_mask
        return out
