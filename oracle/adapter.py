"""TEST INFRASTRUCTURE — CPU restatement of SparsePointAdapter (videoswap/models/adapter_model.py:12-136), inference
branch only.  Checked against the reference class imported verbatim in tests/test_oracle.py.
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

    def forward(self, point_tracker, size, point_embedding, index_list=None):
        """adapter_model.py:97-136 (eval branch) -> 4 maps [F, C_l, h_l, w_l]"""
        tracks = point_tracker.squeeze(0)
        emb = point_embedding.squeeze(0)
        w, h = size
        frames, points = tracks.shape[:2]
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
        return out
