"""Data side of the drop-in (videoswap/data/frame_point_dataset.py, videoswap/data/transform.py): the single-video
dataset `test.py` builds from the YAML (`datasets.type: SingleVideoPointDataset`), on PIL + torch (torchvision is not
installed here).  It produces what the pipeline consumes either side of the denoising loop: the selected frames (PIL,
resized) and the point conditions {pred_tracks[F,P,2], point_embedding[P,1280], point_name2id, img_size=(W,H)}.
"""
from copy import deepcopy
from pathlib import Path

import numpy as np
import torch
from PIL import Image

from .compat import Registry
from .formats import load_tap

DATASET_REGISTRY = Registry('dataset')
TRANSFORM_REGISTRY = Registry('transform')


# ---- the torchvision transforms the YAMLs name (videoswap/data/transform.py:24-38), PIL in / tensor out ----
@TRANSFORM_REGISTRY.register()
class Resize:
    """torchvision.transforms.Resize on a PIL image: an int matches the SHORTER side (aspect kept), a pair is (h, w);
    bilinear with PIL's antialiasing (what torchvision does for PIL inputs)."""

    def __init__(self, size, interpolation=None, max_size=None, antialias=None):
        self.size = size

    def __call__(self, img):
        w, h = img.size
        if isinstance(self.size, int):
            short, long_ = (w, h) if w <= h else (h, w)
            new_short, new_long = self.size, int(self.size * long_ / short)
            nw, nh = (new_short, new_long) if w <= h else (new_long, new_short)
        else:
            nh, nw = self.size
        return img if (nw, nh) == (w, h) else img.resize((nw, nh), Image.BILINEAR)


@TRANSFORM_REGISTRY.register()
class CenterCrop:
    def __init__(self, size):
        self.size = (size, size) if isinstance(size, int) else tuple(size)

    def __call__(self, img):
        th, tw = self.size
        if torch.is_tensor(img):
            h, w = img.shape[-2:]
            top, left = int(round((h - th) / 2.0)), int(round((w - tw) / 2.0))
            return img[..., top:top + th, left:left + tw]
        w, h = img.size
        top, left = int(round((h - th) / 2.0)), int(round((w - tw) / 2.0))
        return img.crop((left, top, left + tw, top + th))


@TRANSFORM_REGISTRY.register()
class ToTensor:
    def __call__(self, pic):
        arr = np.asarray(pic, dtype=np.uint8)
        if arr.ndim == 2:
            arr = arr[:, :, None]
        return torch.from_numpy(arr.copy()).permute(2, 0, 1).float().div(255.0)


@TRANSFORM_REGISTRY.register()
class Normalize:
    def __init__(self, mean, std, inplace=False):
        self.mean, self.std = list(mean), list(std)

    def __call__(self, t):
        mean = torch.tensor(self.mean, dtype=t.dtype).view(-1, 1, 1)
        std = torch.tensor(self.std, dtype=t.dtype).view(-1, 1, 1)
        return (t - mean) / std


def build_transform(opt):
    opt = deepcopy(opt)
    return TRANSFORM_REGISTRY.get(opt.pop('type'))(**opt)


class Compose:
    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


def select_frame_idx(begin_frame_idx, end_frame_idx, n):
    """frame_point_dataset.py:13-22: n frames at a constant stride of total // (n - 1)."""
    interval = (end_frame_idx - begin_frame_idx) // (n - 1)
    return [int(begin_frame_idx + i * interval) for i in range(n)]


@DATASET_REGISTRY.register()
class SingleVideoPointDataset(torch.utils.data.Dataset):
    """frame_point_dataset.py:25-81"""

    def __init__(self, opt):
        self.opt = opt
        self.total_frames = sorted(Path(opt['path']).iterdir())
        self.select_id = select_frame_idx(0, min(len(self.total_frames), opt['total_frames']), opt['num_frames'])
        self.video = [self.total_frames[i] for i in self.select_id]
        self.prompt = opt['prompt']
        self.num_video = opt.get('dataset_enlarge_ratio', 1)
        self.video_transform = Compose(build_transform(t) for t in opt['video_transform'])
        frames = torch.stack([self.video_transform(Image.open(p).convert('RGB')) for p in self.video])
        self.frames = frames.permute(1, 0, 2, 3).contiguous()          # 'f c h w -> c f h w'
        self.size_y, self.size_x = self.frames.shape[-2:]
        self.condition = self.get_conditions(opt['tap_path']) if 'tap_path' in opt else None

    def __len__(self):
        return self.num_video

    def get_frames(self):
        pil_only = Compose(build_transform(t) for t in self.opt['video_transform']
                           if t['type'] not in ('ToTensor', 'Normalize'))
        return [pil_only(Image.open(p).convert('RGB')) for p in self.video]

    def get_conditions(self, tap_path=None):
        if tap_path is None:
            return self.condition
        tap = load_tap(tap_path)
        return {'pred_tracks': tap['pred_tracks'][self.select_id], 'point_embedding': tap['point_embedding'],
                'point_name2id': tap['point_name2id'], 'img_size': (self.size_x, self.size_y)}

    def __getitem__(self, index):
        batch = {'images': self.frames, 'prompt': self.prompt}
        if self.condition is not None:
            batch.update(self.condition)
        return batch


def build_dataset(dataset_type):
    """videoswap/data/__init__.py:24-32 — returns the CLASS (`build_dataset(t)(opt)`, test.py:99)."""
    return DATASET_REGISTRY.get(dataset_type)
