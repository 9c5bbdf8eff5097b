"""On-disk formats either side of the denoising path — readers, validators and writers.

Nothing here does arithmetic; these are the containers real checkpoints arrive in, so that they can run the moment
they are available, plus writers that produce seeded synthetic files in the same layouts (no checkpoint exists
offline: tests, the drop-in run of the reference's `test.py` and BASELINE.json configs[0] use them).

| file | layout | reference |
|---|---|---|
| `TAP.pth` | dict `pred_tracks [F_total, P, 2]` (pixel x, y; negative = invisible), `point_embedding [P, 1280]`, `point_name2id {name: column}` | videoswap/data/frame_point_dataset.py:62-70 |
| `adapter.pth` | `SparsePointAdapter.state_dict()` (`model_list.{l}.mlp.{0,2}.{weight,bias}`) | test.py:69-71, adapter_model.py:12-22,50-71 |
| ED-LoRA `.pth` | `{'params': {'new_concept_embedding': {'<name>': [16, 768]}, 'unet': {...lora_down/up...}, 'text_encoder': {...}}}` | convert_edlora_to_diffusers.py:82-105 |
| motion module `.ckpt` | AnimateDiff state dict; `.pos_encoder` keys gain a `.processor` prefix on load | test.py:60-64 |
| SD `unet/` | `config.json` + `diffusion_pytorch_model.bin` (2-D SD UNet keys) | unet.py:483-523 |
| `scheduler/scheduler_config.json` | DDIMScheduler kwargs | test.py:77 |
| `new_concept_cfg.json` | `{concept: {concept_token_ids, concept_token_names}}` | test.py:83-87 |
"""
import json
import os

import torch

TAP_KEYS = ('pred_tracks', 'point_embedding', 'point_name2id')


class FormatError(ValueError):
    pass


def _load(path):
    """Every container read here (TAP, adapter, ED-LoRA, motion module, SD weights) is tensors / dicts / lists / strings:
    it loads under `weights_only=True`, which refuses to unpickle arbitrary objects.  VSX_UNSAFE_PICKLE=1 opts into the
    reference's behaviour (a bare torch.load, test.py:60-70) for a checkpoint that carries anything else."""
    if not os.path.isfile(path):
        raise FormatError(f'{path} does not exist')
    if os.environ.get('VSX_UNSAFE_PICKLE') == '1':
        return torch.load(path, map_location='cpu', weights_only=False)
    try:
        return torch.load(path, map_location='cpu', weights_only=True)
    except Exception as e:
        raise FormatError(f'{path}: not a plain tensor container ({type(e).__name__}: {str(e)[:200]}); '
                          f'VSX_UNSAFE_PICKLE=1 loads it with full unpickling') from e


def load_checkpoint(path):
    """A state dict / tensor container from disk (motion module .ckpt, adapter.pth)."""
    return _load(path)


# ------------------------------------------------------------------------------------------------
# TAP.pth (semantic point tracks)
# ------------------------------------------------------------------------------------------------
def load_tap(path):
    tap = _load(path)
    if not isinstance(tap, dict) or any(k not in tap for k in TAP_KEYS):
        raise FormatError(f'{path}: a TAP file is a dict with keys {TAP_KEYS}')
    tracks, emb, names = tap['pred_tracks'], tap['point_embedding'], tap['point_name2id']
    if tracks.dim() != 3 or tracks.shape[-1] != 2:
        raise FormatError(f'{path}: pred_tracks must be [frames, points, 2], got {tuple(tracks.shape)}')
    if emb.dim() != 2 or emb.shape[0] != tracks.shape[1]:
        raise FormatError(f'{path}: point_embedding {tuple(emb.shape)} does not match {tracks.shape[1]} points')
    if not all(0 <= int(i) < tracks.shape[1] for i in names.values()):
        raise FormatError(f'{path}: point_name2id refers to points outside [0, {tracks.shape[1]})')
    return tap


def save_tap(path, pred_tracks, point_embedding, point_name2id):
    os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
    torch.save({'pred_tracks': pred_tracks, 'point_embedding': point_embedding,
                'point_name2id': dict(point_name2id)}, path)


# ------------------------------------------------------------------------------------------------
# ED-LoRA checkpoints
# ------------------------------------------------------------------------------------------------
def load_edlora(path):
    """Returns the checkpoint as stored ({'params': {...}} or the bare params dict: convert_edlora accepts both)."""
    state = _load(path)
    params = state['params'] if isinstance(state, dict) and 'params' in state else state
    if not isinstance(params, dict) or not ({'new_concept_embedding', 'unet', 'text_encoder'} & set(params)):
        raise FormatError(f'{path}: not an ED-LoRA checkpoint (params / new_concept_embedding / unet / text_encoder)')
    for part in ('unet', 'text_encoder'):
        for k in params.get(part, {}):
            if not (k.endswith('lora_down.weight') or k.endswith('lora_up.weight')):
                raise FormatError(f'{path}: unexpected key {part}/{k}')
    for name, emb in params.get('new_concept_embedding', {}).items():
        if emb.dim() != 2:
            raise FormatError(f'{path}: concept embedding {name} must be [tokens, dim], got {tuple(emb.shape)}')
    return state


def split_lora_path(spec):
    """`<path>---<alpha>` (options/test_videoswap/**: `lora_path`): returns (path, alpha, enable_edlora)."""
    path, alpha = spec.split('---')
    return path, float(alpha), 'edlora' in path


# ------------------------------------------------------------------------------------------------
# motion module / adapter / SD unet / scheduler
# ------------------------------------------------------------------------------------------------
def rename_motion_module_keys(state_dict):
    """test.py:63 — the AnimateDiff checkpoints name the positional-encoding buffer `...attention_blocks.N.pos_encoder.pe`;
    here (as in the reference) it lives on the processor."""
    return {k.replace('.pos_encoder', '.processor.pos_encoder'): v for k, v in state_dict.items()}


def load_motion_module(unet, path):
    sd = rename_motion_module_keys(_load(path))
    bad = [k for k in sd if 'motion_modules' not in k]
    if bad:
        raise FormatError(f'{path}: not a motion-module checkpoint (e.g. key {bad[0]})')
    return unet.load_state_dict(sd, strict=False)


def load_adapter_weights(adapter, path):
    return adapter.load_state_dict(_load(path))


def scheduler_config_from_pretrained(pretrained_model_path, subfolder='scheduler'):
    path = os.path.join(pretrained_model_path, subfolder or '', 'scheduler_config.json')
    if not os.path.isfile(path):
        raise FormatError(f'{path} does not exist')
    with open(path) as f:
        cfg = json.load(f)
    return {k: v for k, v in cfg.items() if not k.startswith('_')}


def read_new_concept_cfg(pretrained_model_path):
    path = os.path.join(pretrained_model_path, 'new_concept_cfg.json')
    if not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f)


# ------------------------------------------------------------------------------------------------
# synthetic files in the layouts above
# ------------------------------------------------------------------------------------------------
def synthetic_lora_state(unet_state_dict, text_state_dict=None, concepts=('<new1>', '<new2>'), text_dim=768,
                         rank=4, seed=4, std=0.01):
    """Rank-`rank` factors N(0, std^2) on the keys convert_edlora_to_diffusers.py:46-53 merges."""
    g = torch.Generator().manual_seed(seed)

    def factors(sd, suffixes, keep):
        out = {}
        for k, w in sd.items():
            if not any(k.endswith(s) for s in suffixes) or not keep(k):
                continue
            base = k[:-len('weight')]
            down = torch.randn(rank, w.shape[1], generator=g) * std
            up = torch.randn(w.shape[0], rank, generator=g) * std
            if w.dim() == 4:
                down, up = down[:, :, None, None], up[:, :, None, None]
            out[base + 'lora_down.weight'], out[base + 'lora_up.weight'] = down, up
        return out
    from .edlora import TEXT_LORA_KEYS, UNET_LORA_KEYS
    params = {'new_concept_embedding': {c: torch.randn(16, text_dim, generator=g) * 0.02 for c in concepts},
              'unet': factors(unet_state_dict, UNET_LORA_KEYS, lambda k: 'attentions' in k and 'motion' not in k)}
    if text_state_dict is not None:
        params['text_encoder'] = factors(text_state_dict, TEXT_LORA_KEYS, lambda k: 'encoder.layers' in k)
    return {'params': params}


def synthetic_frames(directory, count, width, height, seed=0):
    """`count` RGB PNG frames of a drifting pattern (the dataset reads any image files, sorted by name)."""
    from PIL import Image
    os.makedirs(directory, exist_ok=True)
    g = torch.Generator().manual_seed(seed)
    base = torch.rand(3, height // 8 + 2, width // 8 + 2, generator=g)
    for i in range(count):
        img = torch.nn.functional.interpolate(base[None], size=(height + 16, width + 16), mode='bilinear',
                                              align_corners=False)[0]
        dx = i % 16
        frame = (img[:, 8:8 + height, dx:dx + width] * 255).round().clamp(0, 255).byte().permute(1, 2, 0).numpy()
        Image.fromarray(frame, 'RGB').save(os.path.join(directory, f'{i:05d}.png'))


def synthetic_tap(path, total_frames, width, height, points=8, embedding_dim=1280, seed=2):
    g = torch.Generator().manual_seed(seed)
    tracks = torch.rand(total_frames, points, 2, generator=g) * torch.tensor([float(width), float(height)])
    tracks[torch.rand(total_frames, points, generator=g) < 0.1] = -1.0
    emb = torch.randn(points, embedding_dim, generator=g)
    save_tap(path, tracks, emb, {f'p{i}': i for i in range(points)})
