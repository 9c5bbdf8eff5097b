"""Write a complete synthetic workspace for a reference YAML: every file `test.py -opt <yml>` opens, in the layouts of
videoswap_amd/formats.py, with seeded random weights (no checkpoint or dataset exists offline).  Used by the drop-in
test of the reference's `test.py`, by BASELINE.json configs[0] and by `python -m videoswap_amd.workspace`.

    python -m videoswap_amd.workspace -opt <yml> --root <dir> [--width tiny|full]

`width='tiny'` keeps the architecture (4 levels, 8 heads, 32 groups, motion modules everywhere) at
block_out_channels (64, 128, 256, 256), a 64-wide 2-layer text encoder and a small VAE so that the whole run fits a
CPU test; `full` writes the SD-1.5 shapes (about 2.5 GB of fp16 files)."""
import argparse
import json
import os

import torch
import yaml

from . import formats
from .config import load_options

SD15_UNET_2D_CONFIG = {
    '_class_name': 'UNet2DConditionModel', '_diffusers_version': '0.19.3', 'act_fn': 'silu', 'attention_head_dim': 8,
    'block_out_channels': [320, 640, 1280, 1280], 'center_input_sample': False, 'cross_attention_dim': 768,
    'down_block_types': ['CrossAttnDownBlock2D', 'CrossAttnDownBlock2D', 'CrossAttnDownBlock2D', 'DownBlock2D'],
    'downsample_padding': 1, 'flip_sin_to_cos': True, 'freq_shift': 0, 'in_channels': 4, 'layers_per_block': 2,
    'mid_block_scale_factor': 1, 'norm_eps': 1e-05, 'norm_num_groups': 32, 'out_channels': 4, 'sample_size': 64,
    'up_block_types': ['UpBlock2D', 'CrossAttnUpBlock2D', 'CrossAttnUpBlock2D', 'CrossAttnUpBlock2D']}


def _words(opt):
    ec = opt['val']['editing_config']
    text = [opt['datasets']['prompt'], ec.get('negative_prompt') or '']
    for cfg in ec['editing_prompts'].values():
        for key in ('replace', 'replace_other'):
            if key in cfg:
                text.extend(s.strip() for s in cfg[key].split('->'))
        text.append(cfg.get('negative_prompt') or '')
    words = []
    for t in text:
        for w in t.replace(',', ' , ').lower().split():
            if not (w.startswith('<') and w.endswith('>')) and w not in words:
                words.append(w)
    return words


def write_tokenizer(directory, words):
    """A valid CLIP BPE vocabulary (vocab.json + merges.txt) in which every word of `words` is one token; any other
    text falls back to per-character tokens."""
    os.makedirs(directory, exist_ok=True)
    vocab, merges = {}, []

    def add(tok):
        vocab.setdefault(tok, len(vocab))
    for c in [chr(i) for i in range(33, 127)]:
        add(c)
        add(c + '</w>')
    for w in words:
        parts = list(w[:-1]) + [w[-1] + '</w>']
        while len(parts) > 1:
            pair = (parts[0], parts[1])
            if pair not in merges:
                merges.append(pair)
            parts = [parts[0] + parts[1]] + parts[2:]
            add(parts[0])
    add('<|startoftext|>')
    add('<|endoftext|>')
    with open(os.path.join(directory, 'vocab.json'), 'w') as f:
        json.dump(vocab, f)
    with open(os.path.join(directory, 'merges.txt'), 'w') as f:
        f.write('#version: 0.2\n' + '\n'.join(f'{a} {b}' for a, b in merges) + '\n')
    with open(os.path.join(directory, 'tokenizer_config.json'), 'w') as f:
        json.dump({'model_max_length': 77, 'tokenizer_class': 'CLIPTokenizer', 'bos_token': '<|startoftext|>',
                   'eos_token': '<|endoftext|>', 'pad_token': '<|endoftext|>', 'unk_token': '<|endoftext|>'}, f)
    return len(vocab)


def _save(obj, path):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    torch.save(obj, path)


def write_synthetic_workspace(root, opt, width='tiny', seed=1234, total_frames=None, image_size=None):
    """Creates every path the option dict names, relative to `root`.  Returns a dict of what was written."""
    from .adapter import SparsePointAdapter
    from .clip import CLIPTextConfig, CLIPTextModel
    from .compat import SD15_SCHEDULER_CONFIG
    from .synthetic import synth_weights_
    from .unet import AnimateDiffUNet3DModel, inference_kwargs
    from .vae import SD15_VAE_CONFIG, AutoencoderKL

    def at(rel):
        return rel if os.path.isabs(rel) else os.path.join(root, rel)

    tiny = width == 'tiny'
    boc = [64, 128, 256, 256] if tiny else [320, 640, 1280, 1280]
    text_dim = 64 if tiny else 768
    store = torch.float32 if tiny else torch.float16
    sd_dir = at(opt['path']['pretrained_model_path'])

    # ---- model configs the YAML points at (generated from this package's constants, not copied) ----
    inf_path = at(opt['models']['unet']['inference_config_path'])
    os.makedirs(os.path.dirname(inf_path), exist_ok=True)
    kw = inference_kwargs(max_len=24)
    kw['motion_module_resolutions'] = list(kw['motion_module_resolutions'])
    kw['motion_module_kwargs'] = dict(kw['motion_module_kwargs'],
                                      attention_block_types=list(kw['motion_module_kwargs']['attention_block_types']))
    with open(inf_path, 'w') as f:
        yaml.safe_dump({'unet_additional_kwargs': kw,
                        'noise_scheduler_kwargs': {'beta_start': 0.00085, 'beta_end': 0.012, 'beta_schedule': 'linear'}}, f)
    ad_path = at(opt['models']['adapter']['model_config_path'])
    os.makedirs(os.path.dirname(ad_path), exist_ok=True)
    adapter_cfg = {'channels': boc, 'embedding_channels': 1280, 'downsample_rate': [8, 16, 32, 64], 'mid_dim': 128}
    with open(ad_path, 'w') as f:
        yaml.safe_dump(adapter_cfg, f)

    # ---- SD directory: unet (2-D keys), scheduler, vae, text encoder, tokenizer ----
    unet_cfg = dict(SD15_UNET_2D_CONFIG, block_out_channels=boc, cross_attention_dim=text_dim)
    os.makedirs(os.path.join(sd_dir, 'unet'), exist_ok=True)
    with open(os.path.join(sd_dir, 'unet', 'config.json'), 'w') as f:
        json.dump(unet_cfg, f, indent=1)
    # build and fill the big model on the GPU when there is one (a CPU initialisation of the SD-1.5 width takes a minute)
    dev = 'cuda' if (not tiny and torch.cuda.is_available()) else 'cpu'
    with torch.device(dev):
        unet = AnimateDiffUNet3DModel(block_out_channels=tuple(boc), cross_attention_dim=text_dim, sample_size=64,
                                      **inference_kwargs(max_len=24))
    synth_weights_(unet, seed=seed)
    full_sd = {k: v.cpu() for k, v in unet.state_dict().items()}
    del unet
    _save({k: v.to(store).contiguous() for k, v in full_sd.items() if 'motion_modules' not in k},
          os.path.join(sd_dir, 'unet', 'diffusion_pytorch_model.bin'))
    mm_path = opt['models']['unet'].get('motion_module_path')
    if mm_path:        # AnimateDiff checkpoints name the PE buffer without the `.processor` level (test.py:63)
        _save({k.replace('.processor.pos_encoder', '.pos_encoder'): v.to(store).contiguous()
               for k, v in full_sd.items() if 'motion_modules' in k}, at(mm_path))
    os.makedirs(os.path.join(sd_dir, 'scheduler'), exist_ok=True)
    with open(os.path.join(sd_dir, 'scheduler', 'scheduler_config.json'), 'w') as f:
        json.dump(dict(SD15_SCHEDULER_CONFIG, _class_name='DDIMScheduler', _diffusers_version='0.19.3'), f, indent=1)

    vae_cfg = dict(SD15_VAE_CONFIG)
    if tiny:
        vae_cfg.update(block_out_channels=(32, 64, 64, 64), layers_per_block=1, norm_num_groups=8)
    vae = AutoencoderKL(**vae_cfg)
    synth_weights_(vae, seed=seed + 1)
    os.makedirs(os.path.join(sd_dir, 'vae'), exist_ok=True)
    with open(os.path.join(sd_dir, 'vae', 'config.json'), 'w') as f:
        json.dump({k: (list(v) if isinstance(v, tuple) else v) for k, v in vae_cfg.items()}, f, indent=1)
    _save({k: v.to(store).contiguous() for k, v in vae.state_dict().items()},
          os.path.join(sd_dir, 'vae', 'diffusion_pytorch_model.bin'))

    vocab = write_tokenizer(os.path.join(sd_dir, 'tokenizer'), _words(opt))
    clip_cfg = dict(vocab_size=vocab, hidden_size=text_dim, intermediate_size=2 * text_dim if tiny else 3072,
                    num_hidden_layers=2 if tiny else 12, num_attention_heads=4 if tiny else 12,
                    max_position_embeddings=77, hidden_act='quick_gelu', layer_norm_eps=1e-5)
    clip = CLIPTextModel(CLIPTextConfig(**clip_cfg))
    g = torch.Generator().manual_seed(seed + 2)
    with torch.no_grad():
        for name, p in clip.named_parameters():
            if 'layer_norm' in name and name.endswith('weight'):
                p.copy_(torch.rand(p.shape, generator=g) * 0.5 + 0.75)
            elif p.dim() > 1:
                p.copy_(torch.randn(p.shape, generator=g) * (0.05 if 'embedding' in name else p.shape[1] ** -0.5))
            else:
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
    os.makedirs(os.path.join(sd_dir, 'text_encoder'), exist_ok=True)
    with open(os.path.join(sd_dir, 'text_encoder', 'config.json'), 'w') as f:
        json.dump(dict(clip_cfg, architectures=['CLIPTextModel'], model_type='clip_text_model'), f, indent=1)
    _save({k: v.to(store).contiguous() for k, v in clip.state_dict().items()},
          os.path.join(sd_dir, 'text_encoder', 'pytorch_model.bin'))

    # ---- adapter, ED-LoRA checkpoints ----
    adapter = SparsePointAdapter(**adapter_cfg)
    synth_weights_(adapter, seed=seed + 3)
    if opt['path'].get('pretrained_adapter_path'):          # absent in the training option files (train.py builds a fresh one)
        _save(adapter.state_dict(), at(opt['path']['pretrained_adapter_path']))
    loras = []
    for i, cfg in enumerate(opt['val']['editing_config']['editing_prompts'].values()):
        if not cfg.get('lora_path'):
            continue
        path, _, _ = formats.split_lora_path(cfg['lora_path'])
        target = cfg['replace'].split('->')[1].strip()
        concepts = [w for w in target.split() if w.startswith('<') and w.endswith('>')]
        state = formats.synthetic_lora_state(full_sd, clip.state_dict(), concepts=concepts, text_dim=text_dim,
                                             seed=seed + 10 + i)
        _save(state, at(path))
        loras.append(path)

    # ---- dataset: frames + TAP.pth ----
    ds = opt['datasets']
    n_total = total_frames or ds.get('total_frames', ds['num_frames'])
    resize = next((t['size'] for t in ds.get('video_transform', []) if t['type'] == 'Resize'), 512)
    side = image_size or (resize if isinstance(resize, int) else resize[0])
    formats.synthetic_frames(at(ds['path']), n_total, side, side, seed=seed + 4)
    if 'tap_path' in ds:
        formats.synthetic_tap(at(ds['tap_path']), n_total, side, side, points=8, seed=seed + 5)
    return {'sd_dir': sd_dir, 'loras': loras, 'frames': n_total, 'image_size': side, 'width': width, 'vocab': vocab}


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('-opt', required=True)
    ap.add_argument('--root', required=True)
    ap.add_argument('--width', default='tiny', choices=('tiny', 'full'))
    args = ap.parse_args(argv)
    print(write_synthetic_workspace(args.root, load_options(args.opt), args.width))


if __name__ == '__main__':
    main()
