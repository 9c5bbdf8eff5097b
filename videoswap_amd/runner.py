"""`test.py -opt <yml>` of the reference (test.py:24-136) on this package: the same steps in the same order — UNet from
the SD directory + motion-module checkpoint, point adapter, pipeline assembly from the SD directory, dataset, ED-LoRA
concept cfg, `validation`, result files — without `accelerate` (one process, one GPU).

    python -m videoswap_amd.runner -opt options/test_videoswap/animal/2001_catheadturn_T05_Iter100/....yml
        [--set datasets.num_frames=4 --set val.editing_config.num_inference_steps=2]   # BASELINE.json configs[0]

Relative paths in the YAML are resolved against the current directory, as in the reference; results go to
`$VSX_RESULTS_ROOT` (default: ./results)/<name>/visualization.  The reference's own `test.py` runs UNCHANGED on the
same code through `python -m videoswap_amd.dropin /path/to/test.py -opt <yml>` (shim packages `diffusers`,
`omegaconf`, `videoswap.*`).  `--train` runs the train.py flow (adapter training, options/train_videoswap/**.yml)."""
import argparse
import json
import os
import random

import torch

from . import build_model, build_pipeline, formats
from .compat import DDIMScheduler
from .config import OmegaConf, load_options
from .data import build_dataset
from .edlora import revise_edlora_unet_attention_forward
from .utils import dict2str, save_video_to_dir, set_path_logger


def set_seed(seed):
    random.seed(seed)
    try:
        import numpy as np
        np.random.seed(seed % (2 ** 32))
    except ImportError:      # pragma: no cover
        pass
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def _weight_dtype(opt, unet_cls):
    """`mixed_precision` of the option file -> dtype of the frozen models.  The HIP kernels compute in fp16 storage /
    fp32 accumulate and nothing else: with the product's UNet class anything but 'fp16' is refused HERE (test.py:73-75
    would cast to fp32 and the first kernel call would fail deep inside the forward)."""
    mixed = opt.get('mixed_precision', 'no')
    dtype = {'fp16': torch.float16, 'bf16': torch.bfloat16}.get(mixed, torch.float32)
    if dtype != torch.float16 and unet_cls.__module__.startswith('videoswap_amd.'):
        raise ValueError(f"mixed_precision: {mixed!r} is not supported by the MI355X kernels (fp16 storage, fp32 "
                         f"accumulation); set `mixed_precision: fp16` in the option file")
    return dtype


def build_from_options(opt, device='cuda', classes=None):
    """test.py:45-91 — returns (pipeline, adapter, weight_dtype).  `classes` may override the model classes by
    registry name (tests run the same orchestration on the CPU oracle)."""
    classes = classes or {}
    unet_opt = dict(opt['models']['unet'])
    unet_type = unet_opt.pop('type')
    if unet_type != 'AnimateDiffUNet3DModel':
        raise NotImplementedError(unet_type)
    unet_cls = classes.get(unet_type) or build_model(unet_type)
    weight_dtype = _weight_dtype(opt, unet_cls)
    kwargs = OmegaConf.to_container(OmegaConf.load(unet_opt.pop('inference_config_path')).unet_additional_kwargs)
    unet = unet_cls.from_pretrained_2d(opt['path']['pretrained_model_path'], subfolder='unet',
                                       unet_additional_kwargs=kwargs)
    if unet_opt.get('motion_module_path'):
        sd = formats.rename_motion_module_keys(formats.load_checkpoint(unet_opt['motion_module_path']))
        unet.load_state_dict(sd, strict=False)

    adapter_opt = dict(opt['models']['adapter'])
    adapter_type = adapter_opt.pop('type')
    adapter_cls = classes.get(adapter_type) or build_model(adapter_type)
    adapter = adapter_cls(**OmegaConf.to_container(OmegaConf.load(adapter_opt['model_config_path'])))
    adapter.load_state_dict(formats.load_checkpoint(opt['path']['pretrained_adapter_path']))
    adapter = adapter.to(dtype=weight_dtype)

    pipe_cls = classes.get(opt['val']['val_pipeline']) or build_pipeline(opt['val']['val_pipeline'])
    pipe = pipe_cls.from_pretrained(
        opt['path']['pretrained_model_path'], unet=unet.to(dtype=weight_dtype), adapter=adapter,
        scheduler=DDIMScheduler.from_pretrained(opt['path']['pretrained_model_path'], subfolder='scheduler'),
        torch_dtype=weight_dtype).to(device)
    pipe.enable_vae_slicing()

    cfg = formats.read_new_concept_cfg(opt['path']['pretrained_model_path'])
    if cfg is not None:
        revise_edlora_unet_attention_forward(pipe.unet)
        pipe.set_new_concept_cfg(cfg)
    pipe.scheduler.set_timesteps(opt['val']['editing_config']['num_inference_steps'])
    return pipe, adapter, weight_dtype


def test(root_path, opt, opt_path, device='cuda', classes=None, save=True):
    """test.py:24-124.  Returns (edited_results, visualization dir)."""
    set_path_logger(None, root_path, opt_path, opt, is_train=False)
    print(dict2str(opt))
    if opt.get('manual_seed') is None:
        opt['manual_seed'] = random.randint(1, 10000)
    set_seed(opt['manual_seed'])

    pipe, adapter, weight_dtype = build_from_options(opt, device, classes)

    dataset_opt = opt['datasets']
    dataset_type = dataset_opt.pop('type')
    dataset = build_dataset(dataset_type)(dataset_opt)
    frames = dataset.get_frames()
    conditions = None
    if adapter is not None:
        adapter.eval()
        conditions = dataset.get_conditions()

    edited = pipe.validation(source_video=frames, source_conditions=conditions, source_prompt=opt['datasets']['prompt'],
                             editing_config=opt['val']['editing_config'], dtype=weight_dtype, train_dataset=dataset,
                             save_dir=opt['path']['visualization'])
    save_dir = opt['path']['visualization']
    if save:
        kind, fps = opt['val'].get('save_type', 'frame_gif'), opt['val'].get('fps', 8)
        save_video_to_dir(frames, save_dir=os.path.join(save_dir, 'source'), save_suffix='source', save_type=kind, fps=fps)
        for key, video in edited.items():
            if not isinstance(video, list):
                continue                         # latent outputs (no VAE plugged in) are returned, not written
            out_dir = save_dir if 'frame' not in kind else os.path.join(save_dir, key)
            suffix = f"{key}_{opt['name']}" if opt['val'].get('use_suffix', False) else f'{key}'
            save_video_to_dir(video, save_dir=out_dir, save_suffix=suffix, save_type=kind, fps=fps)
    return edited, save_dir


def get_scheduler(name, optimizer, num_warmup_steps=0, num_training_steps=None):
    """diffusers.optimization.get_scheduler for the schedules the option files name (train.py:115-120)"""
    from torch.optim.lr_scheduler import LambdaLR
    import math
    warm = max(int(num_warmup_steps), 0)
    total = num_training_steps

    def ramp(step):
        return float(step) / float(max(1, warm)) if step < warm else None
    if name == 'constant':
        fn = (lambda step: 1.0)
    elif name == 'constant_with_warmup':
        fn = (lambda step: ramp(step) if ramp(step) is not None else 1.0)
    elif name == 'linear':
        fn = (lambda step: ramp(step) if ramp(step) is not None else
              max(0.0, float(total - step) / float(max(1, total - warm))))
    elif name == 'cosine':
        fn = (lambda step: ramp(step) if ramp(step) is not None else
              max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(step - warm) / float(max(1, total - warm))))))
    else:
        raise NotImplementedError(f'lr_scheduler {name!r}')
    return LambdaLR(optimizer, fn)


def train(root_path, opt, opt_path, device='cuda', classes=None, max_iters=None):
    """train.py:24-224 for one process: models from the SD directory + motion-module checkpoint, a fresh
    SparsePointAdapter, AdamW on the adapter only, `VideoSwapTrainer.step` per iteration, validation and `adapter.pth`
    checkpoints at the configured frequencies.  Returns {'losses', 'checkpoints', 'trainer'}."""
    import logging
    from .clip import CLIPTextModel, load_tokenizer
    from .compat import DDPMScheduler
    from .vae import AutoencoderKL
    classes = classes or {}
    set_path_logger(None, root_path, opt_path, opt, is_train=True)
    logger = logging.getLogger('videoswap')
    logger.info(dict2str(opt))
    if opt.get('manual_seed') is None:
        opt['manual_seed'] = random.randint(1, 10000)
    set_seed(opt['manual_seed'])
    sd_dir = opt['path']['pretrained_model_path']
    weight_dtype = _weight_dtype(opt, classes.get('AnimateDiffUNet3DModel') or build_model('AnimateDiffUNet3DModel'))

    tokenizer = load_tokenizer(sd_dir)
    text_encoder = CLIPTextModel.from_pretrained(sd_dir, subfolder='text_encoder', torch_dtype=weight_dtype)
    vae = AutoencoderKL.from_pretrained(sd_dir, subfolder='vae', torch_dtype=weight_dtype)
    unet_opt = dict(opt['models']['unet'])
    unet_type = unet_opt.pop('type')
    if unet_type != 'AnimateDiffUNet3DModel':
        raise NotImplementedError(unet_type)
    kwargs = OmegaConf.to_container(OmegaConf.load(unet_opt.pop('inference_config_path')).unet_additional_kwargs)
    unet = (classes.get(unet_type) or build_model(unet_type)).from_pretrained_2d(
        sd_dir, subfolder='unet', unet_additional_kwargs=kwargs)
    sd = formats.rename_motion_module_keys(formats.load_checkpoint(unet_opt['motion_module_path']))
    unet.load_state_dict(sd, strict=False)
    adapter_opt = dict(opt['models']['adapter'])
    adapter_type = adapter_opt.pop('type')
    adapter = (classes.get(adapter_type) or build_model(adapter_type))(
        **OmegaConf.to_container(OmegaConf.load(adapter_opt['model_config_path'])))
    for frozen in (vae, unet, text_encoder):
        frozen.requires_grad_(False)
    unet = unet.to(device, weight_dtype)
    vae, text_encoder = vae.to(device, weight_dtype), text_encoder.to(device, weight_dtype)
    adapter = adapter.to(device)                         # fp32 master weights (train.py:135-144: mixed precision)

    val_pipeline = build_pipeline(opt['val']['val_pipeline'])(
        vae=vae, text_encoder=text_encoder, tokenizer=tokenizer, unet=unet, adapter=adapter,
        scheduler=DDIMScheduler.from_pretrained(sd_dir, subfolder='scheduler')).to(device)
    val_pipeline.enable_vae_slicing()
    val_pipeline.scheduler.set_timesteps(opt['val']['editing_config']['num_inference_steps'])

    optim_opt = dict(opt['train']['optimizer'])
    if optim_opt.pop('type') != 'AdamW':
        raise NotImplementedError('optimizer')
    optim_opt['betas'] = tuple(optim_opt.get('betas', (0.9, 0.999)))
    optimizer = torch.optim.AdamW(adapter.parameters(), **optim_opt)
    total_iter = int(opt['train']['total_iter']) if max_iters is None else int(max_iters)
    lr_scheduler = get_scheduler(opt['train']['lr_scheduler'], optimizer, opt['train'].get('warmup_iter', 0),
                                 opt['train']['total_iter'])

    dataset_opt = dict(opt['datasets'])
    dataset = build_dataset(dataset_opt.pop('type'))(dataset_opt)
    loader = torch.utils.data.DataLoader(dataset, batch_size=dataset_opt['batch_size_per_gpu'], shuffle=True,
                                         num_workers=0)
    trainer = build_pipeline(opt['train']['train_pipeline'])(
        vae=vae, text_encoder=text_encoder, tokenizer=tokenizer, unet=unet, adapter=adapter,
        scheduler=DDPMScheduler.from_pretrained(sd_dir, subfolder='scheduler'), weight_dtype=weight_dtype,
        optimizer=optimizer, max_grad_norm=1.0, lr_scheduler=lr_scheduler, tune_cfg=opt['train'].get('tune_cfg'))
    trainer.to(device)
    adapter.to(device=device, dtype=torch.float32)

    def batches():
        while True:
            for b in loader:
                yield b
    stream = batches()
    losses, checkpoints = [], []
    step = 0
    while step < total_iter:
        batch = next(stream)
        batch = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}
        loss = trainer.step(batch)
        losses.append(float(loss))
        step += 1
        if step % opt['logger']['print_freq'] == 0:
            logger.info(f'iter {step}: loss {losses[-1]:.5f} lr {lr_scheduler.get_last_lr()} '
                        f'loss scale {trainer.loss_scale:g}')
        if step % opt['val']['val_freq'] == 0:
            _validate(unet, adapter, dataset, val_pipeline, opt, weight_dtype, step)
        if step % int(opt['logger']['save_checkpoint_freq']) == 0:
            save_dir = os.path.join(opt['path']['models'], f'models_{step}')
            os.makedirs(save_dir, exist_ok=True)
            torch.save({k: v.detach().cpu() for k, v in adapter.state_dict().items()},
                       os.path.join(save_dir, 'adapter.pth'))
            checkpoints.append(os.path.join(save_dir, 'adapter.pth'))
            logger.info(f'save to {save_dir}')
    return {'losses': losses, 'checkpoints': checkpoints, 'trainer': trainer}


def _validate(unet, adapter, dataset, val_pipeline, opt, weight_dtype, global_step):
    """train.py:227-262"""
    unet.eval()
    adapter.eval()
    half = adapter.__class__(**dict(adapter.config)).to(next(adapter.parameters()).device, weight_dtype)
    half.load_state_dict(adapter.state_dict())
    val_pipeline.adapter = half.eval()               # sampling runs the adapter in the activation dtype
    frames = dataset.get_frames()
    edited = val_pipeline.validation(source_video=frames, source_conditions=dataset.get_conditions(),
                                     source_prompt=opt['datasets']['prompt'],
                                     editing_config=opt['val']['editing_config'], dtype=weight_dtype,
                                     train_dataset=dataset, save_dir=opt['path']['visualization'])
    kind, fps = opt['val'].get('save_type', 'frame_gif'), opt['val'].get('fps', 8)
    for key, video in edited.items():
        if isinstance(video, list):
            save_video_to_dir(video, save_dir=os.path.join(opt['path']['visualization'], f'iter_{global_step}'),
                              save_suffix=key, save_type=kind, fps=fps)
    val_pipeline.adapter = adapter
    adapter.train()
    return edited


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('-opt', type=str, required=True)
    ap.add_argument('--train', action='store_true', help='train.py flow (adapter training) instead of test.py')
    ap.add_argument('--set', action='append', default=[], metavar='dotted.key=json', help='override an option')
    ap.add_argument('--device', default='cuda')
    args = ap.parse_args(argv)
    overrides = {}
    for item in args.set:
        k, v = item.split('=', 1)
        try:
            overrides[k] = json.loads(v)
        except json.JSONDecodeError:
            overrides[k] = v
    opt = load_options(args.opt, overrides)
    root = os.path.abspath(os.getcwd())
    if args.train:
        res = train(root, opt, args.opt, device=args.device)
        print('checkpoints:', res['checkpoints'])
        return
    _, out = test(root, opt, args.opt, device=args.device)
    print('results in', out)


if __name__ == '__main__':
    main()
