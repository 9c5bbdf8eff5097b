"""Small host utilities `test.py` uses around the pipeline (videoswap/utils/logger.py:55-128, vis_util.py:68-105):
experiment paths, option printing, writing result frames.  PIL only (imageio / torchvision are not installed: an
`.mp4` request falls back to an animated GIF next to it)."""
import logging
import os
import shutil
import time

import torch


def get_time_str():
    return time.strftime('%Y%m%d_%H%M%S', time.localtime())


def dict2str(opt, indent_level=1):
    msg = '\n'
    for k, v in opt.items():
        if isinstance(v, dict):
            msg += ' ' * (indent_level * 2) + str(k) + ':[' + dict2str(v, indent_level + 1) + ' ' * (indent_level * 2) + ']\n'
        else:
            msg += ' ' * (indent_level * 2) + str(k) + ': ' + str(v) + '\n'
    return msg


def set_path_logger(accelerator, root_path, config_path, opt, is_train=True):
    """logger.py:55-104 — sets opt['path'][results_root | log | visualization], creates the directories, copies the
    option file next to the results.  The results root is `$VSX_RESULTS_ROOT` or `<cwd>/results` rather than the
    directory of the calling script: the reference tree is read-only here."""
    opt['is_train'] = is_train
    kind = 'experiments' if is_train else 'results'
    base = os.environ.get('VSX_RESULTS_ROOT') or os.path.join(os.getcwd(), kind)
    root = os.path.join(base, opt['name'])
    if os.path.exists(root):
        os.rename(root, root + '_archived_' + get_time_str())
    opt.setdefault('path', {})
    opt['path']['experiments_root' if is_train else 'results_root'] = root
    opt['path']['log'] = root
    opt['path']['visualization'] = os.path.join(root, 'visualization')
    if is_train:
        opt['path']['models'] = os.path.join(root, 'models')
    os.makedirs(opt['path']['visualization'], exist_ok=True)
    if config_path and os.path.isfile(config_path):
        shutil.copyfile(config_path, os.path.join(root, os.path.basename(config_path)))
    log_file = os.path.join(root, f"{'train' if is_train else 'test'}_{opt['name']}_{get_time_str()}.log")
    logging.basicConfig(level=logging.INFO, format='%(asctime)s %(levelname)s: %(message)s',
                        handlers=[logging.FileHandler(log_file, 'w'), logging.StreamHandler()], force=True)


def save_images_as_gif(images, save_path, fps=5):
    images[0].save(save_path, save_all=True, append_images=images[1:], loop=0, duration=int(1000 / fps))


def save_video_to_dir(edit_video, save_dir, save_suffix, save_type='frame', fps=8):
    """vis_util.py:68-88: `save_type` is an underscore-joined subset of frame / gif / video."""
    os.makedirs(save_dir, exist_ok=True)
    kinds = save_type.split('_')
    if 'frame' in kinds:
        frame_dir = os.path.join(save_dir, 'frames')
        os.makedirs(frame_dir, exist_ok=True)
        for idx, img in enumerate(edit_video):
            img.save(os.path.join(frame_dir, f'{idx:05d}_{save_suffix}.jpg'))
    if 'gif' in kinds:
        save_images_as_gif(edit_video, os.path.join(save_dir, f'{save_suffix}.gif'), fps=fps)
    if 'video' in kinds:
        try:
            import imageio
            import numpy as np
            with imageio.get_writer(os.path.join(save_dir, f'{save_suffix}.mp4'), fps=fps) as wr:
                for img in edit_video:
                    wr.append_data(np.array(img))
        except ImportError:
            save_images_as_gif(edit_video, os.path.join(save_dir, f'{save_suffix}.mp4.gif'), fps=fps)


class MessageLogger:
    """logger.py:136-195: one line per `print_freq` iterations (iteration, learning rates, elapsed / remaining time,
    the reduced losses)."""

    def __init__(self, opt, start_iter=1, loss_print='f'):
        import time
        self.exp_name = opt['name']
        self.interval = opt['logger']['print_freq']
        self.start_iter = start_iter
        self.max_iters = opt['train']['total_iter']
        self.start_time = time.time()
        self.logger = logging.getLogger('videoswap')
        self.loss_print = loss_print

    def reset_start_time(self):
        import time
        self.start_time = time.time()

    def __call__(self, log_vars):
        import datetime
        import time
        current_iter = log_vars.pop('iter')
        lrs = log_vars.pop('lrs')
        message = f'[{self.exp_name[:5]}..][Iter:{current_iter:8,d}, lr:(' + ''.join(f'{v:.3e},' for v in lrs) + ')] '
        total = time.time() - self.start_time
        per_iter = total / max(current_iter - self.start_iter + 1, 1)
        eta = str(datetime.timedelta(seconds=int(per_iter * (self.max_iters - current_iter - 1))))
        message += f'[eta: {eta}] '
        for k, v in log_vars.items():
            message += f'{k}: {v:.4e} ' if self.loss_print == 'e' else f'{k}: {v:.4f} '
        self.logger.info(message)


def reduce_loss_dict(accelerator, loss_dict):
    """logger.py:198-225: average every loss over the processes -> {name: float}."""
    from collections import OrderedDict
    with torch.no_grad():
        keys = list(loss_dict)
        losses = torch.stack([loss_dict[k].detach().float() for k in keys], 0)
        world = 1
        if accelerator is not None:
            losses = accelerator.reduce(losses)
            world = accelerator.num_processes
        losses = losses / world
        return OrderedDict((k, float(v.mean())) for k, v in zip(keys, losses))
