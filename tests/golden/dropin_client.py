"""A client of the reference's import surface — the imports and the call sequence of the reference's `test.py`
(test.py:14-124), written for the drop-in test on the GPU box, where /root/reference does not exist.  In the build
container the reference's own `test.py` runs unchanged through the same shims (tests/test_dropin_reference.py).

Run with:  python -m videoswap_amd.dropin tests/golden/dropin_client.py -opt <yml>"""
import argparse
import json
import os

import torch
from diffusers import DDIMScheduler
from omegaconf import OmegaConf

from videoswap.data import build_dataset
from videoswap.models import build_model
from videoswap.pipelines import build_pipeline
from videoswap.utils.edlora_util import revise_edlora_unet_attention_forward
from videoswap.utils.logger import dict2str, set_path_logger
from videoswap.utils.vis_util import save_video_to_dir


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('-opt', type=str, required=True)
    args = ap.parse_args()
    opt = OmegaConf.to_container(OmegaConf.load(args.opt), resolve=True)
    set_path_logger(None, os.getcwd(), args.opt, opt, is_train=False)
    print(dict2str(opt)[:200])
    torch.manual_seed(opt.get('manual_seed') or 0)
    weight_dtype = torch.float16 if opt['mixed_precision'] == 'fp16' else torch.float32

    unet_type = opt['models']['unet'].pop('type')
    cfg_path = opt['models']['unet'].pop('inference_config_path')
    unet = build_model(unet_type).from_pretrained_2d(
        opt['path']['pretrained_model_path'], subfolder='unet',
        unet_additional_kwargs=OmegaConf.to_container(OmegaConf.load(cfg_path).unet_additional_kwargs))
    mm = torch.load(opt['models']['unet'].pop('motion_module_path'), map_location='cpu')
    mm = {k.replace('.pos_encoder', '.processor.pos_encoder'): v for k, v in mm.items()}
    missing, unexpected = unet.load_state_dict(mm, strict=False)
    assert not unexpected, unexpected[:3]

    adapter_type = opt['models']['adapter'].pop('type')
    adapter = build_model(adapter_type)(**OmegaConf.to_container(OmegaConf.load(opt['models']['adapter']['model_config_path'])))
    adapter.load_state_dict(torch.load(opt['path']['pretrained_adapter_path']))
    adapter = adapter.to(dtype=weight_dtype)

    pipe = build_pipeline(opt['val']['val_pipeline']).from_pretrained(
        opt['path']['pretrained_model_path'], unet=unet.to(dtype=weight_dtype), adapter=adapter,
        scheduler=DDIMScheduler.from_pretrained(opt['path']['pretrained_model_path'], subfolder='scheduler'),
        torch_dtype=weight_dtype).to('cuda')
    pipe.enable_vae_slicing()
    cfg_file = os.path.join(opt['path']['pretrained_model_path'], 'new_concept_cfg.json')
    if os.path.exists(cfg_file):
        with open(cfg_file) as f:
            revise_edlora_unet_attention_forward(pipe.unet)
            pipe.set_new_concept_cfg(json.load(f))
    pipe.scheduler.set_timesteps(opt['val']['editing_config']['num_inference_steps'])

    dataset_opt = opt['datasets']
    dataset = build_dataset(dataset_opt.pop('type'))(dataset_opt)
    frames = dataset.get_frames()
    adapter.eval()
    edited = pipe.validation(source_video=frames, source_conditions=dataset.get_conditions(),
                             source_prompt=opt['datasets']['prompt'], editing_config=opt['val']['editing_config'],
                             train_dataset=dataset, save_dir=opt['path']['visualization'])
    for key, video in edited.items():
        save_video_to_dir(video, save_dir=os.path.join(opt['path']['visualization'], key), save_suffix=key,
                          save_type='frame_gif', fps=8)
    assert len(edited) == len(opt['val']['editing_config']['editing_prompts'])
    print('DROPIN_OK', sorted(edited))


if __name__ == '__main__':
    main()
