"""The reference's own `test.py`, UNCHANGED, on this repository's class surface (build container only: it needs
/root/reference).  There is no GPU here, so the compute classes behind the shim module names are the CPU oracle's
(tests may use the oracle; the product's shims under videoswap_amd/shims bind the HIP classes and are exercised on the
GPU box by tests/test_config.py::test_dropin_script_through_the_shims).  What this pins is the drop-in contract:
every import `test.py` makes resolves, every call it makes (signatures, keyword names, return shapes, attribute
access, the order of operations) is accepted, and it runs to completion on checkpoints in the real on-disk formats."""
import os
import runpy
import sys
import types

import pytest

from util import GOLDEN, ROOT

REF_TEST = '/root/reference/test.py'
YML = '/root/reference/options/test_videoswap/animal/2001_catheadturn_T05_Iter100/2001_catheadturn_T05_Iter100.yml'


@pytest.mark.skipif(not os.path.isfile(REF_TEST), reason='reference tree not present')
def test_reference_test_py_runs_unchanged(tmp_path, monkeypatch):
    pytest.importorskip('accelerate')
    import yaml
    from oracle.validation import oracle_classes
    from videoswap_amd import config, data, utils
    from videoswap_amd.compat import DDIMScheduler
    from videoswap_amd.config import load_options
    from videoswap_amd.edlora import revise_edlora_unet_attention_forward
    from videoswap_amd.workspace import write_synthetic_workspace

    small = [{'type': 'Resize', 'size': 256}, {'type': 'ToTensor'}, {'type': 'Normalize', 'mean': [0.5], 'std': [0.5]}]
    opt = load_options(YML, {'datasets.num_frames': 4, 'val.editing_config.num_inference_steps': 2,
                             'mixed_precision': 'no', 'val.save_type': 'frame_gif',
                             'datasets.video_transform': small})
    # one swap is enough for the contract (three in the file)
    eps = opt['val']['editing_config']['editing_prompts']
    opt['val']['editing_config']['editing_prompts'] = {'kitten_to_catA': eps['kitten_to_catA']}
    write_synthetic_workspace(str(tmp_path), opt, width='tiny', total_frames=8)
    with open(tmp_path / 'opt.yml', 'w') as f:
        yaml.safe_dump(opt, f)

    classes = oracle_classes()

    class CpuPipeline(classes['VideoSwapPipeline']):
        def to(self, device=None, dtype=None):          # test.py:80 says .to('cuda'); no GPU in this container
            return super().to('cpu', dtype)

    def module(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        monkeypatch.setitem(sys.modules, name, m)
        return m
    module('diffusers', DDIMScheduler=DDIMScheduler)
    module('omegaconf', OmegaConf=config.OmegaConf)
    pkg = module('videoswap')
    pkg.__path__ = []
    module('videoswap.data', build_dataset=data.build_dataset)
    module('videoswap.models', build_model=lambda name: classes[name])
    module('videoswap.pipelines', build_pipeline=lambda name: CpuPipeline)
    u = module('videoswap.utils')
    u.__path__ = []
    module('videoswap.utils.edlora_util', revise_edlora_unet_attention_forward=revise_edlora_unet_attention_forward)
    module('videoswap.utils.logger', dict2str=utils.dict2str, set_path_logger=utils.set_path_logger)
    module('videoswap.utils.vis_util', save_video_to_dir=utils.save_video_to_dir)

    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv('VSX_RESULTS_ROOT', str(tmp_path / 'results'))
    monkeypatch.setattr(sys, 'argv', [REF_TEST, '-opt', 'opt.yml'])
    monkeypatch.setattr(sys, 'dont_write_bytecode', True)
    import torch
    real_generator = torch.Generator

    def cpu_generator(device='cpu'):                    # validation seeds torch.Generator(device='cuda') (:381)
        return real_generator('cpu')
    monkeypatch.setattr(torch, 'Generator', cpu_generator)
    runpy.run_path(REF_TEST, run_name='__main__')
    out = tmp_path / 'results' / opt['name'] / 'visualization'
    assert (out / 'kitten_to_catA' / 'kitten_to_catA.gif').is_file()
    assert (out / 'source' / 'source.gif').is_file()
