"""The reference's own `train.py`, UNCHANGED, driving this repository's classes (build container only: it needs
/root/reference): `accelerate`, `transformers` (tokenizer + CLIP text encoder, as train.py itself imports them) are the
installed packages, `diffusers.*` / `omegaconf` / `videoswap.*` resolve to the shim packages of videoswap_amd/shims —
i.e. the PRODUCT classes (AnimateDiffUNet3DModel, SparsePointAdapter, VideoSwapTrainer, VideoSwapPipeline, AutoencoderKL),
here on CPU tensors through tests/host_emulation.py.  Pins the training drop-in contract: every import resolves, every call
(accelerator.prepare / backward / clip_grad_norm_ / save, trainer.step, MessageLogger, reduce_loss_dict) is accepted, the
loop runs its iterations and writes `adapter.pth` where test.py expects it."""
import os
import runpy
import sys

import pytest
import torch

REF_TRAIN = '/root/reference/train.py'


@pytest.mark.skipif(not os.path.isfile(REF_TRAIN) or torch.cuda.is_available(),
                    reason='needs the reference tree; on a GPU box the product runs train.py through videoswap_amd.dropin')
def test_reference_train_py_runs_unchanged(tmp_path, monkeypatch):
    pytest.importorskip('accelerate')
    import yaml
    import host_emulation
    from videoswap_amd import dropin
    from videoswap_amd.workspace import write_synthetic_workspace
    from test_training import _train_options
    opt = _train_options(tmp_path, {'val.val_freq': 1000, 'train.total_iter': 2, 'logger.save_checkpoint_freq': 2,
                                    'datasets.dataset_enlarge_ratio': 2})
    write_synthetic_workspace(str(tmp_path), opt, width='tiny', total_frames=9)
    with open(tmp_path / 'opt.yml', 'w') as f:
        yaml.safe_dump(opt, f)
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv('VSX_RESULTS_ROOT', str(tmp_path / 'experiments'))
    monkeypatch.setattr(sys, 'argv', [REF_TRAIN, '-opt', 'opt.yml'])
    monkeypatch.setattr(sys, 'dont_write_bytecode', True)
    saved_path, saved_mods = list(sys.path), dict(sys.modules)
    from accelerate.state import AcceleratorState
    AcceleratorState._reset_state(reset_partial_state=True)      # another test may have built an Accelerator('no')
    try:
        dropin.install()
        with host_emulation.installed():
            runpy.run_path(REF_TRAIN, run_name='__main__')
    finally:                                   # other tests bind stub `diffusers.*` modules: put them back
        AcceleratorState._reset_state(reset_partial_state=True)
        sys.path[:] = saved_path
        for name in [m for m in sys.modules if m.split('.')[0] in ('videoswap', 'diffusers', 'omegaconf')]:
            del sys.modules[name]
        sys.modules.update({k: v for k, v in saved_mods.items()
                            if k.split('.')[0] in ('videoswap', 'diffusers', 'omegaconf')})
    ckpt = tmp_path / 'experiments' / opt['name'] / 'models' / 'models_2' / 'adapter.pth'
    assert ckpt.is_file()
    sd = torch.load(ckpt, map_location='cpu')
    assert all(torch.isfinite(v).all() for v in sd.values())
