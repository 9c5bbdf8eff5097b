"""f4 — the adapter training step.  CPU: the oracle adapter's training branch against the reference class imported
verbatim (same python RNG: same dropped points, same loss mask, same maps); the product's `VideoSwapTrainer` pieces
against the oracle restatement of trainer_videoswap.py:57-93 — loss value and the gradient of every adapter parameter
(HIP forward + backward through the frozen UNet vs PyTorch autograd on the fp32 oracle) — and one full optimizer
step.  `device` tests run the backward kernels of libvsx.so (csrc/train.hip) on a GPU box."""
import copy
import os
import random

import pytest
import torch

from util import DEV, cosine, oracle_unet, product_unet_from, rel_l2

NEEDS_BWD = [pytest.mark.device]
TUNE = {'min_timestep': 0.5, 'drop_rate': 0.3, 'loss_type': 'local'}


def synthetic_batch(frames=2, hw=16, points=6, seed=3):
    g = torch.Generator().manual_seed(seed)
    tracks = torch.rand(1, frames, points, 2, generator=g) * hw * 8
    tracks[0, 1, 2] = -1.0
    return {'pred_tracks': tracks, 'img_size': (hw * 8, hw * 8),
            'point_embedding': torch.randn(1, points, 1280, generator=g)}


def test_oracle_adapter_training_branch_equals_reference_verbatim():
    from oracle import adapter, ref_import
    if not ref_import.available():
        pytest.skip('/root/reference is not present')
    ref_import.load_reference_models()
    ref_import._load('videoswap.utils.registry', 'videoswap/utils/registry.py')
    ra = ref_import._load('videoswap.models.adapter_model', 'videoswap/models/adapter_model.py')
    chans = [64, 128, 256, 256]
    o = adapter.SparsePointAdapter(1280, chans).train()
    r = ra.SparsePointAdapter(embedding_channels=1280, channels=chans).train()
    r.load_state_dict(o.state_dict(), strict=True)
    batch = synthetic_batch()
    for loss_type in ('local', 'global'):
        random.seed(11)
        so, mo = o(batch['pred_tracks'], batch['img_size'], batch['point_embedding'], drop_rate=0.4, loss_type=loss_type)
        random.seed(11)
        sr, mr = r(batch['pred_tracks'], batch['img_size'], point_embedding=batch['point_embedding'], drop_rate=0.4,
                   loss_type=loss_type)
        assert torch.equal(mo, mr) and (loss_type == 'global' or 0 < float(mo.mean()) < 1)
        for a, b in zip(so, sr):
            assert torch.equal(a, b)


def _models():
    from oracle import adapter as oadapter
    from oracle import unet3d
    from videoswap_amd.adapter import SparsePointAdapter
    cfg = unet3d.tiny_config()
    ora = oracle_unet(cfg)
    for p in ora.parameters():
        p.requires_grad_(False)
    prod = product_unet_from(ora, cfg)
    chans = list(cfg['block_out_channels'])
    oad = oadapter.SparsePointAdapter(1280, chans).train()
    pad = SparsePointAdapter(embedding_channels=1280, channels=chans).train()
    pad.load_state_dict(oad.state_dict(), strict=True)
    return cfg, ora, prod, oad, pad.to(DEV)              # the adapter keeps fp32 master weights


def _inputs(seed=5, frames=2, hw=16):
    g = torch.Generator().manual_seed(seed)
    latents = torch.randn(1, 4, frames, hw, hw, generator=g)
    noise = torch.randn(1, 4, frames, hw, hw, generator=g)
    text = torch.randn(1, 77, 64, generator=g)
    return latents, noise, torch.tensor([731]), text


@pytest.mark.parametrize('dummy', [pytest.param(0, marks=NEEDS_BWD)])
def test_trainer_loss_and_adapter_gradients_match_the_oracle(dummy):
    from oracle import training as otrain
    from videoswap_amd.compat import SD15_SCHEDULER_CONFIG, DDPMScheduler
    from videoswap_amd.trainer import VideoSwapTrainer
    cfg, ora, prod, oad, pad = _models()
    batch = synthetic_batch()
    latents, noise, t, text = _inputs()
    sched = DDPMScheduler(**{k: SD15_SCHEDULER_CONFIG[k] for k in ('num_train_timesteps', 'beta_start', 'beta_end',
                                                                   'beta_schedule')})
    trainer = VideoSwapTrainer(unet=prod, scheduler=sched, adapter=pad, tune_cfg=TUNE)
    # oracle
    random.seed(21)
    lo, _ = otrain.loss_from(ora, oad, latents, noise, t, text, batch, TUNE, otrain.alphas_cumprod())
    lo.backward()
    # product (fp16 storage; the gradient is taken of loss * 1024 and unscaled: fp16 gradient range)
    random.seed(21)
    dev_batch = dict(batch, point_embedding=batch['point_embedding'].to(DEV))
    lp, _ = trainer.loss_from(latents.to(DEV).half(), noise.to(DEV).half(), t.to(DEV), text.to(DEV).half(), dev_batch)
    (lp * 1024.0).backward()
    print(f'loss: product {float(lp.detach()):.5f} oracle {float(lo.detach()):.5f}')
    assert abs(float(lp.detach()) - float(lo.detach())) < 1e-2 * abs(float(lo.detach()))
    worst = 1.0
    for (name, po), pp in zip(oad.named_parameters(), pad.parameters()):
        assert pp.grad is not None and pp.grad.dtype == torch.float32, name
        g = pp.grad.float().cpu() / 1024.0
        c, e = cosine(g, po.grad), rel_l2(g, po.grad)
        worst = min(worst, c)
        assert c > 0.995 and e < 0.1, (name, c, e)
    print(f'adapter parameter gradients: worst cosine vs oracle autograd {worst:.5f}')
    assert all(p.grad is None for p in prod.parameters())


@pytest.mark.parametrize('dummy', [pytest.param(0, marks=NEEDS_BWD)])
def test_trainer_step_updates_only_the_adapter_and_handles_overflow(dummy):
    from videoswap_amd.compat import SD15_SCHEDULER_CONFIG, DDPMScheduler
    from videoswap_amd.synthetic import SyntheticTextEncoder, WhitespaceTokenizer
    from videoswap_amd.trainer import VideoSwapTrainer
    cfg, ora, prod, oad, pad = _models()
    sched = DDPMScheduler(**{k: SD15_SCHEDULER_CONFIG[k] for k in ('num_train_timesteps', 'beta_start', 'beta_end',
                                                                   'beta_schedule')})

    class FixedVae:                                   # VAE parity has its own tests; the step only needs latents
        def encode(self, x):
            z = torch.nn.functional.avg_pool2d(x[:, :1].float().repeat(1, 4, 1, 1), 8).to(x.dtype)
            return type('E', (), {'latent_dist': type('D', (), {'sample': staticmethod(lambda: z)})()})()

    opt = torch.optim.AdamW(pad.parameters(), lr=1e-3)
    trainer = VideoSwapTrainer(vae=FixedVae(), text_encoder=SyntheticTextEncoder(dim=64, dtype=torch.float16, device=DEV),
                               tokenizer=WhitespaceTokenizer(), unet=prod, scheduler=sched, adapter=pad,
                               optimizer=opt, tune_cfg=TUNE)
    batch = dict(synthetic_batch(), images=torch.rand(1, 3, 2, 128, 128).to(DEV) * 2 - 1, prompt=['a cat on a road'])
    batch['point_embedding'] = batch['point_embedding'].to(DEV)
    before_unet = copy.deepcopy(prod.state_dict())
    before_ad = copy.deepcopy(pad.state_dict())
    random.seed(1)
    torch.manual_seed(1)
    loss = trainer.step(batch)
    assert torch.isfinite(loss) and trainer.skipped_steps == 0
    assert all(torch.equal(before_unet[k], v) for k, v in prod.state_dict().items())
    changed = [k for k, v in pad.state_dict().items() if not torch.equal(before_ad[k], v)]
    assert len(changed) == len(before_ad), 'every adapter parameter takes an AdamW step'
    # overflow: an absurd loss scale gives inf gradients; the update is skipped and the scale halves
    trainer.loss_scale = 1e30
    snap = copy.deepcopy(pad.state_dict())
    trainer.step(batch)
    assert trainer.skipped_steps == 1 and trainer.loss_scale == 5e29
    assert all(torch.equal(snap[k], v) for k, v in pad.state_dict().items())


def _train_options(tmp_path, extra=None):
    """The reference's training option file = its test option file without `pretrained_adapter_path`
    (options/train_videoswap/animal/2001_catheadturn_T05_Iter100/...yml; tests/golden/config1_options.json)."""
    import json
    from util import GOLDEN
    with open(os.path.join(GOLDEN, 'config1_options.json')) as f:
        opt = json.load(f)['options']
    opt['path'].pop('pretrained_adapter_path')
    small = [{'type': 'Resize', 'size': 256}, {'type': 'ToTensor'}, {'type': 'Normalize', 'mean': [0.5], 'std': [0.5]}]
    over = {'datasets.num_frames': 3, 'datasets.video_transform': small, 'datasets.dataset_enlarge_ratio': 4,
            'train.total_iter': 3, 'logger.print_freq': 1, 'logger.save_checkpoint_freq': 3, 'val.val_freq': 3,
            'val.editing_config.num_inference_steps': 2, 'val.save_type': 'frame_gif'}
    over.update(extra or {})
    for dotted, v in over.items():
        node = opt
        parts = dotted.split('.')
        for p in parts[:-1]:
            node = node[p]
        node[parts[-1]] = v
    return opt


@pytest.mark.parametrize('dummy', [pytest.param(0, marks=NEEDS_BWD)])
def test_train_flow_from_the_option_file_then_test_with_the_trained_adapter(dummy, tmp_path):
    """train.py's flow on a synthetic workspace in the real on-disk formats (tiny width): three AdamW steps, the
    checkpoint lands at <experiments>/<name>/models/models_3/adapter.pth in the format test.py loads — and the test.py
    flow then runs with it."""
    from videoswap_amd import runner
    from videoswap_amd.workspace import write_synthetic_workspace
    opt = _train_options(tmp_path)
    write_synthetic_workspace(str(tmp_path), opt, width='tiny', total_frames=9)   # 3 frames at stride 4
    cwd = os.getcwd()
    os.environ['VSX_RESULTS_ROOT'] = str(tmp_path / 'experiments')
    os.chdir(tmp_path)
    try:
        res = runner.train(str(tmp_path), copy.deepcopy(opt), None, device=DEV)
        assert len(res['losses']) == 3 and all(l == l and l < 1e4 for l in res['losses'])
        assert res['trainer'].skipped_steps == 0
        ckpt = res['checkpoints'][-1]
        assert ckpt.endswith(os.path.join('models', 'models_3', 'adapter.pth')) and os.path.isfile(ckpt)
        sd = torch.load(ckpt, map_location='cpu')
        from videoswap_amd.adapter import SparsePointAdapter
        assert set(sd) == set(SparsePointAdapter(channels=[64, 128, 256, 256]).state_dict())
        # the reference's test.py flow with the adapter just trained
        topt = copy.deepcopy(opt)
        topt['path']['pretrained_adapter_path'] = ckpt
        topt['mixed_precision'] = 'fp16'
        os.environ['VSX_RESULTS_ROOT'] = str(tmp_path / 'results')
        edited, _ = runner.test(str(tmp_path), topt, None, device=DEV)
        assert set(edited) == {'kitten_to_catA', 'kitten_to_dogB', 'kitten_to_dogA'}
    finally:
        os.chdir(cwd)


def test_noise_schedule_and_lr_schedules_closed_forms():
    """DDPM add_noise / get_velocity and the learning-rate schedules the option files can name (diffusers
    `get_scheduler`): closed forms."""
    import math
    from oracle import training as otrain
    from videoswap_amd.compat import SD15_SCHEDULER_CONFIG, DDPMScheduler
    from videoswap_amd.runner import get_scheduler
    sched = DDPMScheduler(**{k: SD15_SCHEDULER_CONFIG[k] for k in ('num_train_timesteps', 'beta_start', 'beta_end',
                                                                   'beta_schedule')})
    acp = otrain.alphas_cumprod()
    assert torch.allclose(sched.alphas_cumprod, acp)
    g = torch.Generator().manual_seed(0)
    x, n = torch.randn(2, 4, 3, 5, 5, generator=g), torch.randn(2, 4, 3, 5, 5, generator=g)
    t = torch.tensor([10, 900])
    assert torch.allclose(sched.add_noise(x, n, t), otrain.add_noise(x, n, t, acp))
    v = sched.get_velocity(x, n, t)
    a = acp[t].view(-1, 1, 1, 1, 1)
    assert torch.allclose(v, a.sqrt() * n - (1 - a).sqrt() * x)
    # x0 is recovered from (x_t, v): x0 = sqrt(a) x_t - sqrt(1 - a) v
    xt = sched.add_noise(x, n, t)
    assert torch.allclose(a.sqrt() * xt - (1 - a).sqrt() * v, x, atol=1e-5)

    def lrs(name, warm, total, steps):
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.SGD([p], lr=1.0)
        s = get_scheduler(name, opt, warm, total)
        out = []
        for _ in range(steps):
            out.append(s.get_last_lr()[0])
            opt.step()
            s.step()
        return out
    assert lrs('constant', 0, 10, 4) == [1.0] * 4
    assert lrs('constant_with_warmup', 4, 10, 6) == [0.0, 0.25, 0.5, 0.75, 1.0, 1.0]
    lin = lrs('linear', 2, 10, 11)
    assert lin[:3] == [0.0, 0.5, 1.0] and abs(lin[6] - 0.5) < 1e-9 and lin[10] == 0.0
    cos = lrs('cosine', 0, 8, 9)
    assert abs(cos[4] - 0.5) < 1e-9 and abs(cos[2] - 0.5 * (1 + math.cos(math.pi * 0.25))) < 1e-9 and cos[8] < 1e-9
