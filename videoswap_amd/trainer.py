"""VideoSwapTrainer — the adapter training step (videoswap/pipelines/trainer_videoswap.py:15-97; driver loop
train.py:107-224).  Only the SparsePointAdapter learns (AdamW on its 1.1 M parameters, train.py:112); the UNet, the VAE
and the text encoder are frozen, and the gradient reaches the adapter through the UNet: forward and backward run on
the HIP kernels (videoswap_amd/autograd.py), PyTorch keeps the autograd tape, the optimizer and the loss arithmetic
of the reference (`F.mse_loss` on fp32 copies, trainer_videoswap.py:92-93).

`accelerate` is replaced by what it does here for one process: fp16 activations with a dynamic loss scale (its
GradScaler: scale the loss, unscale the adapter gradients, skip the update and halve the scale on overflow, double it
every `growth_interval` clean steps), and — for several GPUs — an all-reduce (mean) of the 1.1 M gradient values over
torch.distributed (RCCL), which is all DDP does for a 4.4 MB parameter set."""
import random

import torch
import torch.nn.functional as F

from .compat import PIPELINE_REGISTRY
from .pipeline import VideoSwapPipeline


def generate_sampleT(T_boundary, largeT_prob=1.0):
    """trainer_videoswap.py:15-20 (same RNG consumption)"""
    if random.random() <= largeT_prob:
        return random.uniform(T_boundary, 1)
    return random.uniform(0, T_boundary)


@PIPELINE_REGISTRY.register()
class VideoSwapTrainer(VideoSwapPipeline):
    def __init__(self, vae=None, text_encoder=None, tokenizer=None, unet=None, scheduler=None, adapter=None, **kwargs):
        sampler = kwargs.pop('sampler', None)
        # the pipeline base derives its inverse DDIM scheduler from a DDIM config; the trainer's scheduler is a DDPM one
        super().__init__(vae, text_encoder, tokenizer, unet, sampler if sampler is not None else _ddim_like(scheduler),
                         adapter)
        self.scheduler = scheduler
        self.weight_dtype = torch.float16
        self.optimizer = self.lr_scheduler = self.accelerator = None
        self.max_grad_norm = 1.0
        self.tune_cfg = None
        self.loss_scale, self.growth_interval, self._clean_steps = 65536.0, 2000, 0
        self.skipped_steps = 0
        for name, module in kwargs.items():            # trainer_videoswap.py:30-31
            setattr(self, name, module)
        for frozen in (self.unet, self.vae, self.text_encoder):
            if frozen is not None and hasattr(frozen, 'parameters'):
                for p in frozen.parameters():
                    p.requires_grad_(False)

    # ---- the pieces of `step`, separately callable (tests drive them with given latents / noise) ----------------
    def encode_images(self, images):
        """trainer_videoswap.py:38-44: [b, c, f, h, w] frames -> scaled latents [b, 4, f, h/8, w/8]"""
        b, c, f, h, w = images.shape
        flat = images.to(self.weight_dtype).permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
        with torch.no_grad():
            z = self.vae.encode(flat).latent_dist.sample()
        return z.reshape(b, f, *z.shape[1:]).permute(0, 2, 1, 3, 4).contiguous() * 0.18215

    def sample_timesteps(self, bsz, device):
        n = self.scheduler.config.num_train_timesteps
        t = [int(generate_sampleT(self.tune_cfg['min_timestep']) * n) for _ in range(bsz)]
        return torch.tensor(t).to(device).long()

    def encode_prompt(self, prompt, device):
        ids = self.tokenizer(prompt, padding='max_length', max_length=self.tokenizer.model_max_length, truncation=True,
                             return_tensors='pt').input_ids.to(device)
        with torch.no_grad():
            return self.text_encoder(ids)[0]

    def loss_from(self, latents, noise, timesteps, encoder_hidden_states, batch):
        """trainer_videoswap.py:57-93 from the noising on: -> (loss fp32 scalar with the autograd graph, model_pred)"""
        noisy = self.scheduler.add_noise(latents, noise, timesteps)
        adapter_state, loss_mask = self.adapter(batch['pred_tracks'], batch['img_size'],
                                                point_embedding=batch['point_embedding'],
                                                drop_rate=self.tune_cfg['drop_rate'],
                                                loss_type=self.tune_cfg['loss_type'])
        loss_mask = loss_mask.unsqueeze(0).permute(0, 2, 1, 3, 4).to(latents.device)       # 'b f c h w -> b c f h w'
        pred = self.unet(noisy.to(self.weight_dtype), timesteps, encoder_hidden_states.to(self.weight_dtype),
                         down_block_additional_residuals=adapter_state).sample
        kind = self.scheduler.config.prediction_type
        if kind == 'epsilon':
            target = noise
        elif kind == 'v_prediction':
            target = self.scheduler.get_velocity(latents, noise, timesteps)
        else:
            raise ValueError(f'Unknown prediction type {kind}')
        loss = F.mse_loss(pred.float(), target.float(), reduction='none')
        loss = ((loss * loss_mask).sum([1, 2, 3, 4]) / loss_mask.sum([1, 2, 3, 4])).mean()
        return loss, pred

    def backward_and_update(self, loss):
        """accelerator.backward + optimizer / lr-scheduler step.  With an `accelerate.Accelerator` handed in (the
        reference's train.py does) its calls are made verbatim (trainer_videoswap.py:95-101: its GradScaler does the
        loss scaling); without one, the dynamic loss scale of the module docstring.  Returns True when the update was
        applied."""
        acc = self.accelerator
        if acc is not None:
            acc.backward(loss)
            if acc.sync_gradients:
                acc.clip_grad_norm_(self.unet.parameters(), self.max_grad_norm)
            self.optimizer.step()
            self.lr_scheduler.step()
            self.optimizer.zero_grad()
            return True
        (loss * self.loss_scale).backward()
        # every rank must take the SAME skip / step decision (and keep the same loss scale): the scaled gradients of a
        # FIXED parameter list (missing gradients as zeros) are averaged first, finiteness is tested on the reduced
        # buffer — an inf / nan on one rank reaches all of them through the sum
        params = [p for p in self.adapter.parameters() if p.requires_grad]
        # ... followed by one has-gradient flag per parameter: a parameter that took part in the loss on NO rank keeps
        # p.grad = None, so that an optimizer with weight decay / momentum skips it as it would on one GPU
        has = torch.tensor([0.0 if p.grad is None else 1.0 for p in params], dtype=torch.float32, device=params[0].device)
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).float() for p in params]
                         + [has])
        self._all_reduce_mean(flat)
        has = flat[-len(params):] > 0
        flat = flat[:-len(params)]
        finite = bool(torch.isfinite(flat).all())
        if finite:
            flat.mul_(1.0 / self.loss_scale)
            o = 0
            for p, h in zip(params, has.tolist()):
                n = p.numel()
                if h:
                    if p.grad is None:
                        p.grad = torch.empty_like(p)
                    p.grad.copy_(flat[o:o + n].view_as(p))
                o += n
            # trainer_videoswap.py:96-97 clips the UNet's parameters, which have no gradient: kept as the no-op it is
            self.optimizer.step()
            if self.lr_scheduler is not None:
                self.lr_scheduler.step()
            self._clean_steps += 1
            if self._clean_steps % self.growth_interval == 0:
                self.loss_scale *= 2.0
        else:
            self.loss_scale *= 0.5
            self._clean_steps = 0
            self.skipped_steps += 1
        self.optimizer.zero_grad()
        return finite

    @staticmethod
    def _all_reduce_mean(flat):
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        dist.all_reduce(flat)
        flat.div_(dist.get_world_size())

    # ---- trainer_videoswap.py:33-97 ---------------------------------------------------------------------------------
    def step(self, batch=None):
        batch = batch or {}
        self.unet.train()
        self.adapter.train()
        latents = self.encode_images(batch['images'])
        noise = torch.randn_like(latents)
        timesteps = self.sample_timesteps(latents.shape[0], latents.device)
        ehs = self.encode_prompt(batch['prompt'], latents.device)
        loss, _ = self.loss_from(latents, noise, timesteps, ehs, batch)
        self.backward_and_update(loss)
        return loss.detach()


def _ddim_like(scheduler):
    """A DDIM sampler with the training scheduler's noise schedule (validation during training samples with DDIM,
    train.py:95-103)."""
    from .compat import SD15_SCHEDULER_CONFIG, DDIMScheduler
    cfg = dict(SD15_SCHEDULER_CONFIG)
    for k in ('num_train_timesteps', 'beta_start', 'beta_end', 'beta_schedule'):
        if scheduler is not None and k in scheduler.config:
            cfg[k] = scheduler.config[k]
    return DDIMScheduler(**cfg)
