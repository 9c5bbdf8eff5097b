"""TEST INFRASTRUCTURE — the adapter training step (videoswap/pipelines/trainer_videoswap.py:33-97) restated on the
oracle modules in plain PyTorch fp32 with PyTorch's own autograd: noising (DDPM `add_noise`, restated below), adapter in
training mode (dropout + loss mask), frozen UNet, masked MSE.  The product's `VideoSwapTrainer.loss_from` and its
gradients (HIP kernels forward and backward) are compared with this.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import anything under oracle/."""
import torch
import torch.nn.functional as F


def alphas_cumprod(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012):
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


def add_noise(latents, noise, timesteps, acp):
    """diffusers DDPMScheduler.add_noise: sqrt(abar_t) x0 + sqrt(1 - abar_t) eps, per batch item"""
    a = acp[timesteps].view(-1, 1, 1, 1, 1)
    return a.sqrt() * latents + (1 - a).sqrt() * noise


def loss_from(unet, adapter, latents, noise, timesteps, encoder_hidden_states, batch, tune_cfg, acp):
    """trainer_videoswap.py:57-93 (epsilon prediction) -> loss (fp32, autograd graph into the adapter's parameters)"""
    noisy = add_noise(latents, noise, timesteps, acp)
    state, loss_mask = adapter(batch['pred_tracks'], batch['img_size'], batch['point_embedding'],
                               drop_rate=tune_cfg['drop_rate'], loss_type=tune_cfg['loss_type'])
    loss_mask = loss_mask.unsqueeze(0).permute(0, 2, 1, 3, 4)
    pred = unet(noisy, timesteps, encoder_hidden_states, down_block_additional_residuals=state).sample
    loss = F.mse_loss(pred.float(), noise.float(), reduction='none')
    return ((loss * loss_mask).sum([1, 2, 3, 4]) / loss_mask.sum([1, 2, 3, 4])).mean(), pred
