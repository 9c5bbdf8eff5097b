"""VideoSwapPipeline: DDIM inversion -> classifier-free-guided DDIM sampling with point-adapter residuals,
ED-LoRA prompt embeddings and Prompt-to-Prompt attention control — the two hot loops of
videoswap/pipelines/pipeline_videoswap.py (invert :622-721, __call__ :426-619, validation :272-423) with the
reference's call signatures, driving the HIP-backed UNet.

Per step the host does: one UNet call (B = 1 inversion / B = 2 CFG), one fused CFG + DDIM update kernel
(vsx_cfg_ddim_step) and the controller callback.  Latents, attention maps and the inversion trajectory stay in
HBM (the reference copies ~110 MB of attention maps to the CPU every step: attention_store.py:73,98).

VAE / CLIP / tokenizer are outside the measured path and pluggable: pass `latents` (or a 4-channel `video`
tensor, as the reference's prepare_image_latents accepts, :217-218) and `prompt_embeds` to bypass them.
"""
import copy
from dataclasses import dataclass
from typing import Any, Callable, Dict, List, Optional, Union

import torch

from . import formats, ops
from .compat import PIPELINE_REGISTRY, BaseOutput, DDIMInverseScheduler
from .control import AttentionStore, EmptyControl, make_controller, register_attention_control
from .edlora import (convert_edlora, encode_edlora_prompt, revise_edlora_unet_attention_forward)


@dataclass
class VideoSwapPipelineOutput(BaseOutput):
    videos: Any = None


@dataclass
class VideoSwapInversionOutput(BaseOutput):
    latents: torch.Tensor = None


@PIPELINE_REGISTRY.register()
class VideoSwapPipeline:
    _optional_components = ['inverse_scheduler']

    def __init__(self, vae=None, text_encoder=None, tokenizer=None, unet=None, scheduler=None, adapter=None,
                 inverse_scheduler=None):
        if unet is None or scheduler is None:
            raise ValueError('VideoSwapPipeline needs at least unet and scheduler')
        self.vae, self.text_encoder, self.tokenizer = vae, text_encoder, tokenizer
        self.unet, self.scheduler, self.adapter = unet, scheduler, adapter
        # pipeline_videoswap.py:163 — the inverse scheduler is always rebuilt from the sampler's config
        self.inverse_scheduler = DDIMInverseScheduler.from_config(scheduler.config)
        boc = getattr(getattr(vae, 'config', None), 'block_out_channels', None)
        self.vae_scale_factor = 2 ** (len(boc) - 1) if boc else 8         # pipeline_videoswap.py:164
        from .vae import VaeImageProcessor
        self.image_processor = VaeImageProcessor(vae_scale_factor=self.vae_scale_factor)
        self.new_concept_cfg = None
        self.store_controller = AttentionStore()
        self.empty_controller = EmptyControl()
        self._device = torch.device('cpu')

    @classmethod
    def from_pretrained(cls, pretrained_model_path, unet=None, adapter=None, scheduler=None, torch_dtype=None,
                        vae=None, text_encoder=None, tokenizer=None, **unused):
        """StableDiffusionPipeline.from_pretrained as test.py:73-80 calls it: the components passed in are used as they
        are, the others (`vae/`, `text_encoder/`, `tokenizer/`, `scheduler/`) are loaded from the SD-layout directory."""
        from .clip import CLIPTextModel, load_tokenizer
        from .compat import DDIMScheduler
        from .vae import AutoencoderKL
        if vae is None:
            vae = AutoencoderKL.from_pretrained(pretrained_model_path, subfolder='vae', torch_dtype=torch_dtype)
        if text_encoder is None:
            text_encoder = CLIPTextModel.from_pretrained(pretrained_model_path, subfolder='text_encoder',
                                                         torch_dtype=torch_dtype)
        if tokenizer is None:
            tokenizer = load_tokenizer(pretrained_model_path)
        if scheduler is None:
            scheduler = DDIMScheduler.from_pretrained(pretrained_model_path, subfolder='scheduler')
        if unet is None:
            raise ValueError('VideoSwapPipeline.from_pretrained: pass the AnimateDiffUNet3DModel as `unet` (test.py:55-64)')
        return cls(vae=vae, text_encoder=text_encoder, tokenizer=tokenizer, unet=unet, scheduler=scheduler,
                   adapter=adapter)

    # ---- small protocol surface of diffusers' DiffusionPipeline that test.py touches ----
    def to(self, device=None, dtype=None):
        if device is not None:
            self._device = torch.device(device)
        for m in (self.unet, self.adapter, self.vae, self.text_encoder):
            if m is not None and hasattr(m, 'to'):
                m.to(self._device) if dtype is None else m.to(self._device, dtype)
        return self

    @property
    def device(self):
        return self._device

    @property
    def _execution_device(self):
        return self._device

    def enable_vae_slicing(self):
        if self.vae is not None and hasattr(self.vae, 'enable_slicing'):
            self.vae.enable_slicing()

    def set_new_concept_cfg(self, new_concept_cfg=None):
        self.new_concept_cfg = new_concept_cfg
        if self.tokenizer is not None:
            self.tokenizer.new_concept_cfg = new_concept_cfg

    def progress_bar(self, iterable=None, total=None):
        try:
            from tqdm.auto import tqdm
            return tqdm(iterable, total=total, disable=getattr(self, '_quiet', True))
        except Exception:  # pragma: no cover
            class _Null:
                def __enter__(self): return self
                def __exit__(self, *a): return False
                def update(self, n=1): pass
            return _Null()

    # ---- prompt / latent preparation (outside the measured loop) ----
    def _encode_prompt(self, prompt, device, num_images_per_prompt, do_classifier_free_guidance,
                       negative_prompt=None, prompt_embeds=None, negative_prompt_embeds=None, lora_scale=None):
        """diffusers StableDiffusionPipeline._encode_prompt semantics: [uncond; cond] when guiding."""
        if prompt_embeds is None:
            if self.text_encoder is None or self.tokenizer is None:
                raise ValueError('no text encoder/tokenizer: pass prompt_embeds')
            ids = self.tokenizer(prompt, padding='max_length', max_length=self.tokenizer.model_max_length,
                                 truncation=True, return_tensors='pt').input_ids
            prompt_embeds = self.text_encoder(ids.to(device))[0]
        prompt_embeds = prompt_embeds.to(device=device, dtype=self.unet.dtype)
        if do_classifier_free_guidance:
            if negative_prompt_embeds is None:
                if self.text_encoder is None or self.tokenizer is None:
                    raise ValueError('no text encoder/tokenizer: pass negative_prompt_embeds')
                neg = negative_prompt if negative_prompt is not None else ''
                neg = [neg] * prompt_embeds.shape[0] if isinstance(neg, str) else neg
                ids = self.tokenizer(neg, padding='max_length', max_length=prompt_embeds.shape[1], truncation=True,
                                     return_tensors='pt').input_ids
                negative_prompt_embeds = self.text_encoder(ids.to(device))[0]
            negative_prompt_embeds = negative_prompt_embeds.to(device=device, dtype=self.unet.dtype)
            prompt_embeds = torch.cat([negative_prompt_embeds, prompt_embeds])
        return prompt_embeds

    def prepare_latents(self, batch_size, num_channels_latents, video_length, height, width, dtype, device, generator,
                        latents=None):
        """pipeline_videoswap.py:178-202"""
        shape = (batch_size, num_channels_latents, video_length, height // self.vae_scale_factor,
                 width // self.vae_scale_factor)
        if latents is None:
            latents = torch.randn(shape, generator=generator, device=generator.device if generator else 'cpu',
                                  dtype=torch.float32).to(device=device, dtype=dtype)
        else:
            latents = latents.to(device)
        return latents * self.scheduler.init_noise_sigma

    def prepare_image_latents(self, image, batch_size, dtype, device, generator=None):
        """pipeline_videoswap.py:204-233: a 4-channel input is already a latent; otherwise VAE-encode."""
        image = image.to(device=device, dtype=dtype)
        if image.shape[1] == 4:
            latents = image
        else:
            if self.vae is None:
                raise ValueError('no VAE: pass 4-channel latents as `video`')
            latents = self.vae.encode(image).latent_dist.sample(generator) * self.vae.config.scaling_factor
        f, c, h, w = latents.shape
        return latents.reshape(1, f, c, h, w).permute(0, 2, 1, 3, 4).contiguous()   # '(b f) c h w -> b c f h w'

    # ---------------------------------------------------------------------------------------------
    @torch.no_grad()
    def invert(self, prompt=None, video=None, num_inference_steps=50, guidance_scale=1, generator=None, latents=None,
               prompt_embeds=None, cross_attention_guidance_amount=0.1, output_type='pil', return_dict=True,
               callback=None, callback_steps=1, cross_attention_kwargs=None, controller=None):
        """HOT LOOP #1 (pipeline_videoswap.py:622-721): N x {UNet(B=1) ; inverse DDIM step ; controller callback}."""
        device = self._execution_device
        do_cfg = guidance_scale > 1.0
        if latents is None:
            if video is None:
                raise ValueError('invert: pass `latents` [1,4,F,h,w], a [F,4,h,w] latent tensor or frames as `video`')
            video = self.image_processor.preprocess(video)        # PIL list -> [F,3,H,W] in [-1,1]; tensors pass
            latents = self.prepare_image_latents(video, video.shape[0], self.unet.dtype, device, generator)
        latents = latents.to(device=device, dtype=self.unet.dtype).contiguous()
        prompt_embeds = self._encode_prompt(prompt, device, 1, do_cfg, prompt_embeds=prompt_embeds)

        self.inverse_scheduler.set_timesteps(num_inference_steps, device=device)
        timesteps = self.inverse_scheduler.timesteps
        if hasattr(self.unet, 'prepare_timesteps'):         # the timestep-only work of all steps in three launches
            self.unet.prepare_timesteps(timesteps, device)
        for i, t in enumerate(timesteps):
            model_input = torch.cat([latents] * 2) if do_cfg else latents
            noise_pred = self.unet(model_input, t, encoder_hidden_states=prompt_embeds).sample
            a_t, a_n = self.inverse_scheduler.coefficients(t)
            if do_cfg:
                latents = ops.cfg_ddim_step(latents, noise_pred[:1], noise_pred[1:], guidance_scale, a_t, a_n)
            else:
                latents = ops.cfg_ddim_step(latents, noise_pred, None, 1.0, a_t, a_n)
            if controller is not None:
                latents = controller.step_callback(latents).to(latents.dtype)
            if callback is not None and i % callback_steps == 0:
                callback(i, t, latents)
        inverted = latents.detach().clone()
        if not return_dict:
            return inverted
        return VideoSwapInversionOutput(latents=inverted)

    @torch.no_grad()
    def prepare_ddim_inverted_latents(self, video, prompt, num_inference_steps=50, LOW_RESOURCE=True, use_blend=False,
                                      dtype=torch.float16, prompt_embeds=None):
        """pipeline_videoswap.py:235-252"""
        if use_blend:
            register_attention_control(self, self.store_controller)
            default = self.store_controller.LOW_RESOURCE
            self.store_controller.LOW_RESOURCE = LOW_RESOURCE
        else:
            self.store_controller = None
        latents = self.invert(prompt=prompt, video=video, num_inference_steps=num_inference_steps,
                              controller=self.store_controller, prompt_embeds=prompt_embeds).latents
        if use_blend:
            register_attention_control(self, self.empty_controller)
            self.store_controller.LOW_RESOURCE = default
        return latents

    def get_edit_controller(self, source_prompt, target_prompt, num_inference_steps, blend_words, blend_cfg,
                            image_height, image_width):
        """pipeline_videoswap.py:254-269"""
        th = blend_cfg.get('blend_th', 0.3)
        return make_controller(tokenizer=self.tokenizer, prompts=[source_prompt, target_prompt],
                               NUM_DDIM_STEPS=num_inference_steps, is_replace_controller=False,
                               cross_replace_steps=blend_cfg.get('cross_replace_steps', 0.0),
                               self_replace_steps=blend_cfg.get('self_replace_steps', 0.0), blend_words=blend_words,
                               additional_attention_store=self.store_controller, blend_th=(th, th),
                               blend_self_attention=True, blend_latents=True, image_height=image_height,
                               image_width=image_width, device=self._device)

    # ---------------------------------------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, prompt=None, conditions=None, video_length=None, height=None, width=None,
                 num_inference_steps=50, guidance_scale=7.5, negative_prompt=None, num_images_per_prompt=1, eta=0.0,
                 generator=None, latents=None, prompt_embeds=None, negative_prompt_embeds=None, output_type='pil',
                 return_dict=True, callback=None, callback_steps=1, cross_attention_kwargs=None,
                 guidance_rescale=0.0, controller=None, t2i_guidance_scale=1.0, t2i_start=0.0, t2i_end=1.0, **args):
        """HOT LOOP #2 (pipeline_videoswap.py:426-619): N x {UNet(B=2, adapter residuals inside the t2i window) ;
        CFG ; DDIM step ; blend callback}."""
        if eta != 0.0 or guidance_rescale != 0.0 or num_images_per_prompt != 1:
            raise NotImplementedError('eta / guidance_rescale / num_images_per_prompt (unused by every VideoSwap config)')
        device = self._execution_device
        height = height or self.unet.config.sample_size * self.vae_scale_factor
        width = width or self.unet.config.sample_size * self.vae_scale_factor
        if prompt is not None and isinstance(prompt, list):
            batch_size = len(prompt)
        elif prompt is not None:
            batch_size = 1
        else:
            batch_size = prompt_embeds.shape[0]
        do_cfg = guidance_scale > 1.0

        if self.new_concept_cfg is not None or (prompt_embeds is not None and prompt_embeds.dim() == 4):
            prompt_embeds = encode_edlora_prompt(self, prompt, self.new_concept_cfg, device, num_images_per_prompt,
                                                 do_cfg, negative_prompt, prompt_embeds=prompt_embeds,
                                                 negative_prompt_embeds=negative_prompt_embeds)
        else:
            prompt_embeds = self._encode_prompt(prompt, device, num_images_per_prompt, do_cfg, negative_prompt,
                                                prompt_embeds=prompt_embeds,
                                                negative_prompt_embeds=negative_prompt_embeds)
        prompt_embeds = prompt_embeds.contiguous()

        self.scheduler.set_timesteps(num_inference_steps, device=device)
        timesteps = self.scheduler.timesteps
        if hasattr(self.unet, 'prepare_timesteps'):
            self.unet.prepare_timesteps(timesteps, device)
        if latents is not None and video_length is None:
            video_length = latents.shape[2]
            height, width = latents.shape[3] * self.vae_scale_factor, latents.shape[4] * self.vae_scale_factor
        latents = self.prepare_latents(batch_size * num_images_per_prompt, self.unet.config.in_channels, video_length,
                                       height, width, prompt_embeds.dtype, device, generator, latents)
        latents = latents.to(self.unet.dtype).contiguous()

        adapter_state = None
        if conditions is not None:
            if not isinstance(conditions, dict):
                raise NotImplementedError('image-conditioned T2I adapters are not part of VideoSwap')
            point_embedding = conditions.get('point_embedding')
            if point_embedding is not None:
                point_embedding = point_embedding.to(device, dtype=latents.dtype)
            adapter_state = self.adapter(conditions['pred_tracks'], conditions['img_size'],
                                         point_embedding=point_embedding, index_list=conditions.get('index_list'),
                                         scale=t2i_guidance_scale)
            if do_cfg:      # both CFG branches receive the residual (pipeline_videoswap.py:548-550)
                adapter_state = [self._tag(torch.cat([v] * 2, dim=0)) for v in adapter_state]

        n = len(timesteps)
        for i, t in enumerate(timesteps):
            # both CFG halves see the same latents (pipeline_videoswap.py:556).  For one clip the duplicate is a stride-0 view:
            # the UNet then knows the halves are identical up to the first cross-attention and computes that prefix once
            if do_cfg and latents.shape[0] == 1:
                model_input = latents.expand(2, *latents.shape[1:])
            elif do_cfg:        # several clips denoised together: [uncond clips ; cond clips], two equal halves (stated below)
                model_input = torch.cat([latents] * 2)
            else:
                model_input = latents
            if adapter_state is not None and n * t2i_start <= i <= n * t2i_end:
                t2i_residual = list(adapter_state)      # fresh list: the UNet pops from it
            else:
                t2i_residual = None
            # (an explicit keyword, not an attribute on the tensor: any op in between would drop the latter silently, ADVICE r5;
            # only this package's UNet knows the keyword — a foreign UNet object is called with the reference's own signature)
            extra = {'cfg_halves_equal': True} if do_cfg and latents.shape[0] > 1 and getattr(self.unet, 'vsx_cfg_keyword', False) else {}
            noise_pred = self.unet(model_input, t, encoder_hidden_states=prompt_embeds,
                                   cross_attention_kwargs=cross_attention_kwargs,
                                   down_block_additional_residuals=t2i_residual, return_dict=False, **extra)[0]
            a_t, a_n = self.scheduler.coefficients(t)
            if do_cfg:
                latents = ops.cfg_ddim_step(latents, noise_pred[:batch_size], noise_pred[batch_size:], guidance_scale,
                                            a_t, a_n)
            else:
                latents = ops.cfg_ddim_step(latents, noise_pred, None, 1.0, a_t, a_n)
            if controller is not None:
                latents = controller.step_callback(latents).to(latents.dtype).contiguous()
            if callback is not None and i % callback_steps == 0:
                callback(i, t, latents)

        if output_type == 'latent' or self.vae is None:
            video = latents
        else:
            b, c, f, h, w = latents.shape
            flat = latents.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)       # 'b c f h w -> (b f) c h w'
            video = self.vae.decode(flat / self.vae.config.scaling_factor, return_dict=False)[0]
            video = self.image_processor.postprocess(video, output_type=output_type)
        if not return_dict:
            return video
        return VideoSwapPipelineOutput(videos=video)

    @staticmethod
    def _tag(t):
        t.vsx_nhwc = True
        return t

    # ---------------------------------------------------------------------------------------------
    @torch.no_grad()
    def validation(self, source_video, source_conditions, source_prompt, editing_config, dtype=torch.float16,
                   train_dataset=None, save_dir=None, source_prompt_embeds=None, prompt_embeds_fn=None,
                   lora_loader: Optional[Callable[[str], Dict]] = None):
        """pipeline_videoswap.py:272-423: invert once, then per editing prompt: merge ED-LoRA -> build the edit
        controller -> guided sampling -> restore the weights.

        `source_video`: [F,4,h,w] latents (or images when a VAE is plugged in).  `prompt_embeds_fn(prompt)` may
        supply embeddings when no text encoder is plugged in; `lora_loader(path)` replaces torch.load for tests.
        """
        use_inv = editing_config['use_invertion_latents']
        use_blend = editing_config.get('use_blend', False)
        steps = editing_config['num_inference_steps']
        embeds = (lambda p: None) if prompt_embeds_fn is None else prompt_embeds_fn

        ddim_latents = None
        if use_inv:
            ddim_latents = self.prepare_ddim_inverted_latents(
                video=source_video, prompt=source_prompt, num_inference_steps=steps, LOW_RESOURCE=True,
                use_blend=use_blend, dtype=dtype,
                prompt_embeds=source_prompt_embeds if source_prompt_embeds is not None else embeds(source_prompt))
            ddim_latents = ddim_latents.to(dtype=dtype)

        # device-side snapshot of the weights the LoRA merges mutate (:303-305 deep-copies both state dicts up front, 2.5 GB
        # and ~1400 tensors for the UNet: 0.24 s per clip at the benchmark size; here `convert_edlora` saves a weight the
        # first time a merge is about to change it, and the restore below writes exactly those back)
        pretrained = {'unet': {}, 'text_encoder': {}}

        if torch.is_tensor(source_video):
            video_length = source_video.shape[0]
            height, width = source_video.shape[-2] * self.vae_scale_factor, source_video.shape[-1] * self.vae_scale_factor
            if source_video.shape[1] != 4:
                height, width = source_video.shape[-2], source_video.shape[-1]
        else:
            video_length = len(source_video)
            width, height = source_video[0].size

        edited = {}
        for key, swap_cfg in editing_config['editing_prompts'].items():
            lora_path = swap_cfg.get('lora_path', None)
            if lora_path is not None:
                lora_path, lora_alpha = lora_path.split('---')
                enable_edlora = 'edlora' in lora_path
                state = (lora_loader or formats.load_checkpoint)(lora_path)
                _, new_concept_cfg = convert_edlora(self, state, enable_edlora=enable_edlora, alpha=float(lora_alpha),
                                                    snapshot=pretrained)
                if enable_edlora:
                    revise_edlora_unet_attention_forward(self.unet)
                    self.set_new_concept_cfg(new_concept_cfg)

            if source_conditions is not None and swap_cfg.get('tap_path') and train_dataset is not None:
                conditions = train_dataset.get_conditions(swap_cfg['tap_path'])
            else:
                conditions = copy.deepcopy(source_conditions)
            if conditions is not None:
                if swap_cfg.get('select_point'):
                    conditions['index_list'] = [conditions['point_name2id'][n] for n in swap_cfg['select_point']]
                else:
                    conditions['index_list'] = None

            source_subject, target_subject = [s.strip() for s in swap_cfg['replace'].split('->')]
            assert source_subject in source_prompt, 'source subject need in source prompt'
            target_prompt = source_prompt.replace(source_subject, target_subject)
            if 'replace_other' in swap_cfg:
                source_other, target_other = [s.strip() for s in swap_cfg['replace_other'].split('->')]
                assert source_other in target_prompt, 'source subject need in source prompt'
                target_prompt = target_prompt.replace(source_other, target_other)

            if use_blend:
                blend_words = [source_subject.split(' '), target_subject.split(' ')]
                controller = self.get_edit_controller(source_prompt, target_prompt, steps, blend_words=blend_words,
                                                      blend_cfg=swap_cfg.get('blend_cfg', {}), image_height=height,
                                                      image_width=width)
                register_attention_control(self, controller)
            else:
                controller = None

            out = self(prompt=target_prompt, conditions=conditions, negative_prompt=swap_cfg.get(
                           'negative_prompt', editing_config.get('negative_prompt', None)),
                       num_inference_steps=steps, video_length=video_length, height=height, width=width,
                       guidance_scale=swap_cfg.get('guidance_scale', editing_config.get('guidance_scale', 7.5)),
                       num_images_per_prompt=1, latents=ddim_latents, controller=controller,
                       prompt_embeds=embeds(target_prompt), negative_prompt_embeds=embeds(''),
                       t2i_guidance_scale=swap_cfg.get('t2i_guidance_scale', editing_config.get('t2i_guidance_scale', 1.0)),
                       t2i_start=editing_config.get('t2i_start', 0.0), t2i_end=editing_config.get('t2i_end', 1.0),
                       output_type='latent' if self.vae is None else 'pil')
            edited[key] = out.videos.clone() if torch.is_tensor(out.videos) else copy.deepcopy(out.videos)

            if lora_path is not None:     # :417-420 restore
                if pretrained['unet']:
                    self.unet.load_state_dict(pretrained['unet'], strict=False)
                if pretrained['text_encoder'] and self.text_encoder is not None:
                    self.text_encoder.load_state_dict(pretrained['text_encoder'], strict=False)
                self.set_new_concept_cfg(None)
        return edited
