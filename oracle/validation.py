"""TEST INFRASTRUCTURE — the reference's whole inference flow on the oracle modules: `OraclePipeline` restates
VideoSwapPipeline (pipeline_videoswap.py:87-172 assembly, :204-252 latents / inversion, :254-269 edit controller,
:272-423 validation, :426-619 guided sampling) over oracle/unet3d.py, oracle/adapter.py, oracle/vae.py,
oracle/clip.py and the restated DDIM schedulers, in plain PyTorch on whatever device the modules live on (fp32).

The Prompt-to-Prompt controllers and the ED-LoRA host functions are the product's device-agnostic host logic
(videoswap_amd/control.py, videoswap_amd/edlora.py), pinned against the reference's files in tests/test_control.py and
tests/test_reference_pins.py; here they are only CALLED.  `oracle_classes()` gives the registry-name -> class map that
videoswap_amd.runner.test(..., classes=...) and the drop-in test of the reference's `test.py` plug in.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import anything under oracle/."""
import copy
import json
import os

import torch

from . import adapter as oadapter
from . import clip as oclip
from . import pipeline as opipe
from . import unet3d
from . import vae as ovae
from .diffusers_restated import DDIMInverseScheduler, DDIMScheduler


class OracleUNet(unet3d.AnimateDiffUNet3DModel):
    @classmethod
    def from_pretrained_2d(cls, pretrained_model_path, subfolder=None, unet_additional_kwargs=None):
        """unet.py:483-523"""
        path = os.path.join(pretrained_model_path, subfolder) if subfolder else pretrained_model_path
        with open(os.path.join(path, 'config.json')) as f:
            cfg = {k: v for k, v in json.load(f).items() if not k.startswith('_')}
        for k in ('down_block_types', 'up_block_types', 'mid_block_type'):
            cfg.pop(k, None)
        cfg.update(unet_additional_kwargs or {})
        import inspect
        accepted = set(inspect.signature(unet3d.AnimateDiffUNet3DModel.__init__).parameters)
        model = cls(**{k: v for k, v in cfg.items() if k in accepted})
        state = torch.load(os.path.join(path, 'diffusion_pytorch_model.bin'), map_location='cpu')
        model.load_state_dict({k: v.float() for k, v in state.items()}, strict=False)
        return model.eval()

    def load_state_dict(self, state_dict, strict=True):
        return super().load_state_dict({k: v.float() for k, v in state_dict.items()}, strict=strict)


class OracleAdapter(oadapter.SparsePointAdapter):
    def load_state_dict(self, state_dict, strict=True):
        return super().load_state_dict({k: v.float() for k, v in state_dict.items()}, strict=strict)

    def __call__(self, point_tracker, size, point_embedding=None, index_list=None, scale=1.0, **unused):
        tracks = point_tracker if point_tracker.dim() == 4 else point_tracker[None]
        emb = point_embedding if point_embedding.dim() == 3 else point_embedding[None]
        dev = next(self.parameters()).device
        # the reference holds the tracks in the latent dtype (fp16) before the adapter sees them
        maps = super().forward(tracks.half().float().cpu(), size, emb.float().cpu(), index_list=index_list) \
            if dev.type == 'cpu' else [m.to(dev) for m in copy.deepcopy(self).cpu()(tracks, size, emb, index_list, 1.0)]
        return [m.to(dev) * scale for m in maps]

    def forward(self, *a, **k):      # nn.Module.__call__ is bypassed above
        raise RuntimeError('call the adapter object')


class _VaeFacade:
    """diffusers AutoencoderKL call surface over oracle.vae.AutoencoderKL"""

    def __init__(self, model, cfg):
        self.model, self.config = model, type('Cfg', (), dict(cfg))()
        self.dtype = torch.float32

    def to(self, *a, **k):
        self.model.to(*[x for x in a if not isinstance(x, torch.dtype)])
        return self

    def enable_slicing(self):
        pass

    def encode_sample(self, x, generator):
        moments = self.model.moments(x)
        dev = generator.device if generator is not None else moments.device
        noise = torch.randn(moments[:, :moments.shape[1] // 2].shape, generator=generator, device=dev,
                            dtype=torch.float32).to(moments.device)
        mean, logvar = torch.chunk(moments, 2, dim=1)
        return mean + torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0)) * noise

    def decode(self, z):
        return self.model.decode(z)


class OraclePipeline:
    def __init__(self, vae, text_encoder, tokenizer, unet, scheduler, adapter):
        self.vae, self.text_encoder, self.tokenizer = vae, text_encoder, tokenizer
        self.unet, self.scheduler, self.adapter = unet, scheduler, adapter
        self.inverse_scheduler = DDIMInverseScheduler.from_config(scheduler.config)
        self.new_concept_cfg = None
        self.device = torch.device('cpu')
        from videoswap_amd import control
        self.store_controller = control.AttentionStore()

    @classmethod
    def from_pretrained(cls, path, unet=None, adapter=None, scheduler=None, torch_dtype=None, **unused):
        from videoswap_amd.clip import load_tokenizer
        from videoswap_amd.formats import scheduler_config_from_pretrained
        with open(os.path.join(path, 'vae', 'config.json')) as f:
            vcfg = {k: v for k, v in json.load(f).items() if not k.startswith('_')}
        vae = ovae.AutoencoderKL(**vcfg).eval()
        vae.load_state_dict({k: v.float() for k, v in torch.load(os.path.join(path, 'vae', 'diffusion_pytorch_model.bin'),
                                                                 map_location='cpu').items()})
        with open(os.path.join(path, 'text_encoder', 'config.json')) as f:
            ccfg = json.load(f)
        enc = oclip.CLIPTextModel(**ccfg).eval()
        enc.load_state_dict({k: v.float() for k, v in torch.load(os.path.join(path, 'text_encoder', 'pytorch_model.bin'),
                                                                 map_location='cpu').items()})
        enc.dtype = torch.float32
        sch = DDIMScheduler(**scheduler_config_from_pretrained(path, 'scheduler'))
        return cls(_VaeFacade(vae, vcfg), enc, load_tokenizer(path), unet, sch, adapter)

    def to(self, device=None, dtype=None):
        self.device = torch.device(device)
        for m in (self.unet, self.adapter, self.text_encoder):
            m.to(self.device)
        self.vae.to(self.device)
        return self

    def enable_vae_slicing(self):
        pass

    def set_new_concept_cfg(self, cfg=None):
        self.new_concept_cfg = cfg
        self.tokenizer.new_concept_cfg = cfg

    # ---- text ----
    def _ids(self, prompts, max_length=None):
        return self.tokenizer(prompts, padding='max_length', max_length=max_length or self.tokenizer.model_max_length,
                              truncation=True, return_tensors='pt').input_ids.to(self.device)

    def encode(self, prompt):
        return self.text_encoder(self._ids(prompt))[0]

    # ---- frames <-> latents (pipeline_videoswap.py:204-233, 603-610) ----
    def frames_to_latents(self, frames, generator=None):
        from videoswap_amd.vae import VaeImageProcessor
        x = VaeImageProcessor(8).preprocess(frames).to(self.device)
        z = self.vae.encode_sample(x, generator) * self.vae.config.scaling_factor
        f, c, h, w = z.shape
        return z.reshape(1, f, c, h, w).permute(0, 2, 1, 3, 4).contiguous()

    def latents_to_frames(self, latents):
        from videoswap_amd.vae import VaeImageProcessor
        b, c, f, h, w = latents.shape
        flat = latents.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
        return VaeImageProcessor(8).postprocess(self.vae.decode(flat / self.vae.config.scaling_factor), 'pil')

    # ---- validation (pipeline_videoswap.py:272-423) ----
    @torch.no_grad()
    def validation(self, source_video, source_conditions, source_prompt, editing_config, dtype=None, train_dataset=None,
                   save_dir=None, return_latents=False, vae_generator=None):
        from videoswap_amd import control
        from videoswap_amd.edlora import convert_edlora, encode_edlora_prompt
        steps = editing_config['num_inference_steps']
        use_blend = editing_config.get('use_blend', False)
        latents = self.frames_to_latents(source_video, vae_generator)
        store = None
        if use_blend:
            store = self.store_controller = control.AttentionStore()
            store.LOW_RESOURCE = True
            opipe.register_control(self.unet, store, edlora=self.new_concept_cfg is not None)
        inverted = opipe.invert(self.unet, latents, self.encode(source_prompt), steps, controller=store)
        if use_blend:
            opipe.register_control(self.unet, control.EmptyControl(), edlora=self.new_concept_cfg is not None)
            store.LOW_RESOURCE = False
        snapshot_unet = copy.deepcopy(self.unet.state_dict())
        snapshot_text = copy.deepcopy(self.text_encoder.state_dict())
        snapshot_text.pop('text_model.embeddings.token_embedding.weight')
        width, height = source_video[0].size
        out = {}
        for key, cfg in editing_config['editing_prompts'].items():
            lora_path = cfg.get('lora_path')
            enable_edlora = False
            if lora_path is not None:
                lora_path, alpha = lora_path.split('---')
                enable_edlora = 'edlora' in lora_path
                _, concept_cfg = convert_edlora(self, torch.load(lora_path, map_location='cpu'),
                                                enable_edlora=enable_edlora, alpha=float(alpha))
                if enable_edlora:
                    opipe.use_edlora(self.unet)
                    self.set_new_concept_cfg(concept_cfg)
            if source_conditions is not None and cfg.get('tap_path'):
                conditions = train_dataset.get_conditions(cfg['tap_path'])
            else:
                conditions = copy.deepcopy(source_conditions)
            if conditions is not None:
                conditions['index_list'] = [conditions['point_name2id'][n] for n in cfg['select_point']] \
                    if cfg.get('select_point') else None
            src, tgt = [s.strip() for s in cfg['replace'].split('->')]
            assert src in source_prompt
            target_prompt = source_prompt.replace(src, tgt)
            controller = None
            if use_blend:
                bc = cfg.get('blend_cfg', {})
                th = bc.get('blend_th', 0.3)
                controller = control.make_controller(
                    tokenizer=self.tokenizer, prompts=[source_prompt, target_prompt], NUM_DDIM_STEPS=steps,
                    is_replace_controller=False, cross_replace_steps=bc.get('cross_replace_steps', 0.0),
                    self_replace_steps=bc.get('self_replace_steps', 0.0), blend_words=[src.split(' '), tgt.split(' ')],
                    additional_attention_store=store, blend_th=(th, th), blend_self_attention=True,
                    blend_latents=True, image_height=height, image_width=width, device=self.device)
                opipe.register_control(self.unet, controller, edlora=self.new_concept_cfg is not None)
            negative = cfg.get('negative_prompt', editing_config.get('negative_prompt', None))
            if self.new_concept_cfg is not None:
                emb = encode_edlora_prompt(self, target_prompt, self.new_concept_cfg, self.device, 1, True, negative)
                text, neg = emb[1:], emb[:1, 0]
            else:
                text, neg = self.encode(target_prompt), self.encode(negative or '')
            state = None
            if conditions is not None:
                state = self.adapter(conditions['pred_tracks'], conditions['img_size'],
                                     point_embedding=conditions['point_embedding'].to(self.device),
                                     index_list=conditions.get('index_list'),
                                     scale=cfg.get('t2i_guidance_scale', editing_config.get('t2i_guidance_scale', 1.0)))
            sampled = opipe.sample(self.unet, inverted, text, neg, steps,
                                   guidance=cfg.get('guidance_scale', editing_config.get('guidance_scale', 7.5)),
                                   controller=controller, adapter_state=state,
                                   t2i_start=editing_config.get('t2i_start', 0.0), t2i_end=editing_config.get('t2i_end', 1.0))
            out[key] = sampled.clone() if return_latents else self.latents_to_frames(sampled)
            if lora_path is not None:
                self.unet.load_state_dict(snapshot_unet)
                self.text_encoder.load_state_dict(snapshot_text, strict=False)
                self.set_new_concept_cfg(None)
                opipe.reset_processors(self.unet)
        return out


def oracle_classes():
    return {'AnimateDiffUNet3DModel': OracleUNet, 'SparsePointAdapter': OracleAdapter, 'VideoSwapPipeline': OraclePipeline}
