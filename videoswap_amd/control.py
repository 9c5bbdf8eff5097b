"""Prompt-to-Prompt attention control for the denoising loops, device-resident.

Same controller protocol and step bookkeeping as the reference (videoswap/utils/p2p_utils/):
    controller(probs[b, heads, s, t], is_cross, place_in_unet) -> probs      (attention_store.py:46-57)
    controller.step_callback(latents) -> latents                             (attention_util.py:28-62)
    register_attention_control(pipe, controller)                             (attention_register.py:176-211)
    make_controller(...) -> AttentionRefine | AttentionReplace                (attention_util.py:287-355)

What differs: every stored map and latent stays in HBM (the reference moves ~110 MB of maps to the CPU per
step and deep-copies them back on every edit step: attention_store.py:73,98-100,110; attention_util.py:55-58) and
the per-step archive holds references, not copies.  The arithmetic on the maps is unchanged, including the
reference's step indexing (`49 1 50`, attention_util.py:35-36,92).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .attention import (AttnControlProcessor, AttnProcessor, AttnProcessor2_0, EDLoRA_AttnControlProcessor,
                        EDLoRA_AttnProcessor, XFormersAttnProcessor)

_SMALL = 32 ** 2          # maps with fewer query tokens than this are stored / edited
_PLACES = ('down', 'mid', 'up')


def _empty_store(cross_only=False):
    kinds = ('cross',) if cross_only else ('cross', 'self')
    return {f'{p}_{k}': [] for k in kinds for p in _PLACES}


class EmptyControl:
    def step_callback(self, x_t):
        return x_t

    def between_steps(self):
        return

    def __call__(self, attn, is_cross, place_in_unet):
        return attn


class AttentionControl:
    """attention_store.py:20-67"""

    def __init__(self):
        self.LOW_RESOURCE = False       # True: no CFG batch (inversion); False: edit the conditional half only
        self.cur_step = 0
        self.num_att_layers = -1
        self.cur_att_layer = 0

    num_uncond_att_layers = 0

    def step_callback(self, x_t):
        self.cur_att_layer = 0
        self.cur_step += 1
        self.between_steps()
        return x_t

    def between_steps(self):
        return

    def forward(self, attn, is_cross, place_in_unet):
        raise NotImplementedError

    def __call__(self, attn, is_cross, place_in_unet):
        if self.cur_att_layer >= self.num_uncond_att_layers:
            if self.LOW_RESOURCE:
                attn = self.forward(attn, is_cross, place_in_unet)
            else:
                half = attn.shape[0] // 2
                attn[half:] = self.forward(attn[half:], is_cross, place_in_unet)
        self.cur_att_layer += 1
        return attn

    def reset(self):
        self.cur_step = 0
        self.cur_att_layer = 0


class AttentionStore(AttentionControl):
    """attention_store.py:70-133, with the store kept on the device."""

    def __init__(self):
        super().__init__()
        self.step_store = _empty_store()
        self.attention_store = {}
        self.latents_store = []
        self.attention_store_all_step = []

    get_empty_store = staticmethod(_empty_store)

    @staticmethod
    def get_empty_cross_store():
        return _empty_store(cross_only=True)

    def step_callback(self, x_t):
        x_t = super().step_callback(x_t)
        self.latents_store.append(x_t.detach().clone())
        return x_t

    def forward(self, attn, is_cross, place_in_unet):
        if attn.shape[-2] < _SMALL:
            self.step_store[f"{place_in_unet}_{'cross' if is_cross else 'self'}"].append(attn.detach().clone())
        return attn

    def between_steps(self):
        if len(self.attention_store) == 0:
            self.attention_store = {k: [m.clone() for m in v] for k, v in self.step_store.items()}
        else:
            for key, maps in self.attention_store.items():
                for i in range(len(maps)):
                    maps[i] += self.step_store[key][i]
        self.attention_store_all_step.append(self.step_store)     # this step's maps (not copied)
        self.step_store = _empty_store()

    def get_average_attention(self):
        return {k: [m / self.cur_step for m in v] for k, v in self.attention_store.items()}

    def reset(self):
        super().reset()
        self.step_store = _empty_store()
        self.attention_store_all_step = []
        self.attention_store = {}
        self.latents_store = []


# ------------------------------------------------------------------------------------------------
# word bookkeeping (ptp_utils.py:62-135, seq_aligner.py)
# ------------------------------------------------------------------------------------------------
def _bind_concepts(text, tokenizer):
    cfg = getattr(tokenizer, 'new_concept_cfg', None)
    if cfg:
        from .edlora import bind_concept_prompt
        return bind_concept_prompt(text, cfg)[0]
    return text


def get_word_inds(text, word_place, tokenizer):
    """Token positions (1-based: BOS is 0) of the whitespace word(s) `word_place` (a word, or a word index) in
    `text`; a word split into several sub-tokens yields all of them (ptp_utils.py:62-95)."""
    if isinstance(word_place, str):
        text = _bind_concepts(text, tokenizer)
        word_place = _bind_concepts(word_place, tokenizer)
    words = text.split(' ')
    if isinstance(word_place, str):
        wanted = [i for i, w in enumerate(words) if w == word_place]
    else:
        wanted = [int(word_place)]
    out = []
    if wanted:
        pieces = [tokenizer.decode([tok]).strip('#') for tok in tokenizer.encode(text)][1:-1]
        consumed, ptr = 0, 0
        for i, piece in enumerate(pieces):
            consumed += len(piece)
            if ptr in wanted:
                out.append(i + 1)
            if consumed >= len(words[ptr]):
                ptr += 1
                consumed = 0
    return np.array(out)


def get_time_words_attention_alpha(prompts, num_steps, cross_replace_steps, tokenizer, max_num_words=77):
    """alpha[step, prompt, 1, 1, word] in {0,1}: where cross-attention maps are injected (ptp_utils.py:114-135)."""
    if not isinstance(cross_replace_steps, dict):
        cross_replace_steps = {'default_': cross_replace_steps}
    else:
        cross_replace_steps = dict(cross_replace_steps)
    cross_replace_steps.setdefault('default_', (0.0, 1.0))
    alpha = torch.zeros(num_steps + 1, len(prompts) - 1, max_num_words)

    def paint(bounds, prompt_ind, word_inds=None):
        if isinstance(bounds, (int, float)):
            bounds = (0, bounds)
        start, end = int(bounds[0] * alpha.shape[0]), int(bounds[1] * alpha.shape[0])
        sel = slice(None) if word_inds is None else torch.as_tensor(word_inds, dtype=torch.long)
        alpha[:start, prompt_ind, sel] = 0
        alpha[start:end, prompt_ind, sel] = 1
        alpha[end:, prompt_ind, sel] = 0

    for i in range(len(prompts) - 1):
        paint(cross_replace_steps['default_'], i)
    for word, bounds in cross_replace_steps.items():
        if word == 'default_':
            continue
        for i in range(1, len(prompts)):
            inds = get_word_inds(prompts[i], word, tokenizer)
            if len(inds) > 0:
                paint(bounds, i - 1, inds)
    return alpha.reshape(num_steps + 1, len(prompts) - 1, 1, 1, max_num_words)


def _align_tokens(x, y):
    """Needleman-Wunsch global alignment of two token-id lists with gap 0 / match +1 / mismatch -1 and the
    reference's tie-breaking (left, then up, then diagonal: seq_aligner.py:46-88).  Returns [(j, i or -1)] for every
    token j of y: the token i of x it aligns to."""
    nx, ny = len(x), len(y)
    score = np.zeros((nx + 1, ny + 1), dtype=np.int64)      # gap penalty 0: borders stay 0
    move = np.zeros((nx + 1, ny + 1), dtype=np.int8)
    move[0, 1:], move[1:, 0], move[0, 0] = 1, 2, 4
    for i in range(1, nx + 1):
        for j in range(1, ny + 1):
            left, up = score[i, j - 1], score[i - 1, j]
            diag = score[i - 1, j - 1] + (1 if x[i - 1] == y[j - 1] else -1)
            best = max(left, up, diag)
            score[i, j] = best
            move[i, j] = 1 if best == left else (2 if best == up else 3)
    pairs = []
    i, j = nx, ny
    while i > 0 or j > 0:
        m = move[i, j]
        if m == 3:
            i, j = i - 1, j - 1
            pairs.append((j, i))
        elif m == 1:
            j -= 1
            pairs.append((j, -1))
        elif m == 2:
            i -= 1
        else:
            break
    pairs.reverse()
    return pairs


def get_refinement_mapper(prompts, tokenizer, max_len=77):
    """mapper[p, j] = source token aligned with target token j (-1: new word), alphas[p, j] = 1 where a source
    token exists (seq_aligner.py:91-115)."""
    mappers, alphas = [], []
    src = tokenizer.encode(prompts[0])
    for target in prompts[1:]:
        tgt = tokenizer.encode(target)
        pairs = torch.tensor(_align_tokens(src, tgt), dtype=torch.int64).reshape(-1, 2)
        n = pairs.shape[0]
        alpha = torch.ones(max_len)
        alpha[:n] = pairs[:, 1].ne(-1).float()
        mapper = torch.zeros(max_len, dtype=torch.int64)
        mapper[:n] = pairs[:, 1]
        mapper[n:] = len(tgt) + torch.arange(max_len - len(tgt))
        mappers.append(mapper)
        alphas.append(alpha)
    return torch.stack(mappers), torch.stack(alphas)


def get_replacement_mapper(prompts, tokenizer, max_len=77):
    """Word-swap mapper matrices [p, 77, 77] for equal-length prompts (seq_aligner.py:143-191)."""
    out = []
    wx = prompts[0].split(' ')
    for target in prompts[1:]:
        wy = target.split(' ')
        if len(wx) != len(wy):
            raise ValueError('attention replacement edit can only be applied on prompts with the same length'
                             f' but prompt A has {len(wx)} words and prompt B has {len(wy)} words.')
        changed = [i for i in range(len(wy)) if wy[i] != wx[i]]
        src = [get_word_inds(prompts[0], i, tokenizer) for i in changed]
        tgt = [get_word_inds(target, i, tokenizer) for i in changed]
        m = np.zeros((max_len, max_len))
        i = j = cur = 0
        while i < max_len and j < max_len:
            if cur < len(src) and src[cur][0] == i:
                s, t = src[cur], tgt[cur]
                if len(s) == len(t):
                    m[s, t] = 1
                else:
                    for it in t:
                        m[s, it] = 1 / len(t)
                cur += 1
                i += len(s)
                j += len(t)
            elif cur < len(src):
                m[i, j] = 1
                i += 1
                j += 1
            else:
                m[j, j] = 1
                i += 1
                j += 1
        out.append(torch.from_numpy(m).float())
    return torch.stack(out)


# ------------------------------------------------------------------------------------------------
# spatial_blend.py
# ------------------------------------------------------------------------------------------------
class SpatialBlender:
    """Blending mask from the cross-attention maps of the edited words (spatial_blend.py:20-207)."""

    def __init__(self, prompts, words, substruct_words=None, start_blend=0.2, end_blend=0.8, th=(0.9, 0.9),
                 tokenizer=None, NUM_DDIM_STEPS=None, save_path=None, prompt_choose='source', device='cpu'):
        assert prompt_choose in ('source', 'both')
        self.MAX_NUM_WORDS = 77
        self.NUM_DDIM_STEPS = NUM_DDIM_STEPS
        self.prompt_choose = prompt_choose
        self.alpha_layers = self._word_mask(prompts, words, tokenizer).to(device)
        self.substruct_layers = None if substruct_words is None else \
            self._word_mask(prompts, substruct_words, tokenizer).to(device)
        self.start_blend = int(start_blend * NUM_DDIM_STEPS)
        self.end_blend = int(end_blend * NUM_DDIM_STEPS)
        self.counter = 0
        self.th = th
        self.mask_list = []

    def _word_mask(self, prompts, words, tokenizer):
        layers = torch.zeros(len(prompts), 1, 1, 1, 1, self.MAX_NUM_WORDS)
        for i, (prompt, ws) in enumerate(zip(prompts, words)):
            for w in ([ws] if isinstance(ws, str) else ws):
                ind = get_word_inds(prompt, w, tokenizer)
                layers[i, :, :, :, :, ind] = 1
        return layers

    def get_mask(self, maps, alpha, use_pool, h=None, w=None, x_t=None, step_in_store=None):
        """maps [p, heads*k, F, rh, rw, 77] x alpha [p,1,1,1,1,77] -> bool mask [p, F, h, w] (per-frame max-normalised,
        3x3 max-pooled, thresholded)."""
        if h is None and w is None and x_t is not None:
            h, w = x_t.shape[-2:]
        if maps.dim() == 5:
            alpha = alpha[:, None, ...]
        m = (maps * alpha).sum(-1).mean(1)
        if use_pool:
            m = F.max_pool2d(m, (3, 3), (1, 1), padding=(1, 1))
        mask = F.interpolate(m, size=(h, w))
        mask = mask / mask.max(-2, keepdim=True)[0].max(-1, keepdim=True)[0]
        mask = mask.gt(self.th[1 - int(use_pool)])
        if self.prompt_choose == 'both':
            assert mask.shape[0] == 2, 'If using both source and target prompt'
            mask = mask[:1] + mask
        return mask

    def __call__(self, attention_store, step_in_store=None, target_h=None, target_w=None, x_t=None):
        if target_h is None and target_w is None and x_t is not None:
            target_h, target_w = x_t.shape[-2:]
        self.counter += 1
        picked = attention_store['down_cross'][2:4] + attention_store['up_cross'][:3]
        stacked = []
        for item in picked:
            if item.dim() == 4:
                item = item[None]
            p, c, heads, r, words = item.shape
            assert r % (target_h * target_w) == 0 or (target_h * target_w) % r == 0, 'error shape'
            res_h = int((r * (target_h / target_w)) ** 0.5)
            res_w = int(r / res_h)
            # 'p c h (res_h res_w) w -> p h c res_h res_w w'
            item = item.reshape(p, c, heads, res_h, res_w, words).permute(0, 2, 1, 3, 4, 5)
            stacked.append(item.to(self.alpha_layers.device, dtype=self.alpha_layers.dtype))
        maps = torch.cat(stacked, dim=1)
        alpha = self.alpha_layers[0:1] if self.prompt_choose == 'source' else self.alpha_layers
        mask = self.get_mask(maps, alpha, True, target_h, target_w, step_in_store=step_in_store)
        if self.substruct_layers is not None:
            mask = mask * ~self.get_mask(maps, self.substruct_layers, False, target_h, target_w)
        mask = mask.float()
        self.mask_list.append(mask[0][:, None, :, :])
        if x_t is None:
            return mask
        if x_t.dim() == 5:
            mask = mask[:, None, ...]
        if self.start_blend < self.counter < self.end_blend:
            x_t = self._blend(x_t, mask)
        return x_t

    @staticmethod
    def _blend(x_t, mask):
        """x_t[:1] + mask * (x_t - x_t[:1]) (spatial_blend.py:142); the target row runs on the HIP blend kernel."""
        if x_t.is_cuda and x_t.dtype == torch.float16 and x_t.shape[0] == 2 and x_t.dim() == 5:
            from . import ops
            src, tgt = x_t[0].contiguous(), x_t[1].contiguous()        # [C, F, h, w]
            out = ops.masked_blend(tgt, src, mask[1, 0].to(torch.float16).contiguous())
            return torch.stack([src, out])
        return x_t[:1] + mask.to(x_t.dtype) * (x_t - x_t[:1])


# ------------------------------------------------------------------------------------------------
# attention_util.py
# ------------------------------------------------------------------------------------------------
class AttentionControlEdit(AttentionStore):
    """attention_util.py:21-192: replay the inversion's maps into the edit, blend latents with the source."""

    def __init__(self, prompts, num_steps, cross_replace_steps, self_replace_steps, latent_blend, tokenizer=None,
                 additional_attention_store=None, attention_blend=None, image_height=512, image_width=512,
                 device='cpu'):
        super().__init__()
        self.additional_attention_store = additional_attention_store
        self.batch_size = len(prompts)
        if additional_attention_store is not None:
            self.batch_size = len(prompts) // 2
            assert self.batch_size == 1, 'Only support single video editing with additional attention_store'
        self.device = device
        self.cross_replace_alpha = get_time_words_attention_alpha(prompts, num_steps, cross_replace_steps,
                                                                  tokenizer).to(device)
        if isinstance(self_replace_steps, (int, float)):
            self_replace_steps = (0, self_replace_steps)
        self.num_self_replace = int(num_steps * self_replace_steps[0]), int(num_steps * self_replace_steps[1])
        self.attention_blend = attention_blend
        self.latent_blend = latent_blend
        self.attention_position_counter_dict = {k: 0 for k in _empty_store()}
        self.image_height, self.image_width = image_height, image_width

    def step_callback(self, x_t):
        x_t = super().step_callback(x_t)
        if self.latent_blend is None:
            return x_t
        src = self.additional_attention_store
        step_in_store = len(src.latents_store) - self.cur_step
        inverted = src.latents_store[step_in_store].to(device=x_t.device, dtype=x_t.dtype)
        inv_maps = src.attention_store_all_step[step_in_store]
        blend = _empty_store(cross_only=True)
        for key in blend:
            for i, m in enumerate(inv_maps[key]):
                blend[key].append(torch.stack([m.to(x_t.device), self.attention_store[key][i]]))
        out = self.latent_blend(x_t=torch.cat([inverted, x_t], dim=0), attention_store=blend)
        return out[1:, ...]

    def replace_self_attention(self, attn_base, att_replace, reshaped_mask=None):
        if att_replace.shape[-2] >= _SMALL:
            return att_replace
        attn_base = attn_base.to(att_replace.device, dtype=att_replace.dtype)
        attn_base = attn_base.unsqueeze(0).expand(att_replace.shape[0], *attn_base.shape)
        if reshaped_mask is None:
            return attn_base
        reshaped_mask = reshaped_mask.to(att_replace.dtype)
        return reshaped_mask * att_replace + (1 - reshaped_mask) * attn_base

    def replace_cross_attention(self, attn_base, att_replace):
        raise NotImplementedError

    def forward(self, attn, is_cross, place_in_unet):
        super().forward(attn, is_cross, place_in_unet)
        if attn.shape[-2] >= _SMALL:
            return attn
        key = f"{place_in_unet}_{'cross' if is_cross else 'self'}"
        pos = self.attention_position_counter_dict[key]
        self.attention_position_counter_dict[key] += 1
        src = self.additional_attention_store
        step_in_store = len(src.attention_store_all_step) - self.cur_step - 1
        inv_maps = src.attention_store_all_step[step_in_store]
        attn_base = inv_maps[key][pos]
        if is_cross or (self.num_self_replace[0] <= self.cur_step < self.num_self_replace[1]):
            frames = attn.shape[0] // self.batch_size
            new = attn.reshape(self.batch_size, frames, *attn.shape[1:])
            if is_cross:
                a = self.cross_replace_alpha[self.cur_step].to(new.dtype)
                new = self.replace_cross_attention(attn_base, new) * a + (1 - a) * new
            else:
                mask = None
                if self.attention_blend is not None and new.shape[-2] < _SMALL:
                    rate = int(np.sqrt((self.image_height * self.image_width) / new.shape[-2]))
                    h, w = self.image_height // rate, self.image_width // rate
                    m = self.attention_blend(target_h=h, target_w=w, attention_store=inv_maps,
                                             step_in_store=step_in_store)          # [1, F, h, w]
                    mask = m.permute(1, 0, 2, 3).reshape(m.shape[1], m.shape[0], h * w)[..., None]
                new = self.replace_self_attention(attn_base, new, mask)
            attn = new.reshape(self.batch_size * frames, *new.shape[2:]).to(attn.dtype)
        return attn

    def between_steps(self):
        super().between_steps()
        self.step_store = _empty_store()
        self.attention_position_counter_dict = {k: 0 for k in _empty_store()}


class AttentionReplace(AttentionControlEdit):
    """attention_util.py:195-231"""

    def __init__(self, prompts, num_steps, cross_replace_steps, self_replace_steps, latent_blend=None, tokenizer=None,
                 additional_attention_store=None, attention_blend=None, image_height=512, image_width=512,
                 device='cpu'):
        super().__init__(prompts, num_steps, cross_replace_steps, self_replace_steps, latent_blend, tokenizer,
                         additional_attention_store, attention_blend, image_height, image_width, device)
        self.mapper = get_replacement_mapper(prompts, tokenizer).to(device)

    def replace_cross_attention(self, attn_base, att_replace):
        attn_base = attn_base.to(att_replace.device, dtype=att_replace.dtype)
        mapper = self.mapper.to(attn_base.dtype)
        if attn_base.dim() == 3:
            return torch.einsum('hpw,bwn->bhpn', attn_base, mapper)
        return torch.einsum('thpw,bwn->bthpn', attn_base, mapper)


class AttentionRefine(AttentionControlEdit):
    """attention_util.py:234-284"""

    def __init__(self, prompts, num_steps, cross_replace_steps, self_replace_steps, latent_blend=None, tokenizer=None,
                 additional_attention_store=None, attention_blend=None, image_height=512, image_width=512,
                 device='cpu'):
        super().__init__(prompts, num_steps, cross_replace_steps, self_replace_steps, latent_blend, tokenizer,
                         additional_attention_store, attention_blend, image_height, image_width, device)
        mapper, alphas = get_refinement_mapper(prompts, tokenizer)
        self.mapper = mapper.to(device)
        self.alphas = alphas.to(device).reshape(alphas.shape[0], 1, 1, alphas.shape[1])

    def replace_cross_attention(self, attn_base, att_replace):
        attn_base = attn_base.to(att_replace.device, dtype=att_replace.dtype)
        alphas = self.alphas.to(att_replace.dtype)
        if attn_base.dim() == 3:
            picked = attn_base[:, :, self.mapper].permute(2, 0, 1, 3)
        else:
            picked = attn_base[:, :, :, self.mapper].permute(3, 0, 1, 2, 4)
        return picked * alphas + att_replace * (1 - alphas)


def make_controller(tokenizer, prompts, is_replace_controller, cross_replace_steps, self_replace_steps=0.0,
                    blend_words=None, additional_attention_store=None, blend_th=(0.3, 0.3), NUM_DDIM_STEPS=None,
                    blend_latents=False, blend_self_attention=False, image_height=512, image_width=512,
                    device='cpu'):
    """attention_util.py:287-355"""
    latent_blend = attention_blend = None
    if blend_words is not None and blend_words != 'None':
        if blend_latents:
            latent_blend = SpatialBlender(prompts, blend_words, start_blend=0.2, end_blend=0.8, tokenizer=tokenizer,
                                          th=blend_th, NUM_DDIM_STEPS=NUM_DDIM_STEPS, prompt_choose='both',
                                          device=device)
        if blend_self_attention:
            attention_blend = SpatialBlender(prompts, blend_words, start_blend=0.0, end_blend=2, tokenizer=tokenizer,
                                             th=blend_th, NUM_DDIM_STEPS=NUM_DDIM_STEPS, prompt_choose='source',
                                             device=device)
    cls = AttentionReplace if is_replace_controller else AttentionRefine
    return cls(prompts, NUM_DDIM_STEPS, cross_replace_steps=cross_replace_steps,
               self_replace_steps=self_replace_steps, latent_blend=latent_blend, tokenizer=tokenizer,
               additional_attention_store=additional_attention_store, attention_blend=attention_blend,
               image_height=image_height, image_width=image_width, device=device)


# ------------------------------------------------------------------------------------------------
# attention_register.py:176-211
# ------------------------------------------------------------------------------------------------
def register_attention_control(model, controller):
    """Install control processors on every attn1/attn2 of model.unet in traversal order down -> mid -> up and tell
    the controller how many layers report to it."""
    if controller is None:
        controller = EmptyControl()
        controller.num_att_layers = 0

    def visit(module, n_self, n_cross, place):
        for name, layer in module.named_children():
            if layer.__class__.__name__ == 'Attention' and ('attn1' in name or 'attn2' in name):
                proc = layer.processor
                if isinstance(proc, (EDLoRA_AttnProcessor, EDLoRA_AttnControlProcessor)):
                    layer.set_processor(EDLoRA_AttnControlProcessor(n_cross, place, controller))
                elif isinstance(proc, (AttnProcessor, AttnControlProcessor, XFormersAttnProcessor, AttnProcessor2_0)):
                    layer.set_processor(AttnControlProcessor(place, controller))
                else:
                    raise NotImplementedError(f'cannot wrap attention processor {proc!r}')
                if 'attn1' in name:
                    n_self += 1
                else:
                    n_cross += 1
            else:
                n_self, n_cross = visit(layer, n_self, n_cross, place)
        return n_self, n_cross

    unet = model.unet
    n_self, n_cross = visit(unet.down_blocks, 0, 0, 'down')
    n_self, n_cross = visit(unet.mid_block, n_self, n_cross, 'mid')
    n_self, n_cross = visit(unet.up_blocks, n_self, n_cross, 'up')
    controller.num_att_layers = n_self + n_cross
    return controller
