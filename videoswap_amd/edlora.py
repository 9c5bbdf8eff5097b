"""ED-LoRA concept weights on the denoising path (videoswap/utils/edlora_util.py, convert_edlora_to_diffusers.py).

* `revise_edlora_unet_attention_forward` installs `EDLoRA_AttnProcessor(i)` on the i-th cross-attention (16 of them,
  traversal order down -> mid -> up) so that layer i attends to prompt embedding `[:, i]` of a [B, 16, 77, 768] tensor;
* `encode_edlora_prompt` builds that tensor (16 per-layer prompts through the text encoder, negative prompt
  repeated over the layer axis);
* `merge_lora_into_weight` / `convert_edlora` fold W += alpha * up @ down into the UNet / text-encoder weights by
  state-dict key, exactly as the reference does.  The HIP kernels read the parameters in place, so a merge or a
  `load_state_dict` restore is picked up by the next launch (conv weights keep their OHWI storage through
  `copy_`; the only packed copy, conv_in's padded weight, is keyed on the parameter version).
"""
import copy

import torch

from .attention import EDLoRA_AttnProcessor

UNET_LORA_KEYS = ('to_q.weight', 'to_k.weight', 'to_v.weight', 'to_out.0.weight', 'ff.net.0.proj.weight',
                  'ff.net.2.weight', 'proj_out.weight', 'proj_in.weight')
TEXT_LORA_KEYS = ('q_proj.weight', 'k_proj.weight', 'v_proj.weight', 'out_proj.weight', 'fc1.weight', 'fc2.weight')


def revise_edlora_unet_attention_forward(unet):
    """edlora_util.py:85-99"""
    def visit(module, count):
        for name, layer in module.named_children():
            if layer.__class__.__name__ == 'Attention' and 'attn2' in name:
                layer.set_processor(EDLoRA_AttnProcessor(count))
                count += 1
            else:
                count = visit(layer, count)
        return count

    n = visit(unet.down_blocks, 0)
    n = visit(unet.mid_block, n)
    n = visit(unet.up_blocks, n)
    return n


def bind_concept_prompt(prompts, new_concept_cfg):
    """edlora_util.py:102-113: each prompt becomes 16 per-layer prompts with `<concept>` -> `<concept_k>` tokens."""
    if isinstance(prompts, str):
        prompts = [prompts]
    out = []
    for prompt in prompts:
        per_layer = [prompt] * 16
        for concept, cfg in new_concept_cfg.items():
            per_layer = [p.replace(concept, name) for p, name in zip(per_layer, cfg['concept_token_names'])]
        out.extend(per_layer)
    return out


def encode_edlora_prompt(pipe, prompt, new_concept_cfg, device, num_images_per_prompt=1,
                         do_classifier_free_guidance=False, negative_prompt=None, prompt_embeds=None,
                         negative_prompt_embeds=None):
    """edlora_util.py:116-196 -> [B (x2 with CFG: uncond first), 16, 77, D]"""
    assert num_images_per_prompt == 1, 'only support num_images_per_prompt=1 now'
    if prompt is not None and isinstance(prompt, list):
        batch_size = len(prompt)
    elif prompt is not None and prompt_embeds is None:
        batch_size = 1
    else:
        batch_size = prompt_embeds.shape[0]
    dtype = pipe.unet.dtype
    if prompt_embeds is None:
        ids = pipe.tokenizer(bind_concept_prompt(prompt, new_concept_cfg), padding='max_length',
                             max_length=pipe.tokenizer.model_max_length, truncation=True,
                             return_tensors='pt').input_ids
        emb = pipe.text_encoder(ids.to(device))[0]
        prompt_embeds = emb.reshape(batch_size, emb.shape[0] // batch_size, emb.shape[1], emb.shape[2])
    prompt_embeds = prompt_embeds.to(dtype=dtype, device=device)
    _, layer_num, seq_len, _ = prompt_embeds.shape
    if do_classifier_free_guidance:
        if negative_prompt_embeds is None:
            if negative_prompt is None:
                tokens = [''] * batch_size
            elif isinstance(negative_prompt, str):
                tokens = [negative_prompt]
            else:
                tokens = negative_prompt
            if len(tokens) != batch_size:
                raise ValueError(f'`negative_prompt` has batch size {len(tokens)}, but `prompt` has {batch_size}')
            ids = pipe.tokenizer(tokens, padding='max_length', max_length=seq_len, truncation=True,
                                 return_tensors='pt').input_ids
            negative_prompt_embeds = pipe.text_encoder(ids.to(device))[0]
        neg = negative_prompt_embeds.to(dtype=dtype, device=device)
        neg = neg.view(batch_size, 1, neg.shape[1], -1).repeat(1, layer_num, 1, 1)
        prompt_embeds = torch.cat([neg, prompt_embeds])
    return prompt_embeds


def load_new_concept(pipe, new_concept_embedding, enable_edlora=True):
    """convert_edlora_to_diffusers.py:4-33: add 16 (or 1) tokens per concept and write their embeddings."""
    new_concept_cfg = {}
    for concept_name, concept_embedding in new_concept_embedding.items():
        n = 16 if enable_edlora else 1
        names = [f'<{concept_name}_{layer_id}>' for layer_id in range(n)]
        added = pipe.tokenizer.add_tokens(names)
        if added != 0:
            assert added == len(names), 'some token is already in tokenizer'
        ids = [pipe.tokenizer.convert_tokens_to_ids(name) for name in names]
        if hasattr(pipe.text_encoder, 'resize_token_embeddings'):
            pipe.text_encoder.resize_token_embeddings(len(pipe.tokenizer))
            table = pipe.text_encoder.get_input_embeddings().weight.data
            table[ids] = concept_embedding.clone().to(table.device, dtype=table.dtype)
        new_concept_cfg[concept_name] = {'concept_token_ids': ids, 'concept_token_names': names}
    return pipe, new_concept_cfg


def merge_lora_into_weight(original_state_dict, lora_state_dict, model_type, alpha, touched=None):
    """convert_edlora_to_diffusers.py:36-79: W += alpha * (up @ down) for every key that has LoRA factors.  Same values as
    the reference's result; the reference deep-copies the whole state dict first (2.5 GB for the UNet) — here the entries
    without LoRA factors stay references to the caller's tensors (ALIASES of the live model's parameters when the caller
    passed `model.state_dict()`: a converter that edits the result in place must clone those entries first), the merged
    ones are new tensors.  `touched` (a list) receives the merged keys."""
    assert model_type in ('unet', 'text_encoder')
    keys = UNET_LORA_KEYS if model_type == 'unet' else TEXT_LORA_KEYS
    merged = dict(original_state_dict)
    count = 0
    for k in list(merged.keys()):
        down_name = k
        for suffix in keys:
            down_name = down_name.replace(suffix, suffix[:-len('weight')] + 'lora_down.weight')
        up_name = down_name.replace('lora_down', 'lora_up')
        if up_name not in lora_state_dict:
            continue
        count += 1
        w = merged[k]
        down = lora_state_dict[down_name].to(w.device)
        up = lora_state_dict[up_name].to(w.device)
        if w.dim() == 4:
            delta = (up.squeeze() @ down.squeeze()).unsqueeze(-1).unsqueeze(-1)
        else:
            delta = up @ down
        merged[k] = w + alpha * delta.to(w.dtype)
        if touched is not None:
            touched.append(k)
    print(f'load {count} LoRAs of {model_type}')
    return merged


def convert_edlora(pipe, state_dict, enable_edlora, alpha=0.6, snapshot=None):
    """convert_edlora_to_diffusers.py:82-105.  Only the weights that carry LoRA factors are written (the reference
    reloads the whole merged state dict: same end state).  `snapshot` = {'unet': {}, 'text_encoder': {}} receives a copy of
    every weight BEFORE it is modified (keys already present are kept: the first, pristine version), so that the caller
    can undo the merge with `load_state_dict(snapshot[...], strict=False)` instead of keeping a deep copy of everything
    (pipeline_videoswap.py:303-305,417-420 does the latter)."""
    state_dict = state_dict['params'] if 'params' in state_dict.keys() else state_dict
    new_concept_cfg = None
    if 'new_concept_embedding' in state_dict and len(state_dict['new_concept_embedding']) != 0:
        pipe, new_concept_cfg = load_new_concept(pipe, state_dict['new_concept_embedding'], enable_edlora)

    def merge_into(model, lora, model_type):
        current = model.state_dict()
        touched = []
        merged = merge_lora_into_weight(current, lora, model_type=model_type, alpha=alpha, touched=touched)
        if snapshot is not None:
            keep = snapshot.setdefault(model_type, {})
            for k in touched:
                if k not in keep:
                    keep[k] = current[k].detach().clone()
        model.load_state_dict({k: merged[k] for k in touched}, strict=False)
    if 'unet' in state_dict:
        merge_into(pipe.unet, state_dict['unet'], 'unet')
    if 'text_encoder' in state_dict and pipe.text_encoder is not None and hasattr(pipe.text_encoder, 'state_dict'):
        merge_into(pipe.text_encoder, state_dict['text_encoder'], 'text_encoder')
    return pipe, new_concept_cfg
