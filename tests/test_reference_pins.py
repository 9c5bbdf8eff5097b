"""Pins of the host logic that rides on the hot path against the REFERENCE's own files:

* a19 ED-LoRA: videoswap_amd/edlora.py against convert_edlora_to_diffusers.py / edlora_util.py — golden vectors those
  files produced (tests/golden/edlora.pt, generator tests/golden/make_golden_processors.py) and, when /root/reference is
  present, the reference functions imported verbatim, call by call, bit-exact on the CPU;
* a12 processors (CPU half): the goldens in tests/golden/processors.pt are re-derived from the reference when it is
  present, so the GPU test (tests/test_processors_gpu.py) that compares the HIP processors with them is anchored.
"""
import os
import sys

import pytest
import torch

from util import GOLDEN, load_golden

sys.path.insert(0, GOLDEN)
import toy  # noqa: E402

sys.dont_write_bytecode = True


def _same(a, b):
    if torch.is_tensor(a):
        return torch.equal(a, b)
    if isinstance(a, dict):
        return a.keys() == b.keys() and all(_same(a[k], b[k]) for k in a)
    if isinstance(a, (list, tuple)):
        return len(a) == len(b) and all(_same(x, y) for x, y in zip(a, b))
    return a == b


def _product_edlora_cases():
    import make_golden_processors as mg
    from videoswap_amd import edlora
    unet, text, ckpt = toy.lora_case()
    out = {'merged_unet': edlora.merge_lora_into_weight(unet, ckpt['params']['unet'], 'unet', 0.7),
           'merged_text': edlora.merge_lora_into_weight(text, ckpt['params']['text_encoder'], 'text_encoder', 1.0)}
    pipe = mg._Pipe(unet, text)
    _, cfg = edlora.load_new_concept(pipe, ckpt['params']['new_concept_embedding'], enable_edlora=True)
    out['new_concept_cfg'] = cfg
    out['token_table_tail'] = pipe.text_encoder.get_input_embeddings().weight.data[100:].clone()
    prompt = 'a <catA1> <catA2> sitting on a wooden floor'
    out['bound_prompts'] = edlora.bind_concept_prompt(prompt, cfg)
    out['prompt_embeds_cfg'] = edlora.encode_edlora_prompt(pipe, prompt, cfg, 'cpu', 1, True, 'low quality')
    out['prompt_embeds_nocfg'] = edlora.encode_edlora_prompt(pipe, [prompt], cfg, 'cpu', 1, False)
    return out


def test_edlora_host_logic_matches_reference_golden():
    gold = load_golden('edlora.pt')['cases']
    got = _product_edlora_cases()
    assert got.keys() == gold.keys()
    for k in gold:
        assert _same(got[k], gold[k]), k
    # a 1x1-conv weight keeps its 4-D shape after the merge; untouched keys are untouched
    unet, _, _ = toy.lora_case()
    assert got['merged_unet']['down_blocks.0.attentions.0.proj_in.weight'].shape == (32, 32, 1, 1)
    assert torch.equal(got['merged_unet']['down_blocks.0.resnets.0.conv1.weight'],
                       unet['down_blocks.0.resnets.0.conv1.weight'])
    assert not torch.equal(got['merged_unet']['down_blocks.0.motion_modules.0.temporal_transformer.proj_out.weight'],
                           unet['down_blocks.0.motion_modules.0.temporal_transformer.proj_out.weight'])


def test_convert_edlora_walks_the_checkpoint_layout():
    """convert_edlora on the on-disk layout {'params': {new_concept_embedding, unet, text_encoder}} (f3) mutates the
    UNet / text encoder exactly as merge_lora_into_weight says and returns the concept cfg."""
    import make_golden_processors as mg
    from videoswap_amd import edlora
    unet_sd, text_sd, ckpt = toy.lora_case()

    class Holder(torch.nn.Module):
        def __init__(self, sd):
            super().__init__()
            self.p = torch.nn.ParameterDict({k.replace('.', '__'): torch.nn.Parameter(v.clone()) for k, v in sd.items()})

        def state_dict(self, *a, **k):
            return {k.replace('__', '.'): v.detach() for k, v in self.p.items()}

        def load_state_dict(self, sd, strict=True):
            with torch.no_grad():
                for k, v in sd.items():
                    self.p[k.replace('.', '__')].copy_(v)

    pipe = mg._Pipe(unet_sd, text_sd)
    pipe.unet = Holder(unet_sd)
    pipe.unet.dtype = torch.float32
    enc = pipe.text_encoder
    holder = Holder(text_sd)
    enc.state_dict = holder.state_dict
    enc.load_state_dict = holder.load_state_dict
    _, cfg = edlora.convert_edlora(pipe, ckpt, enable_edlora=True, alpha=0.7)
    gold = load_golden('edlora.pt')['cases']
    assert cfg == gold['new_concept_cfg']
    assert _same(pipe.unet.state_dict(), gold['merged_unet'])


@pytest.mark.skipif(not os.path.isdir('/root/reference/videoswap'), reason='reference tree not present')
def test_goldens_are_what_the_reference_produces():
    """Re-derive both golden files from the reference imported verbatim (build container only)."""
    import make_golden_processors as mg
    p2p, reg, edl, conv = mg.reference_modules()
    assert _same(mg.edlora_cases(edl, conv), load_golden('edlora.pt')['cases'])
    fresh, gold = mg.processor_cases(reg, edl), load_golden('processors.pt')['cases']
    assert fresh.keys() == gold.keys()
    for k in gold:
        if torch.is_tensor(gold[k]):
            assert torch.allclose(fresh[k], gold[k], rtol=0, atol=1e-6), k
        else:
            assert fresh[k] == gold[k], k


def test_oracle_processors_match_reference_golden():
    """The oracle's restated processors (oracle/pipeline.py ControlProcessor / EDLoRAProcessor), which the end-to-end
    swap tests use, against what the reference's processors produced."""
    import make_golden_processors as mg
    from oracle import pipeline as op
    gold = load_golden('processors.pt')['cases']
    inp = toy.attention_inputs()
    self_sd, cross_sd = toy.attention_weights()
    a_self, a_cross = mg.oracle_attention(self_sd, False), mg.oracle_attention(cross_sd, True)
    c = toy.ToyController()
    with torch.no_grad():
        got = {
            'edlora_cross_layers_idx3': op.EDLoRAProcessor(3)(a_cross, inp['hidden'], inp['text_layers']),
            'edlora_cross_single': op.EDLoRAProcessor(3)(a_cross, inp['hidden'], inp['text']),
            'edlora_self': op.EDLoRAProcessor(0)(a_self, inp['hidden'], None),
            'control_self_down': op.ControlProcessor('down', c)(a_self, inp['hidden'], None),
            'control_cross_mid': op.ControlProcessor('mid', c)(a_cross, inp['hidden'], inp['text']),
            'edlora_control_cross_up_idx5': op.ControlProcessor('up', c, 5)(a_cross, inp['hidden'], inp['text_layers']),
            'edlora_control_self_up': op.ControlProcessor('up', c, 5)(a_self, inp['hidden'], None),
        }
    for k, v in got.items():
        assert torch.allclose(v, gold[k], rtol=0, atol=2e-5), k
    assert c.calls == gold['controller_calls']
