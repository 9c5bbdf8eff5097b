"""a12 on the GPU: the product's attention processors (videoswap_amd/attention.py) on the HIP-backed `Attention`
against what the REFERENCE's processors (attention_register.py:15-173, edlora_util.py:13-82, imported verbatim by
tests/golden/make_golden_processors.py) produced on the fp32 diffusers `Attention` for the same weights, inputs and
controller.  fp16 storage: rel-L2 <= 3e-3."""
import sys

import pytest
import torch

from util import DEV, GOLDEN, load_golden, rel_l2, sync

sys.path.insert(0, GOLDEN)
import toy  # noqa: E402

pytestmark = pytest.mark.device


def product_attention(sd, cross):
    from videoswap_amd.attention import Attention
    a = Attention(query_dim=320, cross_attention_dim=768 if cross else None, heads=8, dim_head=40).eval()
    a.load_state_dict(sd, strict=True)
    return a.to(DEV, torch.float16)


def test_product_processors_match_the_reference_processors():
    from videoswap_amd.attention import AttnControlProcessor, EDLoRA_AttnControlProcessor, EDLoRA_AttnProcessor
    gold = load_golden('processors.pt')['cases']
    inp = {k: v.half().to(DEV) for k, v in toy.attention_inputs().items()}
    self_sd, cross_sd = toy.attention_weights()
    a_self, a_cross = product_attention(self_sd, False), product_attention(cross_sd, True)
    c = toy.ToyController()
    with torch.no_grad():
        got = {
            'edlora_cross_layers_idx3': EDLoRA_AttnProcessor(3)(a_cross, inp['hidden'], inp['text_layers']),
            'edlora_cross_single': EDLoRA_AttnProcessor(3)(a_cross, inp['hidden'], inp['text']),
            'edlora_self': EDLoRA_AttnProcessor(0)(a_self, inp['hidden'], None),
            'control_self_down': AttnControlProcessor('down', c)(a_self, inp['hidden'], None),
            'control_cross_mid': AttnControlProcessor('mid', c)(a_cross, inp['hidden'], inp['text']),
            'edlora_control_cross_up_idx5': EDLoRA_AttnControlProcessor(5, 'up', c)(a_cross, inp['hidden'],
                                                                                     inp['text_layers']),
            'edlora_control_self_up': EDLoRA_AttnControlProcessor(5, 'up', c)(a_self, inp['hidden'], None),
        }
    sync()
    for k, v in got.items():
        e = rel_l2(v.float().cpu(), gold[k])
        print(f'{k}: rel-L2 {e:.2e}')
        assert e < 3e-3, k
    # same controller traffic as the reference: (is_cross, place, probs shape [b, heads, s, t])
    assert c.calls == gold['controller_calls']


def test_text_shared_by_frames_equals_reference_repeat():
    """The product may receive the text embedding un-repeated ([B, 77, D] with video_length frames per clip); the
    reference repeats it over frames (attention.py:100-103).  Both must give the same result."""
    from videoswap_amd.attention import EDLoRA_AttnProcessor
    gold = load_golden('processors.pt')['cases']
    inp = toy.attention_inputs()
    _, cross_sd = toy.attention_weights()
    a = product_attention(cross_sd, True)
    hidden = inp['hidden'].half().to(DEV)
    text = inp['text_layers'][:1].half().to(DEV)            # one clip of 2 frames
    with torch.no_grad():
        shared = EDLoRA_AttnProcessor(3)(a, hidden, text, video_length=2)
        repeated = EDLoRA_AttnProcessor(3)(a, hidden, text.repeat(2, 1, 1, 1))
    assert rel_l2(shared.float(), repeated.float()) < 1e-3
