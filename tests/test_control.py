"""CPU tests of the Prompt-to-Prompt host logic (videoswap_amd/control.py).

The product controllers are plain device-agnostic torch code, so they are driven here on the CPU through the same
sequence of calls the pipeline makes (inversion store -> edit with AttentionRefine + two SpatialBlenders) and
compared with (a) the reference's own p2p modules imported verbatim, when /root/reference is present, and
(b) the golden trace those produced (tests/golden/p2p_trace.pt)."""
import os

import pytest
import torch

from util import GOLDEN, load_golden

PROMPTS = ['a silver jeep driving down a curvy road in the countryside',
           'a yellow Porsche car driving down a curvy road in the countryside']
STEPS, FRAMES, HEADS = 6, 2, 2
# (place, is_cross, tokens): call order of one tiny "UNet": N >= 1024 layers skip the controller in the product
LAYERS = [('down', False, 16), ('down', True, 16), ('down', False, 16), ('down', True, 16),
          ('mid', False, 4), ('mid', True, 4),
          ('up', False, 16), ('up', True, 16), ('up', False, 16), ('up', True, 16), ('up', False, 16), ('up', True, 16)]
H, W = 4, 4      # "latent" resolution; image = 32 x 32


def fake_probs(step, layer, batch, is_cross, tokens):
    g = torch.Generator().manual_seed(1000 * step + 10 * layer + batch)
    t = 77 if is_cross else tokens
    return torch.softmax(torch.randn(batch * FRAMES, HEADS, tokens, t, generator=g) * 2, -1)


def drive(make_store, make_edit, latents0):
    """inversion with a store controller, then an edit pass; returns the trace of everything observable."""
    trace = []
    store = make_store()
    store.LOW_RESOURCE = True
    store.num_att_layers = len(LAYERS)
    x = latents0.clone()
    for s in range(STEPS):
        for li, (place, is_cross, n) in enumerate(LAYERS):
            store(fake_probs(s, li, 1, is_cross, n), is_cross, place)
        x = store.step_callback(x * 1.01 + 0.01 * s)
    store.LOW_RESOURCE = False
    edit = make_edit(store)
    edit.num_att_layers = len(LAYERS)
    y = x.clone()
    for s in range(STEPS):
        for li, (place, is_cross, n) in enumerate(LAYERS):
            out = edit(fake_probs(100 + s, li, 2, is_cross, n), is_cross, place)
            trace.append(out.clone())
        y = edit.step_callback(y * 0.99 - 0.02)
        trace.append(y.clone())
    return trace


def product_trace():
    from videoswap_amd import control
    from videoswap_amd.synthetic import WhitespaceTokenizer
    tok = WhitespaceTokenizer()

    def make_edit(store):
        return control.make_controller(tok, PROMPTS, False, cross_replace_steps=0.5, self_replace_steps=0.5,
                                       blend_words=[['silver', 'jeep'], ['yellow', 'Porsche', 'car']],
                                       additional_attention_store=store, blend_th=(0.3, 0.3), NUM_DDIM_STEPS=STEPS,
                                       blend_latents=True, blend_self_attention=True, image_height=H * 8,
                                       image_width=W * 8)
    g = torch.Generator().manual_seed(5)
    return drive(control.AttentionStore, make_edit, torch.randn(1, 4, FRAMES, H, W, generator=g))


def test_refinement_mapper_known_answer():
    """attention_util.py:268-280 quotes the mapper/alphas for this prompt pair."""
    from videoswap_amd.control import get_refinement_mapper
    from videoswap_amd.synthetic import WhitespaceTokenizer
    mapper, alphas = get_refinement_mapper(PROMPTS, WhitespaceTokenizer())
    assert mapper[0, :18].tolist() == [0, 1, -1, -1, -1, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 15, 16, 17]
    assert alphas[0, :6].tolist() == [1, 1, 0, 0, 0, 1] and alphas[0, 5:].min() == 1
    assert mapper[0, -1] == 76


def test_controllers_match_golden_trace():
    path = os.path.join(GOLDEN, 'p2p_trace.pt')
    if not os.path.exists(path):
        pytest.skip('golden trace not generated yet')
    gold = load_golden('p2p_trace.pt')
    trace = product_trace()
    assert len(trace) == len(gold)
    for i, (a, b) in enumerate(zip(trace, gold)):
        assert a.shape == b.shape, i
        assert torch.allclose(a.float(), b.float(), atol=1e-6, rtol=1e-5), f'trace item {i}'


def reference_trace():
    from oracle import ref_import
    mods = ref_import.load_reference_p2p()
    from videoswap_amd.synthetic import WhitespaceTokenizer
    tok = WhitespaceTokenizer()

    def make_edit(store):
        return mods['attention_util'].make_controller(
            tok, PROMPTS, False, cross_replace_steps=0.5, self_replace_steps=0.5,
            blend_words=[['silver', 'jeep'], ['yellow', 'Porsche', 'car']], additional_attention_store=store,
            blend_th=(0.3, 0.3), NUM_DDIM_STEPS=STEPS, blend_latents=True, blend_self_attention=True,
            image_height=H * 8, image_width=W * 8)
    g = torch.Generator().manual_seed(5)
    return drive(mods['attention_store'].AttentionStore, make_edit, torch.randn(1, 4, FRAMES, H, W, generator=g))


def test_controllers_match_reference_verbatim():
    from oracle import ref_import
    if not ref_import.available():
        pytest.skip('/root/reference is not present on this machine')
    ref, prod = reference_trace(), product_trace()
    assert len(ref) == len(prod)
    changed = 0
    for i, (a, b) in enumerate(zip(prod, ref)):
        assert torch.allclose(a.float(), b.float(), atol=1e-6, rtol=1e-5), f'trace item {i}'
    # the edit must actually have done something (cross maps replaced, latents blended)
    base = fake_probs(100, 1, 2, True, 16)
    assert not torch.allclose(prod[1], base)


if __name__ == '__main__':   # regenerate the golden trace from the reference (build container only)
    import sys
    sys.dont_write_bytecode = True
    torch.save([t.clone() for t in reference_trace()], os.path.join(GOLDEN, 'p2p_trace.pt'))
    print('wrote p2p_trace.pt')
