"""Parity of the HIP-backed AnimateDiffUNet3DModel (GPU box: the kernels; here: the host mirror on tests/host_emulation.py) against the golden vectors produced by the reference's own
model code (tests/golden/unet_tiny.pt) and against the fp32 CPU oracle.

Tolerance: the fp16 GPU result must be within 2x the rel-L2 error that the oracle itself shows
when run with fp16 storage (measured in this test on the CPU), both taken against the fp32 result."""
import copy

import pytest
import torch

from util import cosine, DEV, load_golden, oracle_unet, product_unet_from, rel_l2, sync

pytestmark = pytest.mark.device


@pytest.fixture(scope='module')
def setup():
    blob = load_golden('unet_tiny.pt')
    ora = oracle_unet(blob['config'], blob['weight_seed'])
    prod = product_unet_from(ora, blob['config'])
    return blob, ora, prod


def run_product(prod, case):
    res = None
    if case['residuals'] is not None:
        res = [r.half().to(DEV) for r in case['residuals']]
    with torch.no_grad():
        out = prod(case['sample'].half().to(DEV), torch.tensor(case['timestep']), case['text'].half().to(DEV),
                   down_block_additional_residuals=res, return_dict=False)[0]
    sync()
    return out.float().cpu()


@pytest.mark.parametrize('name', ['plain_T4_16x16', 'cfg_adapter_T3_16x24', 'first_inverse_step_t-19'])
def test_unet_matches_reference_golden(setup, name):
    blob, ora, prod = setup
    case = blob['cases'][name]
    half = copy.deepcopy(ora).half()
    res = None if case['residuals'] is None else [r.half() for r in case['residuals']]
    with torch.no_grad():
        e16 = rel_l2(half(case['sample'].half(), torch.tensor(case['timestep']), case['text'].half(),
                          down_block_additional_residuals=res).sample.float(), case['out'])
    out = run_product(prod, case)
    assert out.shape == case['out'].shape
    err = rel_l2(out, case['out'])
    print(f'{name}: gpu rel-L2 {err:.3e} (fp16-oracle {e16:.3e}) cos {cosine(out, case["out"]):.8f}')
    assert torch.isfinite(out).all()
    assert err <= 2 * e16, f'{name}: rel-L2 {err:.3e} > 2 x fp16-emulation error {e16:.3e}'


def test_shared_cfg_prefix_equals_the_duplicated_batch(setup):
    """Classifier-free guidance feeds the UNet the same latents twice (pipeline_videoswap.py:556).  Given as a stride-0 batch
    view, the product computes conv_in, the first resnet and the first self-attention once; the result must match the
    materialised `torch.cat([latents] * 2)` batch (same arithmetic per element; GroupNorm partial sums are grouped by the
    image count, hence the small tolerance), and a controller hooked on that self-attention switches the sharing off."""
    blob, ora, prod = setup
    case = blob['cases']['cfg_adapter_T3_16x24']
    x, txt = case['sample'].half().to(DEV), case['text'].half().to(DEV)
    assert x.shape[0] == 2 and txt.shape[0] == 2
    one = x[:1].contiguous()
    dup = one.expand(2, *one.shape[1:])
    cat = torch.cat([one, one])
    res = [r.half().to(DEV) for r in case['residuals']]
    one_row = torch.zeros(1, 8)          # stands for the [1, C] time-embedding row of a scalar timestep
    assert prod._shared_cfg_prefix(dup, txt, one_row) and not prod._shared_cfg_prefix(cat, txt, one_row)
    # two timestep values = two time-embedding rows: the stride-0 sample proves equal latents, not equal timesteps
    assert not prod._shared_cfg_prefix(dup, txt, torch.zeros(2, 8))
    with torch.no_grad():
        a = prod(dup, 481, txt, down_block_additional_residuals=list(res)).sample.float().cpu()
        b = prod(cat, 481, txt, down_block_additional_residuals=list(res)).sample.float().cpu()
    sync()
    assert a.shape == b.shape and torch.isfinite(a).all()
    assert rel_l2(a, b) < 1e-3, rel_l2(a, b)
    assert not torch.equal(a[0], a[1]), 'the halves must differ behind the cross-attention (different text rows)'
    # a processor that is not this package's fused one on that self-attention (a Prompt-to-Prompt hook, a foreign
    # processor) expects both halves: the sharing must step aside
    from videoswap_amd.attention import AttnProcessor, AttnProcessor2_0
    attn1 = prod.down_blocks[0].attentions[0].transformer_blocks[0].attn1
    attn1.set_processor(type('Foreign', (AttnProcessor,), {'vsx_native': False, 'vsx_shareable': False})())
    try:
        assert not prod._shared_cfg_prefix(dup, txt, one_row)
        with torch.no_grad():
            c = prod(dup, 481, txt, down_block_additional_residuals=list(res)).sample.float().cpu()
        assert rel_l2(c, b) < 5e-3         # (another arithmetic path: the LayerNorm in front of a foreign processor is not folded)
    finally:
        attn1.set_processor(AttnProcessor2_0())


def test_shared_cfg_prefix_of_several_clips(setup):
    """Several clips denoised together (bench.py's throughput leg: four per step): `VideoSwapPipeline.__call__` passes
    `cfg_halves_equal=True` with its own `torch.cat([latents] * 2)` (an explicit keyword, ADVICE r5: an attribute on the tensor
    is dropped by any op in between); the UNet then runs conv_in, the first resnet and the first self-attention on ONE half.
    Must match the unstated batch (same arithmetic per element), keep the clips apart, and a batch whose halves merely happen
    to be equal must not be shared unless the caller says so."""
    blob, ora, prod = setup
    case = blob['cases']['cfg_adapter_T3_16x24']
    x, txt = case['sample'].half().to(DEV), case['text'].half().to(DEV)
    clips = torch.cat([x[:1], x[1:2] * 0.5 + 0.1])                    # two different clips
    text4 = torch.cat([txt[:1], txt[:1] * 0.7, txt[1:2], txt[1:2] * 0.7])     # [uncond clip 0, 1 ; cond clip 0, 1]
    plain = torch.cat([clips, clips])
    one_row = torch.zeros(1, 8)
    assert prod._shared_cfg_prefix(plain, text4, one_row, True) == 2 and prod._shared_cfg_prefix(plain, text4, one_row) == 0
    assert prod._shared_cfg_prefix(plain, text4[:2], one_row, True) == 0       # text rows must cover the whole batch
    assert prod.vsx_cfg_keyword        # what VideoSwapPipeline asks before it passes the keyword to a UNet object
    with torch.no_grad():
        a = prod(plain.clone(), 481, text4, cfg_halves_equal=True).sample.float().cpu()      # (survives a clone: no tensor attribute)
        b = prod(plain, 481, text4).sample.float().cpu()
    sync()
    assert a.shape == b.shape and torch.isfinite(a).all()
    assert rel_l2(a, b) < 1e-3, rel_l2(a, b)
    assert not torch.equal(a[0], a[1]) and not torch.equal(a[0], a[2])


def test_shared_cfg_prefix_steps_aside_for_a_registered_controller_and_for_two_timesteps(setup):
    """The Prompt-to-Prompt processors are `vsx_native` (they run on the kernels) but NOT shareable: below 32 x 32 latents the
    controller is called on the first self-attention and splits its argument into the two CFG halves (attention_util.py:
    `attn.shape[0] // 2`).  With a REAL `register_attention_control` at a 16 x 24 latent the stride-0 batch must give exactly
    what the materialised batch gives (same kernels, same shapes), the store must have seen 2 F images on down-block 0, and
    a [2] timestep tensor with two different values must not take item 0's embedding for both."""
    from videoswap_amd import control
    from videoswap_amd.attention import AttnProcessor2_0
    blob, ora, prod = setup
    case = blob['cases']['cfg_adapter_T3_16x24']
    x, txt = case['sample'].half().to(DEV), case['text'].half().to(DEV)
    one = x[:1].contiguous()
    dup, cat = one.expand(2, *one.shape[1:]), torch.cat([one, one])
    frames = x.shape[2]
    pipe = type('P', (), {'unet': prod})()
    seen = []

    class Probe(control.AttentionStore):
        def __call__(self, attn, is_cross, place_in_unet):
            seen.append((place_in_unet, is_cross, attn.shape[0]))
            return super().__call__(attn, is_cross, place_in_unet)
    try:
        with torch.no_grad():
            control.register_attention_control(pipe, Probe())
            assert not prod._shared_cfg_prefix(dup, txt, torch.zeros(1, 8))
            a = prod(dup, 481, txt).sample
            heads = prod.down_blocks[0].attentions[0].transformer_blocks[0].attn1.heads
            first = seen[0]
            assert first[0] == 'down' and first[1] is False
            assert first[2] in (2 * frames, 2 * frames * heads), f'the controller must see both CFG halves, got {first}'
            control.register_attention_control(pipe, Probe())
            b = prod(cat, 481, txt).sample
            assert torch.equal(a, b)
    finally:
        for name, m in prod.named_modules():
            if m.__class__.__name__ == 'Attention' and ('attn1' in name or 'attn2' in name):
                m.set_processor(AttnProcessor2_0())
    # two different timesteps on a stride-0 sample: every batch item gets its own embedding row
    with torch.no_grad():
        t2 = torch.tensor([481, 21])
        both = prod(dup, t2, txt).sample
        item1 = prod(cat, torch.tensor([21, 21]), txt).sample
        item0 = prod(cat, torch.tensor([481, 481]), txt).sample
    assert rel_l2(both[1].float(), item1[1].float()) < 2e-3 and rel_l2(both[0].float(), item0[0].float()) < 2e-3
    assert rel_l2(both[1].float(), item0[1].float()) > 1e-2, 'item 1 must not run with item 0\'s timestep'


def test_unet_output_object_and_determinism(setup):
    blob, ora, prod = setup
    case = blob['cases']['plain_T4_16x16']
    x, txt = case['sample'].half().to(DEV), case['text'].half().to(DEV)
    with torch.no_grad():
        a = prod(x, 481, txt)
        b = prod(x, torch.tensor([481]), txt, return_dict=False)
    assert hasattr(a, 'sample') and isinstance(b, tuple)
    assert torch.equal(a.sample, b[0]), 'the forward must be bit-deterministic (no atomics anywhere)'
    assert a.sample.shape == x.shape and a.sample.dtype == torch.float16


def test_unet_pops_adapter_residual_list(setup):
    blob, ora, prod = setup
    case = blob['cases']['cfg_adapter_T3_16x24']
    res = [r.half().to(DEV) for r in case['residuals']]
    with torch.no_grad():
        prod(case['sample'].half().to(DEV), 21, case['text'].half().to(DEV), down_block_additional_residuals=res)
    assert res == []          # the UNet pops from the caller's list (unet.py:422,435)


def test_zero_initialised_motion_module_is_identity(setup):
    """With AnimateDiff's zero-init proj_out the temporal path must contribute exactly nothing."""
    blob, ora, prod = setup
    from videoswap_amd.unet import VanillaTemporalModule, Geometry
    mm = VanillaTemporalModule(64, temporal_position_encoding=True, num_transformer_block=1).half().to(DEV)
    x = torch.randn(8, 4, 4, 64, device=DEV, dtype=torch.float16)
    with torch.no_grad():
        y = mm(x, Geometry(2, 4))
    assert torch.equal(x, y)


def test_step_invariant_caches_follow_inputs_and_weights(setup):
    """The text K/V and time-embedding caches must be invisible: same results with cold and warm caches, and an
    in-place edit of the text embedding or a weight reload is picked up."""
    blob, ora, prod = setup
    case = blob['cases']['plain_T4_16x16']
    x, txt = case['sample'].half().to(DEV), case['text'].half().to(DEV)
    with torch.no_grad():
        prod.clear_step_caches()
        cold = prod(x, 481, txt).sample
        warm = prod(x, 481, txt).sample                      # every cache hits
        assert torch.equal(cold, warm)
        txt2 = txt.clone()
        ref2 = prod(x, 481, txt2 * 0.5).sample               # what an uncached forward gives for the scaled text
        txt2.mul_(0.5)                                        # same tensor object, new contents
        a = prod(x, 481, txt2).sample
        b = prod(x, 481, txt2).sample
        assert torch.equal(a, ref2) and torch.equal(a, b)
        assert not torch.equal(a, cold)
        # weights: scale one text-K projection and one time-embedding projection in place (load_state_dict style)
        sd = {k: v.clone() for k, v in prod.state_dict().items()}
        kname = next(k for k in sd if k.endswith('attn2.to_k.weight'))
        tname = next(k for k in sd if k.endswith('time_emb_proj.weight'))
        sd[kname] *= 1.5
        sd[tname] *= 1.5
        orig = {k: v.clone() for k, v in prod.state_dict().items()}
        prod.load_state_dict(sd)
        c = prod(x, 481, txt).sample
        prod.clear_step_caches()
        d = prod(x, 481, txt).sample
        prod.load_state_dict(orig)
        e = prod(x, 481, txt).sample
    assert torch.equal(c, d), 'a weight reload must invalidate the cached projections'
    assert not torch.equal(c, cold)
    assert torch.equal(e, cold)


@pytest.mark.gpu
def test_hip_graph_replay_is_bit_identical_to_eager(setup):
    """The captured forward replays the same kernels on the same data layout: bit-identical outputs, across
    timesteps, texts, adapter residuals and a weight reload (which must drop the graph)."""
    blob, ora, prod = setup
    case = blob['cases']['cfg_adapter_T3_16x24']
    x, txt = case['sample'].half().to(DEV), case['text'].half().to(DEV)
    res = [r.half().to(DEV) for r in case['residuals']]
    with torch.no_grad():
        eager = [prod(x, t, txt).sample for t in (21, 481)]
        eager_res = prod(x, 21, txt, down_block_additional_residuals=list(res)).sample
        prod.enable_hip_graphs(True)
        try:
            for _ in range(2):                       # second round: pure replays
                for t, ref in zip((21, 481), eager):
                    assert torch.equal(prod(x, t, txt).sample, ref)
            lst = list(res)
            assert torch.equal(prod(x, 21, txt, down_block_additional_residuals=lst).sample, eager_res)
            assert lst == []
            assert prod._graphs.captures == 2 and prod._graphs.replays >= 5
            other = (txt * 0.5).contiguous()
            ref_other = prod(x, 21, other).sample     # same shapes: same graph, new text copied in
            prod.enable_hip_graphs(False)
            assert torch.equal(prod(x, 21, other).sample, ref_other)
            prod.enable_hip_graphs(True)
            sd = {k: v.clone() for k, v in prod.state_dict().items()}
            name = next(k for k in sd if k.endswith('attn1.to_q.weight'))
            sd2 = dict(sd)
            sd2[name] = sd[name] * 1.25
            prod(x, 21, txt)
            prod.load_state_dict(sd2)
            changed = prod(x, 21, txt).sample
            assert not torch.equal(changed, eager[0])
            prod.load_state_dict(sd)
            assert torch.equal(prod(x, 21, txt).sample, eager[0])
        finally:
            prod.enable_hip_graphs(False)


@pytest.mark.gpu
def test_hip_graph_replay_keeps_the_shared_cfg_prefix_and_keys_on_it(setup):
    """the graph cache cloned the sample into its static buffer, which drops a stride-0 batch view — the shared CFG
    prefix was silently never taken under replay — and its key ignored the proof.  `forward` now decides `half` on the caller's
    tensor and hands it to the cache: (i) a stride-0 CFG batch replays the SHARED body (bit-identical to the eager shared
    forward), (ii) the materialised batch of the same shape gets its OWN graph (two captures), bit-identical to its eager
    forward — a graph captured with sharing is never replayed for a batch whose halves may differ."""
    blob, ora, prod = setup
    case = blob['cases']['cfg_adapter_T3_16x24']
    x, txt = case['sample'].half().to(DEV), case['text'].half().to(DEV)
    one = x[:1].contiguous()
    dup = one.expand(2, *one.shape[1:])
    differ = torch.cat([one, one * 0.5 + 0.25])             # same shape, halves NOT equal
    with torch.no_grad():
        eager_shared = prod(dup, 481, txt).sample
        eager_differ = prod(differ, 481, txt).sample
        assert not torch.equal(eager_shared[1], eager_differ[1])
        prod.enable_hip_graphs(True)
        try:
            for _ in range(2):
                assert torch.equal(prod(dup, 481, txt).sample, eager_shared)
                assert torch.equal(prod(differ, 481, txt).sample, eager_differ)
            assert prod._graphs.captures == 2 and prod._graphs.replays == 4
            halves = sorted(k[3] for k in prod._graphs.entries)
            assert halves == [0, 1], halves
        finally:
            prod.enable_hip_graphs(False)


@pytest.mark.gpu
def test_hip_graphs_step_aside_for_controllers(setup):
    """Prompt-to-Prompt control processors keep host state per call: the graph path must not be taken."""
    from videoswap_amd import control
    blob, ora, prod = setup
    case = blob['cases']['plain_T4_16x16']
    x, txt = case['sample'].half().to(DEV), case['text'].half().to(DEV)
    pipe = type('P', (), {'unet': prod})()
    with torch.no_grad():
        prod.enable_hip_graphs(True)
        try:
            store = control.AttentionStore()
            control.register_attention_control(pipe, store)
            prod(x, 481, txt)
            assert prod._graphs.captures == 0 and prod._graphs.replays == 0
            control.register_attention_control(pipe, control.EmptyControl())
        finally:
            prod.enable_hip_graphs(False)
            from videoswap_amd.attention import AttnProcessor2_0
            for name, m in prod.named_modules():
                if m.__class__.__name__ == 'Attention' and ('attn1' in name or 'attn2' in name):
                    m.set_processor(AttnProcessor2_0())
