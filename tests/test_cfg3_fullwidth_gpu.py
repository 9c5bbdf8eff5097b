"""BASELINE.json configs[2] — the full swap path — at the SD-1.5 width on the GPU, against a golden recorded in the
build container by the fp32 oracle UNet driving the REFERENCE's own Prompt-to-Prompt controllers (attention_store.py,
attention_util.py, spatial_blend.py, seq_aligner.py imported verbatim: tests/golden/make_golden_cfg3.py).

Product side: `VideoSwapPipeline.validation` (pipeline_videoswap.py:272-423) on the HIP kernels — inversion with the
AttentionStore, ED-LoRA merge + per-layer embeddings [2,16,77,768], adapter residuals inside the t2i window,
AttentionRefine + latent / self-attention SpatialBlenders with `use_blend: true`, weights restored.  The store keeps its
maps on the device (videoswap_amd/control.py); the hook path materialises the 16x16 / 8x8 probabilities through
vsx_gemm_f16 + vsx_softmax_rows.

Tolerance: rel-L2(product, golden) <= 2 x rel-L2(fp16-storage oracle, golden), the yardstick being the same
oracle flow run here in fp16 on the device.  The blend mask is a threshold: both fp16 runs may flip mask pixels the fp32
golden does not, which is why the yardstick goes through the same flow.  A third leg — the fp32 oracle on the device with
the PRODUCT's controller classes — must reproduce the golden (reference controllers, CPU) almost exactly: it pins
videoswap_amd.control against the reference at the real 16x16x256-key shapes."""
import copy
import functools
import json
import os
import time

import pytest
import torch

import cfg3_case as case
from util import GOLDEN, ROOT, cosine, load_golden, rel_l2

pytestmark = pytest.mark.gpu
RESULTS = {}


def _record(name, **kw):
    out = os.path.join(ROOT, 'gpurun_out')
    try:
        os.makedirs(out, exist_ok=True)
        RESULTS[name] = kw
        with open(os.path.join(out, 'parity_cfg3.json'), 'w') as f:
            json.dump(RESULTS, f, indent=1, sort_keys=True)
    except OSError:
        pass
    print(name, json.dumps(kw))


@pytest.mark.parametrize('size', [case.SMALL, case.BENCH], ids=lambda s: s.name)
def test_cfg3_full_width_against_reference_controllers(size):
    """SMALL: T = 4, 4 + 4 steps (round 3).  BENCH (round 4, VERDICT weak 2): T = 16 — the frame count bench.py --config 3
    times — with 10 + 10 steps and `cross_replace_steps` / `self_replace_steps` 0.3, so that the step-dependent controller
    branches (attention_util.py:28-138: replacement on steps 0-2, plain store afterwards), the adapter window [0, 0.5]
    (five steps with, five without residuals) and both SpatialBlenders (spatial_blend.py:25-145) each run for several
    steps at the benchmarked width.  Both goldens come from the reference's own controllers (make_golden_cfg3.py)."""
    path = os.path.join(GOLDEN, size.golden)
    if not os.path.exists(path):
        pytest.fail(f'tests/golden/{size.golden} is missing (tests/golden/make_golden_cfg3.py writes it)')
    gold = load_golden(size.golden)
    assert gold['frames'] == size.frames and gold['steps'] == size.steps
    from oracle import adapter as oadapter
    from oracle import unet3d
    from videoswap_amd import control
    from videoswap_amd.adapter import SparsePointAdapter
    from videoswap_amd.compat import SD15_SCHEDULER_CONFIG, DDIMScheduler
    from videoswap_amd.pipeline import VideoSwapPipeline
    from videoswap_amd.synthetic import SyntheticTextEncoder, WhitespaceTokenizer, portable_weights_
    from videoswap_amd.unet import AnimateDiffUNet3DModel

    cfg = unet3d.full_config()
    chans = list(cfg['block_out_channels'])
    with torch.device('cuda'):
        prod = AnimateDiffUNet3DModel(**cfg)
        ora_dev = unet3d.AnimateDiffUNet3DModel(**cfg).eval()
    prod = portable_weights_(prod, seed=case.SEED_W).half().eval()
    ora_dev.load_state_dict({k: v.float() for k, v in prod.state_dict().items()}, strict=True)
    w = sum(float(p.detach().double().abs().sum()) for p in ora_dev.parameters())
    assert abs(w - gold['weights_abs_sum']) <= 1e-9 * gold['weights_abs_sum'], 'the portable weights differ from the golden run'
    oad = oadapter.SparsePointAdapter(1280, chans).eval()
    portable_weights_(oad, seed=case.SEED_A)
    for p in oad.parameters():
        p.data = p.data.half().float()
    pad = SparsePointAdapter(embedding_channels=1280, channels=chans).eval()
    pad.load_state_dict(oad.state_dict(), strict=True)
    pad = pad.to('cuda', torch.float16)

    # ---------------- product ----------------
    latents, conditions = case.inputs(size)
    tok = WhitespaceTokenizer()
    pipe = VideoSwapPipeline(unet=prod, adapter=pad, tokenizer=tok, scheduler=DDIMScheduler(**SD15_SCHEDULER_CONFIG),
                             text_encoder=SyntheticTextEncoder(dim=768, dtype=torch.float16, device='cuda')).to('cuda')
    before = copy.deepcopy(prod.state_dict())
    captured = {}
    invert = pipe.invert

    def capture(*a, **k):
        r = invert(*a, **k)
        captured['inverted'] = r.latents.float().cpu()
        return r
    pipe.invert = capture
    video = latents[0].permute(1, 0, 2, 3).contiguous().half().cuda()               # [F,4,h,w] "video" of latents
    t0 = time.time()
    edited = pipe.validation(video, conditions, case.SOURCE, case.editing_config(size=size),
                             lora_loader=lambda p: case.synthetic_lora(before))
    torch.cuda.synchronize()
    t_prod = time.time() - t0
    got = edited['0'].float().cpu()
    after = prod.state_dict()
    assert all(torch.equal(before[k], after[k]) for k in before), 'weights not restored after the LoRA merge'

    # ---------------- yardstick and controller pin on the device oracle ----------------
    mk = functools.partial(control.make_controller, device='cuda')
    inv32, out32 = case.oracle_flow(ora_dev, oad, control.AttentionStore, mk, size=size)
    ora_h = ora_dev.half()
    inv16, out16 = case.oracle_flow(ora_h, oad, control.AttentionStore, mk, size=size)

    e_pin_inv, e_pin = rel_l2(inv32, gold['inverted']), rel_l2(out32, gold['final'])
    e_inv, e16_inv = rel_l2(captured['inverted'], gold['inverted']), rel_l2(inv16, gold['inverted'])
    e_out, e16_out = rel_l2(got, gold['final']), rel_l2(out16, gold['final'])
    _record(f'cfg3_T{size.frames}_64x64_{size.steps}+{size.steps}_steps', inversion_rel_l2=e_inv, inversion_rel_l2_fp16_oracle=e16_inv, final_rel_l2=e_out,
            final_rel_l2_fp16_oracle=e16_out, final_cosine=cosine(got, gold['final']),
            final_cosine_fp16_oracle=cosine(out16, gold['final']),
            device_fp32_oracle_with_product_controllers_vs_golden=dict(inversion=e_pin_inv, final=e_pin),
            product_vs_fp16_oracle=rel_l2(got, out16), wall_s_product_validation=t_prod)
    assert torch.isfinite(got).all() and got.shape == gold['final'].shape
    # product controllers (device fp32 oracle) == reference controllers (CPU fp32 oracle)
    assert e_pin_inv < 1e-4 and e_pin < 2e-3, (e_pin_inv, e_pin)
    assert e_inv <= 2 * e16_inv + 1e-4, f'inversion {e_inv:.3e} vs fp16-storage oracle {e16_inv:.3e}'
    assert e_out <= 2 * e16_out + 1e-4, f'final latents {e_out:.3e} vs fp16-storage oracle {e16_out:.3e}'
    # round 5: cosine no worse than the yardstick's, the product within 1.5 x the yardstick's own error of
    # the fp16-storage oracle, and an absolute cap on the final latents of the swap path (discrete controller decisions — mask
    # thresholds, replaced maps — make this flow noisier than the plain loops: 1.4e-2 measured, 1.9e-2 for the yardstick)
    assert cosine(got, gold['final']) >= cosine(out16, gold['final']) - 1e-4
    assert rel_l2(got, out16) <= 1.5 * e16_out + 1e-4, (rel_l2(got, out16), e16_out)
    assert e_out <= 3e-2 and e_inv <= 2e-2, (e_inv, e_out)
