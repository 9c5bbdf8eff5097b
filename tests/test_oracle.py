"""CPU tests: the oracle against the golden vectors produced by the reference's own code, against the reference
imported verbatim (when /root/reference is present), host-side logic of the product, and the C ABI surface."""
import ctypes
import os
import re

import pytest
import torch

from util import ROOT, load_golden, oracle_unet, rel_l2, weights_checksum


def test_oracle_matches_reference_golden():
    blob = load_golden('unet_tiny.pt')
    model = oracle_unet(blob['config'], blob['weight_seed'])
    assert abs(weights_checksum(model) - blob['weights_checksum']) < 1e-6 * blob['weights_checksum'], \
        'synthetic-weight generator drifted: goldens no longer apply'
    for name, case in blob['cases'].items():
        res = None if case['residuals'] is None else [r.clone() for r in case['residuals']]
        with torch.no_grad():
            out = model(case['sample'], torch.tensor(case['timestep']), case['text'],
                        down_block_additional_residuals=res).sample
        assert rel_l2(out, case['out']) < 1e-5, name


def test_oracle_equals_reference_code_verbatim():
    from oracle import ref_import, unet3d
    if not ref_import.available():
        pytest.skip('/root/reference is not present on this machine')
    ref = ref_import.load_reference_models()
    cfg = unet3d.tiny_config()
    o = oracle_unet(cfg)
    r = ref.AnimateDiffUNet3DModel(**cfg).eval()
    missing, unexpected = r.load_state_dict(o.state_dict(), strict=True)   # identical key set
    assert not missing and not unexpected
    g = torch.Generator().manual_seed(7)
    x, txt = torch.randn(2, 4, 3, 16, 16, generator=g), torch.randn(2, 77, 64, generator=g)
    with torch.no_grad():
        assert rel_l2(o(x, torch.tensor(301), txt).sample, r(x, torch.tensor(301), txt).sample) < 1e-6
    # the reference's ResnetBlock3D: 5-D GroupNorm pools statistics over frames (differs from per-frame GN)
    import sys
    rb = sys.modules['videoswap.models.animatediff_models.resnet'].ResnetBlock3D(
        in_channels=64, out_channels=64, temb_channels=32, groups=32, eps=1e-5)
    ob = unet3d.ResnetBlock3D(64, 64, 32, 32, 1e-5)
    ob.load_state_dict(rb.state_dict())
    x, temb = torch.randn(1, 64, 3, 8, 8, generator=g), torch.randn(1, 32, generator=g)
    with torch.no_grad():
        assert rel_l2(ob(x, temb), rb(x, temb)) < 1e-6
        per_frame = torch.stack([ob(x[:, :, i:i + 1], temb) for i in range(3)], 2)[:, :, :, 0]
        assert rel_l2(per_frame, rb(x, temb)) > 1e-3


def test_zero_proj_out_makes_motion_module_identity():
    from oracle import unet3d
    mm = unet3d.VanillaTemporalModule(64, temporal_position_encoding=True)
    x = torch.randn(1, 64, 4, 4, 4)
    with torch.no_grad():
        assert torch.equal(mm(x), x)      # AnimateDiff zero-init (motion_module.py:76-77)


def test_ddim_inversion_then_sampling_roundtrip():
    """DDIM inversion followed by DDIM sampling with the same eps-network returns the input up to the step error."""
    from oracle.diffusers_restated import SD15_SCHEDULER_CONFIG, DDIMInverseScheduler, DDIMScheduler
    sch, inv = DDIMScheduler(**SD15_SCHEDULER_CONFIG), DDIMInverseScheduler(**SD15_SCHEDULER_CONFIG)
    sch.set_timesteps(50)
    inv.set_timesteps(50)
    assert sch.timesteps.tolist() == list(range(981, 0, -20))
    assert inv.timesteps.tolist() == [-19] + list(range(1, 962, 20))
    x0 = torch.randn(1, 4, 2, 8, 8)

    def eps(x, t):
        return 0.1 * torch.tanh(x) + 0.01
    x = x0
    for t in inv.timesteps:
        x = inv.step(eps(x, t), t, x).prev_sample
    for t in sch.timesteps:
        x = sch.step(eps(x, t), t, x).prev_sample
    assert rel_l2(x, x0) < 2e-2


def test_inverse_scheduler_first_step_known_answer():
    """Hand-computed coefficients of the first inversion step (t = -19 -> 1) of the SD-1.5 scheduler config
    (scaled_linear 0.00085..0.012, 1000 steps, set_alpha_to_one False): alpha_bar(-19) is `initial_alpha_cumprod`
    = alphas_cumprod[0] = 1 - 0.00085, NOT 1; alpha_bar(1) = (1 - 0.00085) * (1 - beta_1)."""
    import math
    from oracle import diffusers_restated as dr
    from videoswap_amd import compat
    b0 = 0.00085
    b1 = (math.sqrt(0.00085) + (math.sqrt(0.012) - math.sqrt(0.00085)) / 999.0) ** 2
    want = (1.0 - b0, (1.0 - b0) * (1.0 - b1))
    for mod in (dr, compat):
        inv = mod.DDIMInverseScheduler.from_config(mod.DDIMScheduler(**mod.SD15_SCHEDULER_CONFIG).config)
        inv.set_timesteps(50)
        assert int(inv.timesteps[0]) == -19
        assert inv.coefficients(inv.timesteps[0]) == pytest.approx(want, rel=2e-6), mod.__name__
        one = mod.DDIMInverseScheduler(**{**mod.SD15_SCHEDULER_CONFIG, 'set_alpha_to_one': True})
        one.set_timesteps(50)
        assert one.coefficients(-19)[0] == 1.0
    # the step itself (oracle): x_1 = sqrt(a1) * (x - sqrt(1 - a0) e) / sqrt(a0) + sqrt(1 - a1) e
    inv = dr.DDIMInverseScheduler(**dr.SD15_SCHEDULER_CONFIG)
    inv.set_timesteps(50)
    x, e = torch.full((1, 1), 0.5), torch.full((1, 1), -0.25)
    a0, a1 = want
    ref = math.sqrt(a1) * (0.5 - math.sqrt(1 - a0) * -0.25) / math.sqrt(a0) + math.sqrt(1 - a1) * -0.25
    assert float(inv.step(e, -19, x).prev_sample) == pytest.approx(ref, rel=1e-6)


def test_product_schedulers_match_oracle_coefficients():
    from oracle import diffusers_restated as dr
    from videoswap_amd import compat
    for n in (2, 50):
        a, b = compat.DDIMScheduler(**compat.SD15_SCHEDULER_CONFIG), dr.DDIMScheduler(**dr.SD15_SCHEDULER_CONFIG)
        a.set_timesteps(n); b.set_timesteps(n)
        assert a.timesteps.tolist() == b.timesteps.tolist()
        for t in a.timesteps:
            assert a.coefficients(t) == pytest.approx(b.coefficients(t), rel=1e-7)
        a, b = (compat.DDIMInverseScheduler(**compat.SD15_SCHEDULER_CONFIG),
                dr.DDIMInverseScheduler(**dr.SD15_SCHEDULER_CONFIG))
        a.set_timesteps(n); b.set_timesteps(n)
        assert a.timesteps.tolist() == b.timesteps.tolist()
        for t in a.timesteps:
            assert a.coefficients(t) == pytest.approx(b.coefficients(t), rel=1e-7)


def test_product_unet_has_reference_state_dict_keys():
    from oracle import unet3d
    from videoswap_amd.unet import AnimateDiffUNet3DModel
    cfg = unet3d.tiny_config()
    prod = AnimateDiffUNet3DModel(**cfg)
    ora = unet3d.AnimateDiffUNet3DModel(**cfg)
    ps, os_ = prod.state_dict(), ora.state_dict()
    assert list(ps.keys()) == list(os_.keys())
    for k in ps:
        assert ps[k].shape == os_[k].shape, k
    # zero-initialised motion-module proj_out, like the reference (motion_module.py:76-77)
    assert all(float(v.abs().sum()) == 0 for k, v in ps.items() if 'temporal_transformer.proj_out' in k)
    # the module tree the reference's processor registration walks (edlora_util.py:85-99)
    names = [n for n, m in prod.named_modules() if m.__class__.__name__ == 'Attention']
    assert sum('attn1' in n for n in names) == 16 and sum('attn2' in n for n in names) == 16
    assert sum('attention_blocks' in n for n in names) == 40
    # load_state_dict round trip keeps conv weights in the OHWI (channels_last) layout the kernel streams
    prod.load_state_dict(ora.state_dict())
    w = prod.down_blocks[0].resnets[0].conv1.weight
    assert w.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(w, ora.down_blocks[0].resnets[0].conv1.weight)
    assert prod.half().down_blocks[0].resnets[0].conv1.weight.is_contiguous(memory_format=torch.channels_last)


def test_c_abi_exports_every_declared_symbol():
    from videoswap_amd import _lib
    header = open(os.path.join(ROOT, 'include', 'vsx.h')).read()
    declared = set(re.findall(r'\b(vsx_[a-z0-9_]+)\s*\(', header))
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert _lib.load().vsx_abi_version() == _lib.VSX_ABI_VERSION
    assert ctypes.sizeof(_lib.GemmDesc) == 46 * 8      # ABI v4: + pad_lo, pad_hi; ABI 7: + rowscale, colvec; ABI 8: + rowstats, rowstats_parts


def test_ops_refuse_cpu_tensors():
    from videoswap_amd import ops
    from videoswap_amd._lib import VsxError
    with pytest.raises(VsxError):
        ops.linear(torch.zeros(8, 8, dtype=torch.float16), torch.zeros(8, 8, dtype=torch.float16))


def test_oracle_adapter_equals_reference_code_verbatim():
    from oracle import adapter, ref_import
    if not ref_import.available():
        pytest.skip('/root/reference is not present on this machine')
    ref_import.load_reference_models()
    ref_import._load('videoswap.utils.registry', 'videoswap/utils/registry.py')
    ra = ref_import._load('videoswap.models.adapter_model', 'videoswap/models/adapter_model.py')
    chans = [64, 128, 256, 256]
    r = ra.SparsePointAdapter(embedding_channels=1280, channels=chans).eval()
    o = adapter.SparsePointAdapter(1280, chans).eval()
    o.load_state_dict(r.state_dict(), strict=True)
    g = torch.Generator().manual_seed(0)
    W, H = 192, 128
    tr = torch.rand(1, 3, 6, 2, generator=g) * torch.tensor([float(W), float(H)])
    tr[0, 0, 1] = -1
    tr[0, 1, 2] = torch.tensor([W - 0.5, H - 0.5])
    emb = torch.randn(1, 6, 1280, generator=g)
    with torch.no_grad():
        a, b = r(tr, (W, H), emb, index_list=[0, 1, 2, 4]), o(tr, (W, H), emb, index_list=[0, 1, 2, 4])
    for x, y in zip(a, b):
        assert torch.equal(x, y)


def test_product_adapter_has_reference_state_dict_keys():
    from oracle import adapter
    from videoswap_amd.adapter import SparsePointAdapter
    p, o = SparsePointAdapter(), adapter.SparsePointAdapter()
    assert list(p.state_dict().keys()) == list(o.state_dict().keys())
    assert p.config.downsample_rate == [8, 16, 32, 64]


def test_lora_merge_matches_reference_formula():
    """W' = W + alpha * up @ down on exactly the keys of convert_edlora_to_diffusers.py:46-53, conv weights via the
    squeezed 1x1 factors; restoring the snapshot brings the weights back bit-exactly."""
    from oracle import unet3d
    from videoswap_amd.edlora import merge_lora_into_weight
    from videoswap_amd.unet import AnimateDiffUNet3DModel
    cfg = unet3d.tiny_config()
    m = AnimateDiffUNet3DModel(**cfg)
    sd = m.state_dict()
    g = torch.Generator().manual_seed(4)
    lora, touched = {}, []
    for k, w in sd.items():
        if any(k.endswith(s) for s in ('attn2.to_q.weight', 'attn1.to_out.0.weight', 'ff.net.0.proj.weight',
                                       'attentions.0.proj_in.weight')) and 'motion' not in k:
            base = k[:-len('weight')]
            out_f, in_f = w.shape[0], w.shape[1]
            down, up = torch.randn(4, in_f, generator=g) * 0.01, torch.randn(out_f, 4, generator=g) * 0.01
            if w.dim() == 4:
                down, up = down[:, :, None, None], up[:, :, None, None]
            lora[base + 'lora_down.weight'], lora[base + 'lora_up.weight'] = down, up
            touched.append(k)
    merged = merge_lora_into_weight(sd, lora, 'unet', alpha=0.7)
    assert len(touched) > 10
    for k in sd:
        if k in touched:
            d, u = lora[k[:-6] + 'lora_down.weight'], lora[k[:-6] + 'lora_up.weight']
            delta = (u.squeeze() @ d.squeeze()).reshape(sd[k].shape)
            assert torch.allclose(merged[k], sd[k] + 0.7 * delta)
        else:
            assert torch.equal(merged[k], sd[k])
    import copy
    snapshot = copy.deepcopy(sd)
    m.load_state_dict(merged)
    m.load_state_dict(snapshot)
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), snapshot.values()))
