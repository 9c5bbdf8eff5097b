"""f1 — the VAE either side of the loop.  CPU: module tree / state-dict keys of the HIP-backed AutoencoderKL equal the
diffusers layout restated in oracle/vae.py (so real `vae/` checkpoints load), legacy attention key names are
accepted, the image processor round-trips.  GPU: encode (moments, mode, sample with a shared noise) and decode against
the fp32 oracle."""
import pytest
import torch

from util import DEV, rel_l2


def test_vae_state_dict_layout_matches_diffusers_layout():
    from oracle import vae as ovae
    from videoswap_amd.vae import SD15_VAE_CONFIG, AutoencoderKL
    for cfg in (ovae.tiny_vae_config(), dict(SD15_VAE_CONFIG)):
        o, p = ovae.AutoencoderKL(**cfg), AutoencoderKL(**cfg)
        os_, ps = o.state_dict(), p.state_dict()
        assert list(os_.keys()) == list(ps.keys())
        for k in os_:
            assert os_[k].shape == ps[k].shape, k
    assert 'encoder.down_blocks.0.downsamplers.0.conv.weight' in ps
    assert 'decoder.up_blocks.0.upsamplers.0.conv.weight' in ps
    assert 'encoder.mid_block.attentions.0.to_out.0.bias' in ps
    assert len(p.config.block_out_channels) == 4 and p.config.scaling_factor == 0.18215


def test_vae_accepts_legacy_attention_names():
    from oracle import vae as ovae
    from videoswap_amd.vae import AutoencoderKL
    cfg = ovae.tiny_vae_config()
    sd = ovae.synth_weights_(ovae.AutoencoderKL(**cfg)).state_dict()
    legacy = {}
    for k, v in sd.items():
        for new, old in (('to_q', 'query'), ('to_k', 'key'), ('to_v', 'value'), ('to_out.0', 'proj_attn')):
            if '.attentions.' in k:
                k = k.replace(f'.{new}.', f'.{old}.')
        legacy[k] = v
    assert any('proj_attn' in k for k in legacy)
    p = AutoencoderKL(**cfg)
    p.load_state_dict(legacy, strict=True)
    assert torch.equal(p.state_dict()['decoder.mid_block.attentions.0.to_out.0.weight'],
                       sd['decoder.mid_block.attentions.0.to_out.0.weight'])


def test_image_processor_roundtrip():
    from PIL import Image
    from videoswap_amd.vae import VaeImageProcessor
    proc = VaeImageProcessor(8)
    g = torch.Generator().manual_seed(0)
    imgs = [Image.fromarray((torch.rand(40, 51, 3, generator=g) * 255).byte().numpy()) for _ in range(2)]
    x = proc.preprocess(imgs)
    assert x.shape == (2, 3, 40, 48) and float(x.min()) >= -1 and float(x.max()) <= 1     # 51 -> 48 (multiple of 8)
    back = proc.postprocess(x, output_type='pil')
    assert back[0].size == (48, 40)
    assert torch.equal(proc.preprocess(back), x)


@pytest.mark.parametrize('full', [pytest.param(False, marks=pytest.mark.device),
                                  pytest.param(True, marks=pytest.mark.gpu)])
def test_vae_encode_decode_match_oracle(full):
    from oracle import vae as ovae
    from videoswap_amd.vae import SD15_VAE_CONFIG, AutoencoderKL
    cfg = dict(SD15_VAE_CONFIG) if full else ovae.tiny_vae_config()
    size = (128, 192) if full else (64, 96)
    o = ovae.synth_weights_(ovae.AutoencoderKL(**cfg)).eval()
    p = AutoencoderKL(**cfg).eval()
    p.load_state_dict(o.state_dict(), strict=True)
    p = p.to(DEV, torch.float16)
    g = torch.Generator().manual_seed(3)
    x = torch.rand(3, 3, *size, generator=g) * 2 - 1
    odev = o.to(DEV)
    with torch.no_grad():
        ref_m = odev.moments(x.to(DEV)).cpu()
        dist = p.encode(x.half().to(DEV)).latent_dist
        assert dist.parameters.shape == ref_m.shape
        e_mom = rel_l2(dist.parameters.float().cpu(), ref_m)
        noise = torch.randn(ref_m[:, :4].shape, generator=g)
        e_mode = rel_l2(dist.mode().float().cpu(), torch.chunk(ref_m, 2, 1)[0])
        z = torch.randn(3, 4, size[0] // 8, size[1] // 8, generator=g)
        ref_d = odev.decode(z.to(DEV)).cpu()
        dec = p.decode(z.half().to(DEV)).sample
        e_dec = rel_l2(dec.float().cpu(), ref_d)
        p.enable_slicing()
        dec2 = p.decode(z.half().to(DEV), return_dict=False)[0]
    print(f'vae {"SD-1.5" if full else "tiny"}: moments {e_mom:.2e}, mode {e_mode:.2e}, decode {e_dec:.2e}')
    assert dec.shape == (3, 3, *size) and torch.equal(dec, dec2)
    assert e_mom < 6e-3 and e_mode < 6e-3 and e_dec < 6e-3
    s = dist.sample(torch.Generator().manual_seed(1))
    assert s.shape == (3, 4, size[0] // 8, size[1] // 8) and torch.isfinite(s).all()
    assert noise.shape == s.shape
