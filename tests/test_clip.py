"""f2 — CLIP text encoder.  The oracle restatement is PINNED against the `transformers` CLIPTextModel installed here
(the one third-party model of the path that is importable); the HIP-backed product class is then compared with the
oracle on the GPU, and its state-dict layout with the transformers-4.25 key names the reference's checkpoints use."""
import pytest
import torch

from util import DEV, rel_l2


def _ids(n=3, vocab=300, seed=1):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(1, vocab - 2, (n, 77), generator=g)
    ids[:, 0] = vocab - 2
    ids[:, 20:] = vocab - 1          # padded with EOS, as the tokenizer does
    return ids


def test_oracle_clip_matches_transformers():
    transformers = pytest.importorskip('transformers')
    from oracle import clip as oclip
    cfg = oclip.tiny_clip_config()
    ora = oclip.synth_weights_(oclip.CLIPTextModel(**cfg)).eval()
    hf_cfg = transformers.CLIPTextConfig(**cfg, bos_token_id=298, eos_token_id=299, pad_token_id=299)
    hf = transformers.CLIPTextModel(hf_cfg).eval()
    sd = ora.state_dict()
    hf_keys = set(hf.state_dict().keys())
    strip = not any(k.startswith('text_model.') for k in hf_keys)        # transformers 5.x dropped the prefix
    missing, unexpected = hf.load_state_dict({(k[len('text_model.'):] if strip else k): v for k, v in sd.items()},
                                             strict=False)
    assert not unexpected and all('position_ids' in m for m in missing), (missing, unexpected)
    ids = _ids()
    with torch.no_grad():
        want = hf(input_ids=ids)[0]
        got = ora(ids)[0]
    assert torch.allclose(got, want, rtol=0, atol=2e-5), float((got - want).abs().max())


def test_product_clip_state_dict_layout():
    from oracle import clip as oclip
    from videoswap_amd.clip import CLIPTextConfig, CLIPTextModel
    cfg = oclip.tiny_clip_config()
    o, p = oclip.CLIPTextModel(**cfg), CLIPTextModel(CLIPTextConfig(**cfg))
    assert list(o.state_dict().keys()) == list(p.state_dict().keys())
    assert 'text_model.embeddings.token_embedding.weight' in p.state_dict()
    assert 'text_model.encoder.layers.1.self_attn.out_proj.weight' in p.state_dict()
    # un-prefixed (transformers 5.x) keys load too; the token table follows the checkpoint's vocabulary size
    sd = {k[len('text_model.'):]: v for k, v in oclip.synth_weights_(oclip.CLIPTextModel(**dict(cfg, vocab_size=310))).state_dict().items()}
    p.load_state_dict(sd, strict=True)
    assert p.get_input_embeddings().num_embeddings == 310
    p.resize_token_embeddings(326)
    assert p.get_input_embeddings().weight.shape == (326, 64) and float(p.get_input_embeddings().weight.detach()[310:].abs().sum()) == 0


@pytest.mark.parametrize('full', [pytest.param(False, marks=pytest.mark.device),
                                  pytest.param(True, marks=pytest.mark.gpu)])
def test_product_clip_matches_oracle(full):
    from oracle import clip as oclip
    from videoswap_amd.clip import CLIPTextConfig, CLIPTextModel
    cfg = dict(vocab_size=1000, hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12,
               max_position_embeddings=77, hidden_act='quick_gelu', layer_norm_eps=1e-5) if full else oclip.tiny_clip_config()
    ora = oclip.synth_weights_(oclip.CLIPTextModel(**cfg)).eval()
    prod = CLIPTextModel(CLIPTextConfig(**cfg)).eval()
    prod.load_state_dict(ora.state_dict(), strict=True)
    prod = prod.to(DEV, torch.float16)
    ids = _ids(n=17, vocab=cfg['vocab_size'])             # 16 per-layer ED-LoRA prompts + the negative prompt
    with torch.no_grad():
        want = ora(ids)[0]
        got = prod(ids)[0].float().cpu()
    e = rel_l2(got, want)
    print(f'clip {"L/14 text" if full else "tiny"}: rel-L2 {e:.2e}')
    assert got.shape == want.shape and e < 5e-3
