"""Shared helpers of the parity tests (oracle side = checker only)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
# where the product side of a `device` test lives: the GPU when there is one, else CPU tensors under tests/host_emulation.py
DEV = 'cuda' if torch.cuda.is_available() else 'cpu'


def sync():
    if DEV == 'cuda':
        torch.cuda.synchronize()


def load_golden(name):
    return torch.load(os.path.join(GOLDEN, name), map_location='cpu', weights_only=False)


def rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-12))


def cosine(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm()).clamp_min(1e-12))


def weights_checksum(model):
    return float(sum(p.double().abs().sum() for p in model.state_dict().values()))


def oracle_unet(cfg=None, seed=1234):
    from oracle import unet3d
    cfg = cfg or unet3d.tiny_config()
    m = unet3d.AnimateDiffUNet3DModel(**cfg).eval()
    unet3d.synth_weights_(m, seed=seed)
    return m


def product_unet_from(oracle_model, cfg, device=DEV, dtype=torch.float16):
    """Build the HIP-backed UNet and load the oracle's (reference-keyed) state dict into it."""
    from videoswap_amd.unet import AnimateDiffUNet3DModel
    m = AnimateDiffUNet3DModel(**cfg).eval()
    missing, unexpected = m.load_state_dict(oracle_model.state_dict(), strict=True)
    assert not missing and not unexpected
    return m.to(device=device, dtype=dtype)
