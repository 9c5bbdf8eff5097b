"""Generate tests/golden/*.pt from the REFERENCE's own model code (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference (showlab/VideoSwap @ /root/reference) is imported verbatim through oracle/ref_import.py (its
diffusers dependency is replaced by the restatement in oracle/diffusers_restated.py — the only unpinned part).
Weights are NOT stored: they are regenerated from the seed by oracle.unet3d.synth_weights_ (a checksum is stored
and verified by the tests); inputs and the reference's outputs are stored as fp32 tensors.
"""
import os
import sys

import torch

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_import, unet3d  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def weights_checksum(model):
    return float(sum(p.double().abs().sum() for p in model.state_dict().values()))


def main():
    ref = ref_import.load_reference_models()
    cfg = unet3d.tiny_config()
    src = unet3d.AnimateDiffUNet3DModel(**cfg).eval()
    unet3d.synth_weights_(src, seed=1234)
    model = ref.AnimateDiffUNet3DModel(**cfg).eval()
    model.load_state_dict(src.state_dict(), strict=True)
    g = torch.Generator().manual_seed(0)
    cases = {}
    with torch.no_grad():
        x = torch.randn(1, 4, 4, 16, 16, generator=g)
        txt = torch.randn(1, 77, 64, generator=g)
        cases['plain_T4_16x16'] = dict(sample=x, timestep=481, text=txt, residuals=None,
                                       out=model(x, torch.tensor(481), txt).sample)
        x = torch.randn(2, 4, 3, 16, 24, generator=g)
        txt = torch.randn(2, 77, 64, generator=g)
        res = [torch.randn(6, c, 16 // s, 24 // s, generator=g) * 0.1
               for c, s in zip((64, 128, 256, 256), (1, 2, 4, 8))]
        out = model(x, torch.tensor(21), txt, down_block_additional_residuals=[r.clone() for r in res]).sample
        cases['cfg_adapter_T3_16x24'] = dict(sample=x, timestep=21, text=txt, residuals=res, out=out)
        x = torch.randn(1, 4, 2, 8, 8, generator=g)
        txt = torch.randn(1, 77, 64, generator=g)
        cases['first_inverse_step_t-19'] = dict(sample=x, timestep=-19, text=txt, residuals=None,
                                                out=model(x, torch.tensor(-19), txt).sample)
    blob = dict(config=cfg, weight_seed=1234, weights_checksum=weights_checksum(src), cases=cases,
                generator='tests/golden/make_golden.py on the reference imported verbatim (oracle/ref_import.py)')
    path = os.path.join(HERE, 'unet_tiny.pt')
    torch.save(blob, path)
    print('wrote', path, os.path.getsize(path), 'bytes; checksum', blob['weights_checksum'])


if __name__ == '__main__':
    main()
