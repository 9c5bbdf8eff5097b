"""Build container only: record the configs[2] golden at the SD-1.5 width (tests/cfg3_case.py) with the REFERENCE's own
Prompt-to-Prompt controllers — videoswap/utils/p2p_utils/{attention_store,attention_util,spatial_blend,seq_aligner}.py
imported verbatim from /root/reference (oracle/ref_import.py) — driven by the fp32 oracle UNet on the CPU.

    python tests/golden/make_golden_cfg3.py          # ~10 min on 8 cores; writes tests/golden/cfg3_fullwidth.pt
    python tests/golden/make_golden_cfg3.py bench    # T = 16, 10 + 10 steps (cfg3_case.BENCH): ~2 h on 8 cores

The GPU test (tests/test_cfg3_fullwidth_gpu.py) loads the file on the GPU box, where /root/reference does not exist."""
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.dont_write_bytecode = True


def main():
    import cfg3_case as case
    size = case.BENCH if len(sys.argv) > 1 and sys.argv[1] == 'bench' else case.SMALL
    if len(sys.argv) > 2:
        torch.set_num_threads(int(sys.argv[2]))
    from oracle import adapter as oadapter
    from oracle import ref_import, unet3d
    from videoswap_amd.synthetic import portable_weights_
    mods = ref_import.load_reference_p2p()
    t0 = time.time()
    cfg = unet3d.full_config()
    ora = unet3d.AnimateDiffUNet3DModel(**cfg).eval()
    portable_weights_(ora, seed=case.SEED_W)
    oad = oadapter.SparsePointAdapter(1280, list(cfg['block_out_channels'])).eval()
    portable_weights_(oad, seed=case.SEED_A)
    for m in (ora, oad):                              # the product holds fp16 weights: the oracle gets the same values
        for p in m.parameters():
            p.data = p.data.half().float()
    print(f'models built in {time.time() - t0:.0f} s', flush=True)
    inv, out = case.oracle_flow(ora, oad, mods['attention_store'].AttentionStore,
                                mods['attention_util'].make_controller,
                                log=lambda s: print(f'[{time.time() - t0:.0f} s] {s}', flush=True), size=size)
    w = sum(float(p.detach().double().abs().sum()) for p in ora.parameters())
    torch.save({'inverted': inv, 'final': out, 'weights_abs_sum': w,
                'steps': size.steps, 'frames': size.frames, 'blend': size.blend,
                'controllers': 'reference p2p_utils verbatim', 'oracle': 'oracle.unet3d fp32, CPU'},
               os.path.join(HERE, size.golden))
    print(f'wrote {size.golden} in {time.time() - t0:.0f} s; |inv| {float(inv.norm()):.4f} |out| {float(out.norm()):.4f} '
          f'weights {w:.6e}')


if __name__ == '__main__':
    main()
