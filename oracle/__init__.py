"""TEST INFRASTRUCTURE — CPU oracle of the VideoSwap denoising path.

A plain-PyTorch fp32 restatement of the reference algorithm (showlab/VideoSwap @ 2024-12-20), used ONLY as the
checker: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it; nothing under
videoswap_amd/ does.  Pinning status:
  * oracle/unet3d.py is checked against the reference's own unet.py / unet_blocks.py / attention.py /
    motion_module.py / resnet.py imported verbatim in the build container (oracle/ref_import.py) and against the
    golden vectors those produced (tests/golden/);
  * the diffusers==0.19.3 pieces (oracle/diffusers_restated.py) are restated from the published source and are
    PARITY UNPINNED (the package is not installable here and the reference has no tests).
"""
