"""Shim of the `diffusers` names the reference's scripts import at top level (test.py:14, train.py:12-13): resolved to
the in-repo facade (videoswap_amd.compat / vae).  Only reachable through `python -m videoswap_amd.dropin`."""
from videoswap_amd.compat import DDIMInverseScheduler, DDIMScheduler, DDPMScheduler  # noqa: F401
from videoswap_amd.vae import AutoencoderKL  # noqa: F401

__version__ = '0.19.3+vsx'
__path__ = [__import__('os').path.dirname(__file__)]
