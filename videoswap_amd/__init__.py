"""videoswap_amd — MI355X-native denoising path for VideoSwap.

Host classes with the reference's names and call signatures (AnimateDiffUNet3DModel, SparsePointAdapter,
VideoSwapPipeline, Attention + processors, Prompt-to-Prompt controllers) over hand-written gfx950 HIP kernels
reached through the C ABI of include/vsx.h (libvsx.so).  Importing the package does not load the library;
the first kernel call does, and raises if it has not been built (`python -m videoswap_amd.build`).
"""
from .compat import MODEL_REGISTRY, PIPELINE_REGISTRY  # noqa: F401

__version__ = '0.1.0'


def build_model(name):
    """videoswap/models/__init__.py:24-32"""
    from . import adapter, unet  # noqa: F401  (registers the classes)
    return MODEL_REGISTRY.get(name)


def build_pipeline(name):
    """videoswap/pipelines/__init__.py:24-32"""
    from . import pipeline, trainer  # noqa: F401  (pipelines/__init__.py:12-21: pipeline_* and trainer_* modules)
    return PIPELINE_REGISTRY.get(name)
