"""TEST INFRASTRUCTURE (build container only) — import the reference's model files VERBATIM.

/root/reference cannot be imported as a package here (diffusers / torchvision / xformers are not installed,
SURVEY.md §8c), but its model files only need a handful of diffusers symbols.  This module registers stub
`diffusers.*` modules backed by oracle/diffusers_restated.py, then loads
videoswap/models/animatediff_models/{resnet,attention,motion_module,unet_blocks,unet}.py and
videoswap/utils/p2p_utils/* straight from /root/reference (never copied, never written: bytecode caching is
disabled).  Used by tests/test_oracle_vs_reference.py and tests/golden/make_golden.py to pin oracle/unet3d.py.
/root/reference does not exist on the GPU box: nothing that runs there imports this file.
"""
import importlib.util
import logging
import os
import sys
import types

REFERENCE_ROOT = os.environ.get('VIDEOSWAP_REFERENCE', '/root/reference')


def available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, 'videoswap/models/animatediff_models/unet.py'))


def _module(name, **attrs):
    m = types.ModuleType(name)
    # a real spec: importlib.util.find_spec (transformers probes optional packages with it) rejects `__spec__ = None`
    m.__spec__ = importlib.util.spec_from_loader(name, loader=None)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _package(name, path=None):
    m = _module(name)
    m.__path__ = [path] if path else []
    return m


def install_stubs():
    from . import diffusers_restated as dr
    if 'diffusers' in sys.modules and getattr(sys.modules['diffusers'], '_vsx_stub', False):
        return
    d = _package('diffusers')
    d._vsx_stub = True
    _module('diffusers.configuration_utils', ConfigMixin=dr.ConfigMixin, register_to_config=dr.register_to_config,
            FrozenDict=dr.FrozenDict)
    _package('diffusers.models')
    _module('diffusers.models.attention', AdaLayerNorm=dr.AdaLayerNorm, Attention=dr.Attention,
            FeedForward=dr.FeedForward)
    _module('diffusers.models.attention_processor', Attention=dr.Attention, AttnProcessor=dr.AttnProcessor,
            AttnProcessor2_0=dr.AttnProcessor2_0, XFormersAttnProcessor=dr.XFormersAttnProcessor)
    _module('diffusers.models.modeling_utils', ModelMixin=dr.ModelMixin)
    _module('diffusers.models.embeddings', TimestepEmbedding=dr.TimestepEmbedding, Timesteps=dr.Timesteps)
    lg = types.SimpleNamespace(get_logger=logging.getLogger)
    u = _package('diffusers.utils')
    u.BaseOutput = dr.BaseOutput
    u.logging = lg
    u.WEIGHTS_NAME = 'diffusion_pytorch_model.bin'
    _module('diffusers.utils.import_utils', is_xformers_available=lambda: False)
    if 'torchvision' not in sys.modules:   # motion_module.py:8 imports it and never uses it
        tv = _package('torchvision')
        tv.utils = _module('torchvision.utils', save_image=lambda *a, **k: None)
    if 'cv2' not in sys.modules:           # ptp_utils.py:5 (drawing helpers only)
        _module('cv2')
    if 'omegaconf' not in sys.modules:     # ptp_utils.py:7 (isinstance check against DictConfig only)
        oc = _package('omegaconf')
        oc.dictconfig = _module('omegaconf.dictconfig', DictConfig=type('DictConfig', (dict,), {}))


def _load(dotted, relpath):
    if dotted in sys.modules:
        return sys.modules[dotted]
    path = os.path.join(REFERENCE_ROOT, relpath)
    spec = importlib.util.spec_from_file_location(dotted, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[dotted] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference_models():
    """Returns the reference's `unet` module (AnimateDiffUNet3DModel etc.), loaded from /root/reference."""
    if not available():
        raise RuntimeError(f'{REFERENCE_ROOT} is not present')
    sys.dont_write_bytecode = True
    install_stubs()
    base = 'videoswap/models/animatediff_models'
    for pkg in ('videoswap', 'videoswap.models', 'videoswap.models.animatediff_models', 'videoswap.utils',
                'videoswap.utils.p2p_utils'):
        if pkg not in sys.modules:
            _package(pkg)
    for name in ('resnet', 'attention', 'motion_module', 'unet_blocks', 'unet'):
        _load(f'videoswap.models.animatediff_models.{name}', f'{base}/{name}.py')
    return sys.modules['videoswap.models.animatediff_models.unet']


def load_reference_p2p():
    """attention_store / ptp_utils / seq_aligner / spatial_blend / attention_util of the reference, verbatim."""
    load_reference_models()
    base = 'videoswap/utils/p2p_utils'
    _load('videoswap.utils.edlora_util', 'videoswap/utils/edlora_util.py')
    out = {}
    for name in ('attention_store', 'seq_aligner', 'ptp_utils', 'spatial_blend', 'attention_util'):
        out[name] = _load(f'videoswap.utils.p2p_utils.{name}', f'{base}/{name}.py')
    return out
