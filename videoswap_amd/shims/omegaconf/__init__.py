"""Shim of `omegaconf` (test.py:15,58,69,133): PyYAML-backed `OmegaConf.load / create / to_container`."""
from videoswap_amd.config import DictConfig, ListConfig, OmegaConf  # noqa: F401
