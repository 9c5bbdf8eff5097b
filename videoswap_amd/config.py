"""Config surface of the reference (BASELINE.json configs[0]): the `options/**.yml` files are read with
PyYAML (they are safe_load-clean: plain mappings, `!!float` tags, `~` nulls) and turned into the plain dict
`test.py` passes around (test.py:128-136: `OmegaConf.to_container(OmegaConf.load(path), resolve=True)`).

`OmegaConf` here is the small part of omegaconf's API the reference touches — `load`, `create`, `to_container`,
attribute access on the loaded node (`OmegaConf.load(p).unet_additional_kwargs`, test.py:58), `${a.b}` interpolation —
so that the shim package `videoswap_amd/shims/omegaconf` can stand in for the absent dependency.
"""
import copy
import re

import yaml


class DictConfig(dict):
    """dict with attribute access, as omegaconf's DictConfig is used by the reference (test.py:58,69)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name) from None

    def __setattr__(self, name, value):
        self[name] = value


class ListConfig(list):
    pass


def _wrap(node):
    if isinstance(node, dict):
        return DictConfig({k: _wrap(v) for k, v in node.items()})
    if isinstance(node, (list, tuple)):
        return ListConfig(_wrap(v) for v in node)
    return node


def _unwrap(node):
    if isinstance(node, dict):
        return {k: _unwrap(v) for k, v in node.items()}
    if isinstance(node, (list, tuple)):
        return [_unwrap(v) for v in node]
    return node


_INTERP = re.compile(r'\$\{([^}]+)\}')


def _lookup(root, dotted):
    node = root
    for part in dotted.split('.'):
        node = node[int(part)] if isinstance(node, list) else node[part]
    return node


def _resolve(node, root):
    if isinstance(node, dict):
        return {k: _resolve(v, root) for k, v in node.items()}
    if isinstance(node, list):
        return [_resolve(v, root) for v in node]
    if isinstance(node, str):
        whole = _INTERP.fullmatch(node)
        if whole:
            return _resolve(copy.deepcopy(_lookup(root, whole.group(1))), root)
        return _INTERP.sub(lambda m: str(_resolve(_lookup(root, m.group(1)), root)), node)
    return node


class OmegaConf:
    @staticmethod
    def load(path):
        with open(path, 'r') as f:
            return _wrap(yaml.safe_load(f) or {})

    @staticmethod
    def create(obj=None):
        if isinstance(obj, str):
            obj = yaml.safe_load(obj)
        return _wrap(obj if obj is not None else {})

    @staticmethod
    def to_container(cfg, resolve=False, **unused):
        plain = _unwrap(cfg)
        return _resolve(plain, plain) if resolve else plain

    @staticmethod
    def to_yaml(cfg, **unused):
        return yaml.safe_dump(_unwrap(cfg), sort_keys=False)


def load_options(path, overrides=None):
    """The dict `test.py` builds from `-opt <yml>`; `overrides` maps dotted keys to values
    (e.g. {'datasets.num_frames': 4, 'val.editing_config.num_inference_steps': 2} for BASELINE.json configs[0])."""
    opt = OmegaConf.to_container(OmegaConf.load(path), resolve=True)
    for dotted, value in (overrides or {}).items():
        node = opt
        parts = dotted.split('.')
        for part in parts[:-1]:
            node = node.setdefault(part, {})
        node[parts[-1]] = value
    return opt
