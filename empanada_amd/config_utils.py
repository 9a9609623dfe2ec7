"""`config_utils.load_train_config / load_inference_config` (scripts/inference3d_multigpu.py:33,390-391) do not
exist in the reference; they are yaml loaders with the `BASE:` inheritance of
empanada/config_loaders.py:33-70 (a config may name a base file whose keys it overrides, recursively)."""
import os

import yaml

__all__ = ['load_config', 'read_yaml', 'load_train_config', 'load_inference_config']


def _merge(base, update):
    for k, v in update.items():
        if isinstance(v, dict) and isinstance(base.get(k), dict):
            _merge(base[k], v)
        else:
            base[k] = v
    return base


def read_yaml(url):
    """config_loaders.py:10-17"""
    with open(url, mode='r') as handle:
        return yaml.load(handle, Loader=yaml.SafeLoader)


def load_config(url):
    """config_loaders.py:33-70 -- yaml with an optional top-level `BASE: <relative path>` that is loaded first."""
    with open(url, mode='r') as handle:
        config = yaml.load(handle, Loader=yaml.SafeLoader) or {}
    base = config.pop('BASE', None)
    if base is not None:
        base_url = base if os.path.isabs(base) else os.path.join(os.path.dirname(url), base)
        config = _merge(load_config(base_url), config)
    return config


def load_train_config(url):
    config = load_config(url)
    for key in ('MODEL', 'TRAIN'):
        assert key in config, f"training config needs a {key} section"
    return config


def load_inference_config(url):
    config = load_config(url)
    for key in ('engine_params', 'matcher_params', 'labels'):
        assert key in config.get('INFERENCE', config), f"inference config needs {key}"
    return config
