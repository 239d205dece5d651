"""Config loading with ordered overlays and dotted keys -- the YAML surface of the reference (annotator/utils/load.py:16-84).

`load_config([main, overlay, ...])`: the first file is the base, every later file is applied on top; a key such as
`deploy_options.enable_multigpu` descends (creating dicts on the way) and replaces the leaf."""

import json
import os
import pickle

import yaml


def load_config(path):
    if isinstance(path, str):
        return load_config([path])
    assert isinstance(path, (tuple, list))
    assert path
    config = None
    for i, single in enumerate(path):
        loaded = _load_config_single(single)
        config = loaded if i == 0 else _apply_config(config, loaded)
    return config


def _apply_config(base_config, add_config):
    """Overlay add_config on base_config in place; 'a.b.c' keys address nested dict entries (load.py:44-57)."""
    for dest, value in add_config.items():
        target = base_config
        keys = dest.split('.')
        for k in keys[:-1]:
            if k not in target:
                target[k] = dict()
            target = target[k]
        target[keys[-1]] = value
    return base_config


def _load_config_single(path):
    extension = os.path.splitext(path)[1][1:]
    if extension == 'json':
        with open(path) as f:
            return json.load(f)
    if extension == 'yaml':
        with open(path) as f:
            return yaml.safe_load(f)
    if extension == 'pickle':
        with open(path, 'rb') as f:
            return pickle.load(f)
    raise NotImplementedError(f'Unexpected extension {extension}')
