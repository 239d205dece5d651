"""Dumping run options and training results -- the save-directory contract of the reference
(annotator/utils/dump.py:17-82): `save_path/options.yaml` (read back by evaluate) and `save_path/results.pkl`."""

import json
import os
import pickle

import yaml


def dump_options(path, avoid_overwrite=False, **options):
    """Writes `options` to `path` (format by extension).  While a file exists at `path` the base name gets a '_'
    appended before the extension (dump.py:30-33), so an earlier run's options are never overwritten."""
    while os.path.exists(path):
        root, ext = os.path.splitext(os.path.basename(path))
        path = os.path.join(os.path.dirname(path), '{}_{}'.format(root, ext))
    format_ = os.path.splitext(path)[1][1:]
    os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
    if format_ == 'json':
        with open(path, 'w') as f:
            json.dump(options, f)
    elif format_ == 'yaml':
        with open(path, 'w') as f:
            yaml.safe_dump(options, f)
    elif format_ == 'pickle':
        with open(path, 'wb') as f:
            pickle.dump(options, f)
    else:
        raise NotImplementedError(f'Umimplemented format {format_}')
    return path


def dump_train_results(path, train_results, format_='pickle'):
    """`train_results` is the History-like object returned by engine.TFKerasModel.train (dump.py:52-82)."""
    format_ = format_.lower()
    os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
    content = {
        'epoch': train_results.epoch,
        'history': train_results.history,
        'params': train_results.params,
        'model': type(train_results.model).__name__,
    }
    if format_ == 'pickle':
        with open(path, 'wb') as f:
            pickle.dump(content, f)
    elif format_ == 'yaml':
        with open(path, 'w') as f:
            yaml.safe_dump(content, f)
    else:
        raise NotImplementedError(f'Umimplemented format {format_}')
