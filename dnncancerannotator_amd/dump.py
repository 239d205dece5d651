"""Run options and training results on disk: the part of the save-directory layout that `annotator evaluate` and later
tooling read back -- `<save_path>/options.yaml` (runs/evaluate.py:65-66 reloads it) and `<save_path>/results.pkl`.
Behavioural counterpart of annotator/utils/dump.py:17-82 (same function names, arguments, file formats and file-naming rule),
written against tests/test_host.py."""

import json
import os
import pickle

import yaml

# extension / format name -> (file mode, writer)
_WRITERS = {
    'json': ('w', lambda obj, f: json.dump(obj, f)),
    'yaml': ('w', lambda obj, f: yaml.safe_dump(obj, f)),
    'pickle': ('wb', lambda obj, f: pickle.dump(obj, f)),
}


def _write(path, obj, format_):
    if format_ not in _WRITERS:
        raise NotImplementedError('Unimplemented format %r (expected one of %s)' % (format_, ', '.join(sorted(_WRITERS))))
    mode, writer = _WRITERS[format_]
    os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
    with open(path, mode) as f:
        writer(obj, f)


def _free_name(path):
    """First of path, stem_.ext, stem__.ext, ... that does not exist yet: an earlier run's file is never replaced."""
    folder, name = os.path.split(path)
    stem, ext = os.path.splitext(name)
    while os.path.exists(os.path.join(folder, stem + ext)):
        stem += '_'
    return os.path.join(folder, stem + ext)


def dump_options(path, avoid_overwrite=False, **options):
    """Store the keyword `options` at `path`; the extension picks the format (json / yaml / pickle).  Returns the path written,
    which differs from `path` when that name was taken.  (`avoid_overwrite` is accepted for signature compatibility: as in
    the reference the taken-name rule applies regardless.)"""
    target = _free_name(path)
    _write(target, options, os.path.splitext(target)[1].lstrip('.'))
    return target


def dump_train_results(path, train_results, format_='pickle'):
    """Store what engine.TFKerasModel.train returned (a keras-History-like object: .epoch, .history, .params, .model) as a
    plain dict, pickled or as YAML."""
    record = dict(epoch=train_results.epoch, history=train_results.history, params=train_results.params,
                  model=type(train_results.model).__name__)
    _write(path, record, format_.lower())
    return path
