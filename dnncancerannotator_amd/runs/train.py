"""`annotator train` -- annotator/runs/train.py:21-94: dump the options, build datasets, train, dump the results."""

import os

from .. import data, dump, engine, load


def make_dataset(paths, options, training):
    """Dataset for `--data_path`.  Supported sources: the reference's `.tfrecords` exam files (tfrecord.py; for training with the
    `augment_options` of data_options.train -- random crop / flip / contrast run on the device, random_warp is skipped),
    `synthetic[:HxW[xC]]` (seeded synthetic slices) and `.npz` files holding `x` [N,H,W,C] in [0,1] and `y` [N,H,W].
    The image-folder pipeline (data.py:170-180) is outside the accelerated hot path."""
    batch_size = options.get('batch_size', 8)
    first = paths[0]
    if first.startswith('synthetic'):
        dims = [int(v) for v in first.split(':')[1].split('x')] if ':' in first else []
        h, w = (dims + [512, 512])[:2] if len(dims) >= 2 else (512, 512)
        c = dims[2] if len(dims) > 2 else 1
        return data.SyntheticDataset(batch_size, h, w, c, repeat=training, n_batches=4 if training else 2)
    if all(p.endswith('.tfrecords') for p in paths):          # the reference's exam files (data.py:166-169)
        from .. import distributed
        from ..tfrecord import TFRecordDataset
        ctx = distributed.context()
        slice_types = options.get('slice_types', ['TRA', 'ADC', 'DWI', 'DCEE', 'DCEL', 'label'])
        # train_ds (data.py:62-111): output_size defaults to 256 x 256 and there is always at least the random crop;
        # eval_ds (data.py:114-143): centre crop to output_size (default 512 x 512), no augmentation
        return TFRecordDataset(paths, slice_types, batch_size,
                               output_size=tuple(options.get('output_size', (256, 256) if training else (512, 512))),
                               repeat=training, drop_remainder=training,
                               augment_options=options.get('augment_options') if training else False,
                               buffer_size=options.get('buffer_size', 0) if training else 0,
                               device_convert=not training,      # evaluation: uint8 to the device, / 255 and the split there
                               shard=(ctx.rank, ctx.world),      # data parallel: every rank assembles only its part of a batch
                               normalize_exams=bool(options.get('normalize_exams', True)) if training else False)   # data.py:68,137
    if all(p.endswith('.npz') for p in paths):
        import numpy as np
        xs, ys = zip(*((z['x'], z['y']) for z in map(np.load, paths)))
        return data.ArrayDataset(np.concatenate(xs), np.concatenate(ys), batch_size, repeat=training, drop_remainder=training)
    raise NotImplementedError('data_path %r: supported sources are .tfrecords, synthetic[:HxW[xC]] and .npz files' % (paths,))


def train(config, save_path, data_path, max_steps, early_stop_steps=None, save_freq=500, validate=False,
          val_data_path=None, visualize=False, profile=False):
    config = load.load_config(config)
    dump.dump_options(os.path.join(save_path, 'options.yaml'), avoid_overwrite=True, config=config, save_path=save_path,
                      data_path=data_path)
    options = config.get('data_options', {})
    ds = make_dataset(data_path, options.get('train', {}), training=True)
    if validate:
        assert val_data_path is not None
        val_ds = make_dataset(val_data_path, options.get('eval', {}), training=False)
    else:
        val_ds = None
    model = engine.TFKerasModel(config)
    results = model.train(ds, save_path=os.path.join(save_path), max_steps=max_steps, early_stop_steps=early_stop_steps,
                          save_freq=save_freq, val_data=val_ds, visualization={} if not visualize else {'train': None},
                          profile=profile)
    if model.ctx.rank == 0:
        dump.dump_train_results(os.path.join(save_path, 'results.pkl'), results, format_='pickle')
    return results
