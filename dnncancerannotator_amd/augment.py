"""Host side of the train-time augmentation (annotator/data.py:62-111 `train_ds`, :538-575 option parsing): the random draws.

The reference applies, per image and in the order of `data_options.train.augment_options` (configs/additionals/data_options.yaml:9-13):
    random_crop      crop to `output_size` at the centre offset + clip(int(N(0, stddev)), min_, max_)     (data.py:677-689)
    random_flip      tf.image.random_flip_left_right: flip when U[0,1) < 0.5                              (data.py:620-625)
    random_contrast  tf.image.random_contrast(lower, upper) on the feature channels                       (data.py:586-609)
    random_warp      tfa.image.sparse_image_warp                                                          (data.py:725-763)
The first three run on the device (`dnnca_augment_u8`); this module draws their per-image parameters with a numpy generator
(TensorFlow's own random streams cannot be reproduced without TensorFlow, so the draws are not bit-compatible with a TF run --
their distributions are the reference's).  random_warp is not part of the accelerated path: it is skipped with a warning.
"""

import logging
from collections import namedtuple

import numpy as np

AugmentPlan = namedtuple('AugmentPlan', ['crop', 'flip', 'contrast', 'output_size'])
RawBatch = namedtuple('RawBatch', ['raw', 'params', 'output_size', 'label_index'])

_KNOWN = ('random_crop', 'random_flip', 'random_contrast', 'random_warp')
_warned = set()


def parse_augment_options(options, output_size):
    """data.py:538-551 + the defaults of train_ds (data.py:87-93).  options None -> {'random_crop': {}} like train_ds."""
    if options is None:
        options = {'random_crop': {}}
    crop, flip, contrast = None, False, None
    for name, conf in options.items():
        conf = dict(conf or {})
        if name not in _KNOWN:
            raise KeyError('unknown augment option %r (the reference looks up augment_%s, data.py:543)' % (name, name))
        if name == 'random_crop':
            crop = dict(stddev=4, max_=6, min_=-6)
            crop.update({k: v for k, v in conf.items() if k != 'output_size'})
            if 'output_size' in conf:
                output_size = tuple(conf['output_size'])
        elif name == 'random_flip':
            flip = True
        elif name == 'random_contrast':
            contrast = dict(lower=0.8, upper=1.2)
            contrast.update({k: v for k, v in conf.items() if k != 'target_channels'})
            contrast['target_channels'] = conf.get('target_channels')          # None: every feature channel (data.py:91)
        elif name == 'random_warp' and name not in _warned:
            _warned.add(name)
            logging.warning('augment option random_warp (tfa sparse_image_warp) is outside the accelerated path: skipped')
    return AugmentPlan(crop, flip, contrast, tuple(output_size))


def draw_params(rng, n, plan):
    """Per-image draws [(dy, dx, flip, contrast)] for `n` images."""
    out = []
    for _ in range(n):
        dy = dx = 0
        if plan.crop is not None:
            d = rng.normal(0.0, plan.crop['stddev'], 2)
            d = np.clip(np.trunc(d).astype(np.int64), plan.crop['min_'], plan.crop['max_'])      # tf.cast(float -> int32) truncates
            dy, dx = int(d[0]), int(d[1])
        flip = int(plan.flip and rng.random() < 0.5)
        contrast = float(rng.uniform(plan.contrast['lower'], plan.contrast['upper'])) if plan.contrast is not None else 1.0
        out.append((dy, dx, flip, contrast))
    return out
