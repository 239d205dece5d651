"""Host side of the train-time augmentation (annotator/data.py:62-111 `train_ds`, :538-575 option parsing): the random draws.

The reference applies, per image and in the order of `data_options.train.augment_options` (configs/additionals/data_options.yaml:9-13):
    random_crop      crop to `output_size` at the centre offset + clip(int(N(0, stddev)), min_, max_)     (data.py:677-689)
    random_flip      tf.image.random_flip_left_right: flip when U[0,1) < 0.5                              (data.py:620-625)
    random_contrast  tf.image.random_contrast(lower, upper) on the feature channels                       (data.py:586-609)
    random_warp      tfa.image.sparse_image_warp                                                          (data.py:725-763)
All four run on the device (`dnnca_augment_u8`, `dnnca_warp_f32`); this module draws their per-image parameters with a numpy
generator (TensorFlow's own random streams cannot be reproduced without TensorFlow, so the draws are not bit-compatible with a TF
run -- their distributions are the reference's) and solves the small spline system of the warp.
"""

import logging
from collections import namedtuple

import numpy as np

AugmentPlan = namedtuple('AugmentPlan', ['crop', 'flip', 'contrast', 'warp', 'output_size'])
RawBatch = namedtuple('RawBatch', ['raw', 'params', 'output_size', 'label_index', 'warp'])      # warp: (ctrl, wv) or None



def plain_params(n):
    """the draws of a batch that is only converted (evaluation: centre crop, no flip, no contrast)"""
    return [(0, 0, 0, 1.0)] * n


def raw_to_float(batch):
    """Host statement of what the device does with an un-augmented RawBatch (params None): centre crop to output_size, / 255,
    label channel -> y, the other channels -> x (annotator/data.py:195-206,766-788).  For consumers without the device path."""
    if batch.params is not None:
        raise ValueError('raw_to_float converts evaluation batches only (no random draws)')
    raw = np.asarray(batch.raw)
    oh, ow = batch.output_size
    gy, gx = (raw.shape[1] - oh) // 2, (raw.shape[2] - ow) // 2
    c = raw[:, gy:gy + oh, gx:gx + ow, :]
    feat = [i for i in range(raw.shape[-1]) if i != batch.label_index]
    x = np.empty(c.shape[:-1] + (len(feat),), np.float32)
    for j, i in enumerate(feat):
        x[..., j] = c[..., i]
    np.divide(x, np.float32(255.0), out=x)
    y = c[..., batch.label_index].astype(np.float32)
    np.divide(y, np.float32(255.0), out=y)
    return x, y


_KNOWN = ('random_crop', 'random_flip', 'random_contrast', 'random_warp')
_warned = set()


def parse_augment_options(options, output_size):
    """data.py:538-551 + the defaults of train_ds (data.py:87-93).  options None -> {'random_crop': {}} like train_ds."""
    if options is None:
        options = {'random_crop': {}}
    crop, flip, contrast, warp = None, False, None, None
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
        elif name == 'random_warp':
            warp = dict(n_points=100, max_diff=5, stddev=2.0)                  # data.py:725 random_warp defaults
            warp.update({k: v for k, v in conf.items() if k != 'process_in_batch'})
    return AugmentPlan(crop, flip, contrast, warp, tuple(output_size))


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


def draw_warp(rng, n, size, n_points=100, max_diff=5, stddev=2.0):
    """Control points of random_warp (data.py:748-752) for `n` square images of edge `size`: source uniform in [0, size)^2,
    destination = source + clip(N(0, stddev), -max_diff, max_diff).  Returns (source, dest) float32 [n, n_points, 2]."""
    raw = rng.uniform(0.0, float(size), (n, n_points, 2)).astype(np.float32)
    diff = np.clip(rng.normal(0.0, stddev, (n, n_points, 2)), -max_diff, max_diff).astype(np.float32)
    return raw, raw + diff


def _phi2(r):
    """tfa interpolate_spline._phi for order 2 on squared distances: 0.5 r log(max(r, 1e-10))"""
    return 0.5 * r * np.log(np.maximum(r, 1e-10))


def solve_warp(source, dest):
    """The polyharmonic-spline system of tfa.image.sparse_image_warp (order 2, no regularisation, no boundary points):
    train points = dest, train values = dest - source; [[phi(|ci-cj|^2), B], [B^T, 0]] [w; v] = [f; 0] with B = [c, 1].
    Returns (ctrl = dest, wv [n, n_points + 3, 2]) float64 for dnnca_warp_f32."""
    source, dest = np.asarray(source, np.float64), np.asarray(dest, np.float64)
    n, k, _ = dest.shape
    out = np.empty((n, k + 3, 2), np.float64)
    for b in range(n):
        c, f = dest[b], dest[b] - source[b]
        d2 = ((c[:, None, :] - c[None, :, :]) ** 2).sum(-1)
        lhs = np.zeros((k + 3, k + 3))
        lhs[:k, :k] = _phi2(d2)
        lhs[:k, k:k + 2] = c
        lhs[:k, k + 2] = 1.0
        lhs[k:, :k] = lhs[:k, k:].T
        rhs = np.zeros((k + 3, 2))
        rhs[:k] = f
        out[b] = np.linalg.solve(lhs, rhs)
    return dest, out
