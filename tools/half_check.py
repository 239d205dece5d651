#!/usr/bin/env python3
"""half_check.py -- does storing tensors as bf16 (View::h, ig_plan_half) change any result?  Builds the same bf16 model
twice, once with DNNCA_NO_HALF=1, and compares training-mode logits (no atomics on that path: must be bit-identical),
one step's gradients and the updated weights (float atomics in the weight gradients: identical up to summation order).

    python tools/half_check.py [--size 128] [--batch 2] [--filters 64] [--down 2]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dnncancerannotator_amd import device as dev                      # noqa: E402
from dnncancerannotator_amd.synthetic import synthetic_batch         # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--size', type=int, default=128)
ap.add_argument('--batch', type=int, default=2)
ap.add_argument('--filters', type=int, default=64)
ap.add_argument('--down', type=int, default=2)
a = ap.parse_args()

dev.init_device(0)
x, y = synthetic_batch(a.batch, a.size, a.size, 1)
out = []
for no_half in (False, True, True):
    if no_half:
        os.environ['DNNCA_NO_HALF'] = '1'
    else:
        os.environ.pop('DNNCA_NO_HALF', None)
    m = dev.DeviceModel('unet', 1, a.size, a.size, a.batch, n_filters_first=a.filters, n_downsample=a.down, rate=2,
                        kernel_size=3, conv_stride=1, bn=True, padding='same', dtype='bf16')
    m.init_glorot(seed=2)
    _, logits = m.forward(x, training=True, return_logits=True)
    cfg = m.loss_cfg(weight_mul=3.0)
    m.train_step(x, y, 1e-3, cfg)
    out.append(dict(logits=np.array(logits), grads=np.array(m.get_grads()), params=np.array(m.get_params())))
    m.close()

def rel(u, v):
    return float(np.abs(u - v).max() / (np.abs(v).max() + 1e-30))

h, f, f2 = out
print('logits      half vs f32-stored: max |diff| = %.3g (bit-identical: %s)' % (np.abs(h['logits'] - f['logits']).max(), np.array_equal(h['logits'], f['logits'])))
print('gradients   half vs f32-stored: rel %.3g      f32-stored run to run: rel %.3g' % (rel(h['grads'], f['grads']), rel(f2['grads'], f['grads'])))
print('new weights half vs f32-stored: rel %.3g      f32-stored run to run: rel %.3g' % (rel(h['params'], f['params']), rel(f2['params'], f['params'])))
