"""Per-kernel HIP-event table of one train step (development aid; run on the GPU box).

    python tools/profile_step.py [unet|mulmo|unet_big] [--batch B] [--size S] [--generic] [--steps N]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dnncancerannotator_amd import device as dev                # noqa: E402
from dnncancerannotator_amd.synthetic import synthetic_batch    # noqa: E402

CONFIGS = {
    'unet': dict(arch='unet', C=1, B=8, opts=dict(n_filters_first=3, n_downsample=3, bn=False)),
    'unet_big': dict(arch='unet', C=1, B=4, opts=dict(n_filters_first=64, n_downsample=4, bn=True)),
    'mulmo': dict(arch='mulmo', C=3, B=8, opts=dict(n_filters_first=16, n_downsample=4, bn=True)),
}

ap = argparse.ArgumentParser()
ap.add_argument('config', nargs='?', default='unet')
ap.add_argument('--batch', type=int, default=0)
ap.add_argument('--size', type=int, default=512)
ap.add_argument('--steps', type=int, default=5)
ap.add_argument('--generic', action='store_true')
ap.add_argument('--dtype', default='f32')
ap.add_argument('--layers', action='store_true', help='one row per kernel@layer')
a = ap.parse_args()
c = CONFIGS[a.config]
B = a.batch or c['B']
dev.init_device(0)
m = dev.DeviceModel(c['arch'], c['C'], a.size, a.size, B, rate=2, kernel_size=3, conv_stride=1, padding='same',
                    force_generic=a.generic, dtype=a.dtype, **c['opts'])
m.init_glorot(seed=2)
x, y = synthetic_batch(B, a.size, a.size, c['C'])
xb, yb = dev.DeviceBuffer(x), dev.DeviceBuffer(y)
cfg = m.loss_cfg(weight_mul=3.0)
for _ in range(3):
    m.train_step_dev(xb, yb, B, 1e-3, cfg)
m.sync()
m.timer_start()
for _ in range(a.steps):
    m.train_step_dev(xb, yb, B, 1e-3, cfg)
ms = m.timer_stop()
print('%s B=%d %dx%d: %.3f ms/step un-instrumented -> %.1f slices/s' % (a.config, B, a.size, a.size, ms / a.steps, B * a.steps / ms * 1e3))
m.profile_enable(3 if a.layers else 1)
for _ in range(a.steps):
    m.train_step_dev(xb, yb, B, 1e-3, cfg)
m.sync()
rows = sorted(m.profile(), key=lambda r: -r[2])
tot = sum(r[2] for r in rows)
print('%-44s %8s %10s %10s %9s %9s' % ('kernel', 'launches', 'us/launch', 'us/step', 'GB/s', 'GFLOP/s'))
for name, n, tms, by, fl in rows:
    us = tms / n * 1e3
    print('%-44s %8d %10.2f %10.1f %9.1f %9.1f' % (name, n / a.steps, us, tms / a.steps * 1e3, by / us / 1e3, fl / us / 1e3))
print('sum of kernels: %.1f us/step' % (tot / a.steps * 1e3))
print('final loss', m.last_step_out().loss)
