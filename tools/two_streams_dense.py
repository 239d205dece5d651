"""Feasibility probe for the dense configurations: N independent models of batch B/N (each with its own HIP streams) stepped
concurrently against one model of batch B -- how much of the alternation of matrix-core-bound convs and HBM-bound BatchNorm
passes does stream-level concurrency hide?  (an upper bound for running mulmo_unet's three encoders on three streams)

    python tools/two_streams_dense.py [mulmo|unet_big]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dnncancerannotator_amd import device as dev                # noqa: E402
from dnncancerannotator_amd.synthetic import synthetic_batch    # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else 'mulmo'
dev.init_device(0)
if which == 'mulmo':
    arch, C, B, dtype = 'mulmo', 3, 8, 'f32'
    opts = dict(n_filters_first=16, n_downsample=4, rate=2, kernel_size=3, conv_stride=1, bn=True, padding='same')
else:
    arch, C, B, dtype = 'unet', 1, 4, 'bf16'
    opts = dict(n_filters_first=64, n_downsample=4, rate=2, kernel_size=3, conv_stride=1, bn=True, padding='same')


def make(b):
    m = dev.DeviceModel(arch, C, 512, 512, b, dtype=dtype, **opts)
    m.init_glorot(seed=2)
    x, y = synthetic_batch(b, 512, 512, C)
    return m, dev.DeviceBuffer(x), dev.DeviceBuffer(y), m.loss_cfg(weight_mul=3.0)


def run(models, steps=20):
    for _ in range(3):
        for m, xb, yb, cfg in models:
            m.train_step_dev(xb, yb, xb.shape[0], 1e-3, cfg)
    for m, *_ in models:
        m.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        for m, xb, yb, cfg in models:
            m.train_step_dev(xb, yb, xb.shape[0], 1e-3, cfg)
    for m, *_ in models:
        m.sync()
    return (time.perf_counter() - t0) / steps


t = run([make(B)])
print('%s: one model  B=%d        : %.3f ms/step -> %.0f slices/s' % (which, B, t * 1e3, B / t), flush=True)
t = run([make(B // 2), make(B // 2)])
print('%s: two models B=%d + B=%d  : %.3f ms per pair of steps -> %.0f slices/s' % (which, B // 2, B // 2, t * 1e3, B / t), flush=True)
t1 = run([make(B // 2)])
print('%s: one model  B=%d        : %.3f ms/step (x2 = %.3f)' % (which, B // 2, t1 * 1e3, 2e3 * t1), flush=True)
