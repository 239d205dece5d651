"""Feasibility probe: two independent half-batch models (each on its own HIP stream) stepped concurrently against one
full-batch model: how much of the per-kernel latency chain does stream-level concurrency hide?"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dnncancerannotator_amd import device as dev                # noqa: E402
from dnncancerannotator_amd.synthetic import synthetic_batch    # noqa: E402

dev.init_device(0)
opts = dict(n_filters_first=3, n_downsample=3, rate=2, kernel_size=3, conv_stride=1, bn=False, padding='same')


def make(B):
    m = dev.DeviceModel('unet', 1, 512, 512, B, **opts)
    m.init_glorot(seed=2)
    x, y = synthetic_batch(B, 512, 512, 1)
    return m, dev.DeviceBuffer(x), dev.DeviceBuffer(y), m.loss_cfg(weight_mul=3.0)


def run(models, steps=200):
    for _ in range(20):
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


full = [make(8)]
t = run(full)
print('one model  B=8          : %.3f ms/step -> %.0f slices/s' % (t * 1e3, 8 / t))
halves = [make(4), make(4)]
t = run(halves)
print('two models B=4 + B=4    : %.3f ms per pair of steps -> %.0f slices/s' % (t * 1e3, 8 / t))
quarters = [make(2) for _ in range(4)]
t = run(quarters)
print('four models B=2 x 4     : %.3f ms per round -> %.0f slices/s' % (t * 1e3, 8 / t))
