#!/usr/bin/env python3
"""host_rate_dense.py [mulmo|unet_big] -- host enqueue time per train step of a dense configuration against the GPU's (is the step
GPU-bound with the encoder streams' extra event traffic?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dnncancerannotator_amd import device as dev
cfgname = sys.argv[1] if len(sys.argv) > 1 else 'mulmo'
dev.init_device(0)
if cfgname == 'mulmo':
    arch, C, B, opts, dtype = 'mulmo', 3, 8, dict(n_filters_first=16, n_downsample=4, bn=True), 'f32'
else:
    arch, C, B, opts, dtype = 'unet', 1, 4, dict(n_filters_first=64, n_downsample=4, bn=True), 'bf16'
m = dev.DeviceModel(arch, C, 512, 512, B, dtype=dtype, rate=2, kernel_size=3, conv_stride=1, padding='same', **opts)
m.init_glorot(seed=2)
rng = np.random.default_rng(0)
x = rng.random((B, 512, 512, C)).astype(np.float32)
y = (rng.random((B, 512, 512)) < 0.05).astype(np.float32)
xb, yb = dev.DeviceBuffer(x), dev.DeviceBuffer(y)
cfg = m.loss_cfg(weight_mul=3.0)
for _ in range(5):
    m.train_step_dev(xb, yb, B, 1e-3, cfg)
m.sync()
N = 20
t0 = time.perf_counter()
for _ in range(N):
    m.train_step_dev(xb, yb, B, 1e-3, cfg)
t1 = time.perf_counter()
m.sync()
t2 = time.perf_counter()
print('%s B=%d: host enqueue %.1f us/step, enqueue + drain %.1f us/step' % (cfgname, B, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6))
