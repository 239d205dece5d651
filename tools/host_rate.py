#!/usr/bin/env python3
"""host_rate.py -- is the unet.yaml step GPU-bound?  Enqueues N train steps without synchronising and reports the host's
time per step (ctypes call + 38 kernel launches) next to the GPU's."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dnncancerannotator_amd import device as dev
from dnncancerannotator_amd.synthetic import synthetic_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev.init_device(0)
m = dev.DeviceModel('unet', 1, 512, 512, B, n_filters_first=3, n_downsample=3, rate=2, kernel_size=3, conv_stride=1, bn=False, padding='same')
m.init_glorot(seed=2)
x, y = synthetic_batch(B, 512, 512, 1)
xb, yb = dev.DeviceBuffer(x), dev.DeviceBuffer(y)
cfg = m.loss_cfg(weight_mul=3.0)
for _ in range(20):
    m.train_step_dev(xb, yb, B, 1e-3, cfg)
m.sync()
N = 300
t0 = time.perf_counter()
for _ in range(N):
    m.train_step_dev(xb, yb, B, 1e-3, cfg)
t1 = time.perf_counter()
m.sync()
t2 = time.perf_counter()
print('B=%d: host enqueue %.1f us/step, enqueue + drain %.1f us/step' % (B, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6))
