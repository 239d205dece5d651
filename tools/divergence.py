"""Diagnostic (GPU box): two models from identical weights take the same train steps in lockstep; prints how far their weights and
losses drift apart per step.  Float atomics make single steps differ by ~1e-7; anything that jumps is a race."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dnncancerannotator_amd import device as dev                # noqa: E402
from dnncancerannotator_amd.synthetic import synthetic_batch    # noqa: E402

dev.init_device(0)
opts = dict(n_filters_first=3, n_downsample=3, rate=2, kernel_size=3, conv_stride=1, bn=False, padding='same')
B, S = 8, 512
K = int(sys.argv[1]) if len(sys.argv) > 1 else 60
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-3
x, y = synthetic_batch(B, S, S, 1)
xb, yb = dev.DeviceBuffer(x), dev.DeviceBuffer(y)
ms = []
for rep in range(2):
    m = dev.DeviceModel('unet', 1, S, S, B, **opts)
    m.init_glorot(seed=2)
    ms.append(m)
cfg = ms[0].loss_cfg(weight_mul=3.0)
for s in range(K):
    outs = [m.train_step_dev(xb, yb, B, lr, cfg, want_out=True) for m in ms]
    g = [m.get_grads() for m in ms]
    p = [m.get_params() for m in ms]
    dg = np.abs(g[0] - g[1]).max() / (np.abs(g[0]).max() + 1e-30)
    dp = np.abs(p[0] - p[1]).max()
    if s < 10 or s % 10 == 0 or s == K - 1:
        print('step %3d: loss %.6f %.6f  |dgrad|/max %.2e  max|dparam| %.2e' % (s, outs[0].loss, outs[1].loss, dg, dp), flush=True)
