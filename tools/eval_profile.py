"""Per-kernel HIP-event table of one evaluation step (forward with moving statistics + loss + pixel confusion counts)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dnncancerannotator_amd import device as dev                # noqa: E402
from dnncancerannotator_amd.synthetic import synthetic_batch    # noqa: E402

dev.init_device(0)
B, S = 8, 512
m = dev.DeviceModel('unet', 1, S, S, B, rate=2, kernel_size=3, conv_stride=1, padding='same', n_filters_first=3, n_downsample=3, bn=False)
m.init_glorot(seed=2)
x, y = synthetic_batch(B, S, S, 1)
cfg = m.loss_cfg(weight_mul=3.0)
thr = np.linspace(0.0, 1.0, 50).astype(np.float32)
for _ in range(3):
    m.eval_step(x, y, cfg)
    m.pixel_confusion(y, thr)
m.sync()
t0 = time.perf_counter()
for _ in range(20):
    m.eval_step(x, y, cfg)
    m.pixel_confusion(y, thr)
m.sync()
print('eval step incl. host copies: %.3f ms' % ((time.perf_counter() - t0) / 20 * 1e3))
m.profile_enable(1)
for _ in range(5):
    m.eval_step(x, y, cfg)
    m.pixel_confusion(y, thr)
m.sync()
for name, n, tms, by, fl in sorted(m.profile(), key=lambda r: -r[2])[:12]:
    print('%-28s %6d %10.2f us/launch' % (name, n / 5, tms / n * 1e3))
