"""Soak run (development aid, GPU box): repeated model creation / destruction across architectures and dtypes, a few hundred
training steps each, finite losses that go down, device memory back where it started."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dnncancerannotator_amd import device as dev                # noqa: E402
from dnncancerannotator_amd.synthetic import synthetic_batch    # noqa: E402

dev.init_device(0)
CASES = [
    ('unet', 1, 8, 256, 'f32', dict(n_filters_first=3, n_downsample=3, bn=False), 300),
    ('unet', 1, 2, 128, 'bf16', dict(n_filters_first=64, n_downsample=4, bn=True), 120),
    ('mulmo', 3, 2, 128, 'f32', dict(n_filters_first=16, n_downsample=4, bn=True), 120),
    ('unet', 1, 4, 96, 'f32', dict(n_filters_first=16, n_downsample=2, bn=True), 150),
]
t0 = time.time()
for rep in range(3):
    for arch, C, B, S, dtype, opts, steps in CASES:
        m = dev.DeviceModel(arch, C, S, S, B, rate=2, kernel_size=3, conv_stride=1, padding='same', dtype=dtype, **opts)
        m.init_glorot(seed=rep)
        x, y = synthetic_batch(B, S, S, C, seed_x=rep, seed_y=rep + 10)
        xb, yb = dev.DeviceBuffer(x), dev.DeviceBuffer(y)
        cfg = m.loss_cfg(weight_mul=3.0)
        losses = []
        for s in range(steps):
            out = m.train_step_dev(xb, yb, B, 1e-3, cfg, want_out=(s % 10 == 0 or s == steps - 1))
            if out is not None:
                losses.append(out.loss)
        assert all(np.isfinite(losses)), (arch, dtype, losses)
        assert losses[-1] < losses[0], (arch, dtype, losses[0], losses[-1])
        print('rep %d %-6s %-5s B=%d %3dx%-3d: loss %.4f -> %.4f  (%d steps)' % (rep, arch, dtype, B, S, S, losses[0], losses[-1], steps), flush=True)
        m.close()
        xb.free(); yb.free()
print('soak ok in %.1f s' % (time.time() - t0))
