#!/usr/bin/env python3
"""fullsize_train_check.py [steps] -- the two dense BASELINE configurations at full size (unet_big bf16 4 x 512 x 512, mulmo_unet
fp32 8 x 512 x 512 x 3): `steps` Adam steps on one synthetic batch; the loss must stay finite and go down (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dnncancerannotator_amd import device as dev
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
dev.init_device(0)
for name, arch, C, B, opts, dtype in (('unet_big', 'unet', 1, 4, dict(n_filters_first=64, n_downsample=4, bn=True), 'bf16'),
                                      ('mulmo_unet', 'mulmo', 3, 8, dict(n_filters_first=16, n_downsample=4, bn=True), 'f32')):
    m = dev.DeviceModel(arch, C, 512, 512, B, dtype=dtype, rate=2, kernel_size=3, conv_stride=1, padding='same', **opts)
    m.init_glorot(seed=3)
    rng = np.random.default_rng(1)
    x = rng.random((B, 512, 512, C)).astype(np.float32)
    yy, xx = np.mgrid[0:512, 0:512]
    y = np.stack([((yy - 200 - 20 * b) ** 2 + (xx - 260) ** 2 < (30 + 5 * b) ** 2) for b in range(B)]).astype(np.float32)
    x[..., 0] += 0.5 * y
    xb, yb = dev.DeviceBuffer(x), dev.DeviceBuffer(y)
    cfg = m.loss_cfg(weight_mul=3.0)
    losses = []
    for s in range(steps):
        out = m.train_step_dev(xb, yb, B, 1e-3, cfg, want_out=(s % 10 == 0 or s == steps - 1))
        if out is not None:
            losses.append(out.loss)
    ok = all(np.isfinite(losses)) and losses[-1] < 0.5 * losses[0]
    print('%-10s %s: loss %s -> %.4f over %d steps: %s' % (name, dtype, ' '.join('%.3f' % l for l in losses[:4]), losses[-1], steps, 'ok' if ok else 'FAILED'), flush=True)
    m.close()
    if not ok:
        sys.exit(1)
