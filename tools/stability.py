"""Stability run of the configs/unet.yaml train step at the BASELINE size (GPU box): N independent models from the same weights take
the same K steps; every trajectory must stay finite and go down.  The trajectories themselves drift apart -- the weight-gradient
slabs are float atomics (1e-8 differences per step) and Adam turns the sign noise of near-zero gradients into lr-sized updates, so
two runs differ by 1e-5 in the weights after ten steps and visibly in the loss after a hundred (tools/divergence.py prints the
growth step by step; the per-layer kernels of round 1 behave the same) -- which is why parity is judged per step."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dnncancerannotator_amd import device as dev                # noqa: E402
from dnncancerannotator_amd.synthetic import synthetic_batch    # noqa: E402

dev.init_device(0)
opts = dict(n_filters_first=3, n_downsample=3, rate=2, kernel_size=3, conv_stride=1, bn=False, padding='same')
B, S, K, N = 8, 512, int(sys.argv[1]) if len(sys.argv) > 1 else 400, 4
x, y = synthetic_batch(B, S, S, 1)
xb, yb = dev.DeviceBuffer(x), dev.DeviceBuffer(y)
traj = []
for rep in range(N):
    m = dev.DeviceModel('unet', 1, S, S, B, **opts)
    m.init_glorot(seed=2)
    cfg = m.loss_cfg(weight_mul=3.0)
    losses = []
    for s in range(K):
        out = m.train_step_dev(xb, yb, B, 1e-3, cfg, want_out=(s % 20 == 0 or s == K - 1))
        if out is not None:
            losses.append(out.loss)
    p = m.get_params()
    assert np.all(np.isfinite(losses)) and np.all(np.isfinite(p)), rep
    traj.append((np.array(losses), p))
    m.close()
    print('run %d: loss %.6f -> %.6f' % (rep, losses[0], losses[-1]), flush=True)
ref_l, ref_p = traj[0]
for l, p in traj[1:]:
    dl = np.abs(l - ref_l).max() / max(1.0, np.abs(ref_l).max())
    dp = np.abs(p - ref_p).max()
    print('  vs run 0: loss trajectory diff %.2e, weights diff %.2e' % (dl, dp))
for l, p in traj:
    assert l[-1] < l[0]
print('stability ok')
