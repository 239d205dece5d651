#!/usr/bin/env python3
"""What the data-parallel step costs on top of the single-replica step, measured on ONE GPU: with DNNCA_FORCE_RCCL=1 a one-rank
RCCL communicator is created and every step runs the real DP sequence (slab fold, ncclAllReduce of the flat gradient vector +
loss on the step's stream, Adam with 1/world) instead of the fused fold + Adam launch.  A sum over one rank moves no data over
xGMI, so this is the fixed cost of the sequence (launches + RCCL's kernel), the lower bound of the per-step DP overhead."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dnncancerannotator_amd import device as dev                # noqa: E402
from dnncancerannotator_amd.synthetic import synthetic_batch    # noqa: E402

dev.init_device(0)
opts = dict(n_filters_first=3, n_downsample=3, rate=2, kernel_size=3, conv_stride=1, bn=False, padding='same')
x, y = synthetic_batch(8, 512, 512, 1)
xb, yb = dev.DeviceBuffer(x), dev.DeviceBuffer(y)


def run(force, steps=300):
    if force:
        os.environ['DNNCA_FORCE_RCCL'] = '1'
    else:
        os.environ.pop('DNNCA_FORCE_RCCL', None)
    m = dev.DeviceModel('unet', 1, 512, 512, 8, **opts)
    m.init_glorot(seed=2)
    m.comm_init(0, 1, dev.DeviceModel.comm_unique_id() if force else None)
    cfg = m.loss_cfg(weight_mul=3.0)
    for _ in range(30):
        m.train_step_dev(xb, yb, 8, 1e-3, cfg)
    m.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        m.train_step_dev(xb, yb, 8, 1e-3, cfg)
    m.sync()
    dt = (time.perf_counter() - t0) / steps
    m.profile_enable(1)
    for _ in range(5):
        m.train_step_dev(xb, yb, 8, 1e-3, cfg)
    m.sync()
    rows = {r[0]: r[2] / r[1] * 1e3 for r in m.profile() if r[1]}
    m.close()
    return dt, rows


a, ra = run(False)
b, rb = run(True)
print('single replica          : %.4f ms/step' % (a * 1e3))
print('one-rank RCCL communicator: %.4f ms/step (+%.1f us)' % (b * 1e3, (b - a) * 1e6))
print('launches only in one of the two (us per launch, HIP-event bracket):')
for k in sorted(set(ra) ^ set(rb)):
    print('   %-24s %8.2f  %s' % (k, (ra.get(k) or rb.get(k)), 'single' if k in ra else 'dp'))
