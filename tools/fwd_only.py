"""Runs forward passes only (for rocprofv3 counter collection on the forward kernels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dnncancerannotator_amd import device as dev
from dnncancerannotator_amd.synthetic import synthetic_batch
import ctypes as C
from dnncancerannotator_amd._lib import check
mode = sys.argv[1] if len(sys.argv) > 1 else 'fwd'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev.init_device(0)
m = dev.DeviceModel('unet', 1, 512, 512, 8, n_filters_first=3, n_downsample=3, rate=2, kernel_size=3, conv_stride=1, bn=False, padding='same')
m.init_glorot(seed=2)
x, y = synthetic_batch(8, 512, 512, 1)
xb, yb = dev.DeviceBuffer(x), dev.DeviceBuffer(y)
cfg = m.loss_cfg(weight_mul=3.0)
for _ in range(n):
    if mode == 'fwd':
        check(m.lib.dnnca_forward_dev(m.handle, xb.ptr, 8, 1))
    else:
        m.train_step_dev(xb, yb, 8, 1e-3, cfg)
m.sync()
print('done')
