"""Phase timing of the fused backward kernel from in-kernel s_memtime stamps (diagnostic build path, DNNCA_STAMPS=C,NS,CO)."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dnncancerannotator_amd import device as dev
from dnncancerannotator_amd.synthetic import synthetic_batch
dev.init_device(0)
m = dev.DeviceModel('unet', 1, 512, 512, 8, n_filters_first=3, n_downsample=3, rate=2, kernel_size=3, conv_stride=1, bn=False, padding='same')
m.init_glorot(seed=2)
x, y = synthetic_batch(8, 512, 512, 1)
xb, yb = dev.DeviceBuffer(x), dev.DeviceBuffer(y)
cfg = m.loss_cfg(weight_mul=3.0)
for _ in range(5):
    m.train_step_dev(xb, yb, 8, 1e-3, cfg)
m.sync()
n = 512 * 4 * 8
buf = (C.c_ulonglong * n)()
f = m.lib.dnnca_debug_read_stamps
f.restype = C.c_int; f.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
assert f(m.handle, buf, n) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(512, 4, 8).astype(np.int64)
names = ['commit+barrier', 'issue next', 'dgrad', 'wgrad', 'end barrier']
d = np.diff(a[:, :, :6], axis=2)
print('phase cycles (median over blocks) per tile iteration 0..3')
for i, nme in enumerate(names):
    print('%-16s' % nme, ' '.join('%8d' % np.median(d[:, t, i]) for t in range(4)))
print('%-16s' % 'iteration', ' '.join('%8d' % np.median(a[:, t, 5] - a[:, t, 0]) for t in range(4)))
print('kernel span (first stamp of any block -> last stamp): %d cycles' % (a[:, :, :6].max() - a[:, :, :6][a[:, :, :6] > 0].min()))
print('block start spread: %d' % (a[:, 0, 0].max() - a[:, 0, 0].min()))
