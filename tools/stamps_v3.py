"""Phase timing of k_bwd3v from in-kernel s_memtime stamps (tuning build: DNNCA_TUNING=1 python -m dnncancerannotator_amd.build;
run with DNNCA_STAMPS=3,1,3).  The last stamped launch of the step is the pool-fold variant (encoder block 0)."""
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
NB = 512
n = NB * 4 * 8
buf = (C.c_ulonglong * n)()
f = m.lib.dnnca_debug_read_stamps
f.restype = C.c_int; f.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
assert f(m.handle, buf, n) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(NB, 4, 8).astype(np.int64)
t0 = a[:, 0, 6].min()
print('block start spread (stamp 6): %d cycles' % (a[:, 0, 6].max() - t0))
names = ['commit+barrier', 'issue next', 'compute rows', 'end barrier']
d = np.diff(a[:, :, :5], axis=2)
print('phase cycles (median over blocks) per tile iteration 0..3')
for i, nme in enumerate(names):
    print('%-16s' % nme, ' '.join('%8d' % np.median(d[:, t, i]) for t in range(4)))
print('%-16s' % 'tile start - t0', ' '.join('%8d' % np.median(a[:, t, 0] - t0) for t in range(4)))
print('start -> first tile top (median): %d' % np.median(a[:, 0, 0] - a[:, 0, 6]))
print('loop end - t0 (median / max): %d / %d' % (np.median(a[:, 3, 5] - t0), (a[:, 3, 5] - t0).max()))
print('kernel end - t0 (median / max): %d / %d' % (np.median(a[:, 3, 7] - t0), (a[:, 3, 7] - t0).max()))
