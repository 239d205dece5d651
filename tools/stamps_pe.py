"""Prologue / loop / epilogue split of a pixel-group backward kernel from in-kernel stamps (tuning build; DNNCA_STAMPS=C,NS,CO)."""
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
NB = 1024
n = NB * 4 * 8
buf = (C.c_ulonglong * n)()
f = m.lib.dnnca_debug_read_stamps
f.restype = C.c_int; f.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
assert f(m.handle, buf, n) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(NB, 4, 8).astype(np.int64)
live = a[:, 0, 6] > 0
a = a[live]
print('blocks stamped:', len(a))
start, first_top, loop_end, end = a[:, 0, 6], a[:, 0, 0], a[:, 3, 5], a[:, 3, 7]
print('prologue (start -> first tile top)   median %6d  max %6d' % (np.median(first_top - start), (first_top - start).max()))
print('tile loop (first top -> loop end)    median %6d  max %6d' % (np.median(loop_end - first_top), (loop_end - first_top).max()))
print('epilogue (loop end -> end)           median %6d  max %6d' % (np.median(end - loop_end), (end - loop_end).max()))
print('whole block                          median %6d  max %6d' % (np.median(end - start), (end - start).max()))
