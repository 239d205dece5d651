import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dnncancerannotator_amd import device as dev
dev.init_device(0)
m = dev.DeviceModel('unet', 1, 64, 64, 1, n_filters_first=3, n_downsample=3, padding='same')
f = m.lib.dnnca_debug_mfma_rate
f.restype = C.c_int; f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
t = C.c_float()
for blocks in (256, 512):
    for mode, nch in ((0, 1), (0, 2), (0, 4), (1, 2), (1, 4), (2, 4)):
        f(m.handle, mode, nch, blocks, 2000, C.byref(t))
        print('blocks %4d (x8 waves) mode %d chains %d: %.1f TFLOP/s (f32 MFMA 16x16x4; peak 157)' % (blocks, mode, nch, t.value))
