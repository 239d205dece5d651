import os, sys, ctypes as C, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dnncancerannotator_amd import device as dev
dev.init_device(0)
m = dev.DeviceModel('unet', 1, 64, 64, 1, n_filters_first=3, n_downsample=3, padding='same')
f = m.lib.dnnca_debug_launch_cost
f.restype = C.c_int; f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float)]
us = C.c_float()
for blocks in (1, 256, 2048):
    for n in (10, 100, 1000):
        t = time.perf_counter(); f(m.handle, n, blocks, C.byref(us)); w = (time.perf_counter() - t) * 1e6 / n
        print('blocks %5d n %5d: %.2f us/launch (gpu events), %.2f us/launch host wall' % (blocks, n, us.value, w))
