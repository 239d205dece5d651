"""Phase timing of ig3x::k_ig3x_conv3 (block 0, thread 0) from in-kernel s_memtime stamps -- tuning build only:
    DNNCA_TUNING=1 python -m dnncancerannotator_amd.build
    DNNCA_LIB=$PWD/dnncancerannotator_amd/libdnnca_tuning.so python tools/x3_stamps.py [--filters 16] [--size 512] [--batch 8]
The stamped launch is the last conv of a forward pass (decoder, filters -> filters at full resolution)."""
import argparse, os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dnncancerannotator_amd import device as dev
from dnncancerannotator_amd.synthetic import synthetic_batch
ap = argparse.ArgumentParser()
ap.add_argument('--filters', type=int, default=16)
ap.add_argument('--size', type=int, default=512)
ap.add_argument('--batch', type=int, default=8)
a = ap.parse_args()
dev.init_device(0)
m = dev.DeviceModel('unet', 1, a.size, a.size, a.batch, n_filters_first=a.filters, n_downsample=1, rate=2, kernel_size=3,
                    conv_stride=1, bn=True, padding='same')
m.init_glorot(seed=2)
x, y = synthetic_batch(a.batch, a.size, a.size, 1)
for _ in range(3):
    m.forward(x, training=True)
m.sync()
n = 64 * 8
buf = (C.c_ulonglong * n)()
f = m.lib.dnnca_debug_x3_stamps
f.restype = C.c_int; f.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
assert f(buf, n) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(64, 8).astype(np.int64)
nchunks = a.filters // 16
nshow = 12
names = ['commit (wait loads, split, LDS writes)', 'barrier 1', 'issue next item', 'MFMA phase', 'barrier 2']
print('s_memtime ticks (shader cycles); first %d items of block 0 (%d chunks per unit)' % (nshow, nchunks))
print('%-40s' % 'phase', ' '.join('%6d' % i for i in range(nshow)))
for i, nme in enumerate(names):
    print('%-40s' % nme, ' '.join('%6d' % (t[it, i + 1] - t[it, i]) for it in range(nshow)))
print('%-40s' % 'item total', ' '.join('%6d' % (t[it, 5] - t[it, 0]) for it in range(nshow)))
print('%-40s' % 'gap to next item (epilogue at unit end)', ' '.join('%6d' % (t[it + 1, 0] - t[it, 5]) for it in range(nshow)))
