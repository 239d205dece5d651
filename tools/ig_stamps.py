"""Phase timing of igb::k_igb_conv3 (block 0, thread 0) from in-kernel s_memtime stamps -- tuning build only:
    DNNCA_TUNING=1 python -m dnncancerannotator_amd.build && python tools/ig_stamps.py [--filters 256] [--size 64] [--batch 4]
The stamped launch is the last conv of a forward pass (decoder, filters -> filters at full resolution)."""
import argparse, os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dnncancerannotator_amd import device as dev
from dnncancerannotator_amd.synthetic import synthetic_batch
ap = argparse.ArgumentParser()
ap.add_argument('--filters', type=int, default=256)
ap.add_argument('--size', type=int, default=64)
ap.add_argument('--batch', type=int, default=4)
a = ap.parse_args()
dev.init_device(0)
m = dev.DeviceModel('unet', 1, a.size, a.size, a.batch, n_filters_first=a.filters, n_downsample=1, rate=2, kernel_size=3,
                    conv_stride=1, bn=True, padding='same', dtype='bf16')
m.init_glorot(seed=2)
x, y = synthetic_batch(a.batch, a.size, a.size, 1)
for _ in range(3):
    m.forward(x, training=True)
m.sync()
n = 64 * 8
buf = (C.c_ulonglong * n)()
f = m.lib.dnnca_debug_ig_stamps
f.restype = C.c_int; f.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
assert f(buf, n) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(64, 8).astype(np.int64)
nchunks = a.filters // 32
names = ['stage_of', 'first fragments issued', 'taps 0-2 (+staging)', 'taps 3-5 (+staging)', 'taps 6-8 (+staging)', 'barrier']
print('s_memtime ticks (100 MHz: 1 tick = 10 ns = 24 cycles at 2.4 GHz); items of unit 0 (%d chunks)' % nchunks)
print('%-26s' % 'phase', ' '.join('%6d' % i for i in range(min(nchunks, 12))))
for i, nme in enumerate(names):
    print('%-26s' % nme, ' '.join('%6d' % (t[it, i + 1] - t[it, i]) for it in range(min(nchunks, 12))))
print('%-26s' % 'item total', ' '.join('%6d' % (t[it, 6] - t[it, 0]) for it in range(min(nchunks, 12))))
print('%-26s' % 'gap to next item', ' '.join('%6d' % (t[it + 1, 0] - t[it, 6]) for it in range(min(nchunks, 12) - 1)))
print('epilogue of unit 0: %d ticks' % (t[nchunks - 1, 7] - t[nchunks - 1, 6]))
e = t[48]
print('inside the epilogue (first unit; ticks): to row 0 %d, rows %s, statistics sums + LDS %d, barrier %d; then the bucket adds' % (
    e[1] - e[0], ' '.join(str(int(e[2 + r] - e[1 + r])) for r in range(4)), e[6] - e[5], e[7] - e[6]))
