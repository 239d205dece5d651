"""Phase timing of ig::k_ig_wgrad2 (block 0, thread 0) from in-kernel s_memtime stamps -- tuning build only:
    DNNCA_TUNING=1 python -m dnncancerannotator_amd.build && DNNCA_LIB=$PWD/dnncancerannotator_amd/libdnnca_tuning.so python tools/wg_stamps.py [--filters 16]
The stamped launch is the last ig_wgrad2 launch of a backward pass: the second conv of the first encoder block (filters -> filters
at full resolution, fp32)."""
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
os.environ['DNNCA_NO_WG_STREAM'] = '1'
dev.init_device(0)
m = dev.DeviceModel('unet', 1, a.size, a.size, a.batch, n_filters_first=a.filters, n_downsample=1, rate=2, kernel_size=3,
                    conv_stride=1, bn=True, padding='same')
m.init_glorot(seed=2)
x, y = synthetic_batch(a.batch, a.size, a.size, 1)
cfg = m.loss_cfg(weight_mul=3.0)
for _ in range(3):
    m.train_step(x, y, 1e-3, cfg)
m.sync()
n = 64 * 8
buf = (C.c_ulonglong * n)()
f = m.lib.dnnca_debug_wg_stamps
f.restype = C.c_int; f.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
assert f(buf, n) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(64, 8).astype(np.int64)
nt = int((t[:, 0] > 0).sum())
names = ['barrier 1', 'commit', 'issue', 'barrier 2', 'LDS reads + MFMAs']
print('s_memtime ticks; %d tiles of block 0' % nt)
print('%-20s' % 'phase', ' '.join('%6d' % i for i in range(min(nt, 14))))
for i, nme in enumerate(names):
    print('%-20s' % nme, ' '.join('%6d' % (t[it, i + 1] - t[it, i]) for it in range(min(nt, 14))))
print('%-20s' % 'tile total', ' '.join('%6d' % (t[it, 5] - t[it, 0]) for it in range(min(nt, 14))))
print('plan:', [r[0] for r in m.plan() if 'wgrad' in r[0]])
