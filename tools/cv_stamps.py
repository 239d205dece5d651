"""Phase timing of ig::k_ig_conv3 (block 0, thread 0) from in-kernel s_memtime stamps -- tuning build only:
    DNNCA_TUNING=1 python -m dnncancerannotator_amd.build && DNNCA_LIB=$PWD/dnncancerannotator_amd/libdnnca_tuning.so python tools/cv_stamps.py [--filters 16] [--bwd]
The stamped launch is the LAST k_ig_conv3 launch before the read-out: forward -> the second conv of the decoder block (filters ->
filters at full resolution); --bwd -> the data gradient of the first encoder block's second conv."""
import argparse, os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dnncancerannotator_amd import device as dev
from dnncancerannotator_amd.synthetic import synthetic_batch
ap = argparse.ArgumentParser()
ap.add_argument('--filters', type=int, default=16)
ap.add_argument('--size', type=int, default=512)
ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--bwd', action='store_true')
a = ap.parse_args()
os.environ['DNNCA_NO_WG_STREAM'] = '1'
dev.init_device(0)
m = dev.DeviceModel('unet', 1, a.size, a.size, a.batch, n_filters_first=a.filters, n_downsample=1, rate=2, kernel_size=3,
                    conv_stride=1, bn=True, padding='same')
m.init_glorot(seed=2)
x, y = synthetic_batch(a.batch, a.size, a.size, 1)
cfg = m.loss_cfg(weight_mul=3.0)
for _ in range(3):
    if a.bwd:
        m.train_step(x, y, 1e-3, cfg)
    else:
        m.forward(x, training=True)
m.sync()
n = 64 * 8
buf = (C.c_ulonglong * n)()
f = m.lib.dnnca_debug_wg_stamps
f.restype = C.c_int; f.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
assert f(buf, n) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(64, 8).astype(np.int64)
nchunks = max(1, a.filters // 16)
nt = min(14, int((t[:, 0] > 0).sum()))
names = ['next_stage (+coef)', 'first fragments', 'steps 0-5 (+staging)', 'steps 6-11 (+staging)', 'barrier']
print('s_memtime ticks; items of block 0 (%d chunks per unit)' % nchunks)
print('%-24s' % 'phase', ' '.join('%6d' % i for i in range(nt)))
for i, nme in enumerate(names):
    print('%-24s' % nme, ' '.join('%6d' % (t[it, i + 1] - t[it, i]) for it in range(nt)))
print('%-24s' % 'epilogue (unit ends)', ' '.join('%6d' % (t[it, 6] - t[it, 5] if t[it, 6] > t[it, 5] else 0) for it in range(nt)))
print('%-24s' % 'item start to start', ' '.join('%6d' % (t[it + 1, 0] - t[it, 0]) for it in range(nt - 1)))
