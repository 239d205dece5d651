"""Phase timing of the block-fused forward kernels from s_memtime stamps (tuning build only).
   python tools/fz_stamps.py <level 0|1|2> <down|up>"""
import os, sys, ctypes as C
os.environ['DNNCA_FZ_ONLY'] = '%s%d' % (sys.argv[2], int(sys.argv[1]))
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dnncancerannotator_amd import device as dev
from dnncancerannotator_amd.synthetic import synthetic_batch
level, kind = int(sys.argv[1]), sys.argv[2]
dev.init_device(0)
m = dev.DeviceModel('unet', 1, 512, 512, 8, n_filters_first=3, n_downsample=3, rate=2, kernel_size=3, conv_stride=1, bn=False, padding='same')
m.init_glorot(seed=2)
x, y = synthetic_batch(8, 512, 512, 1)
# order of fused launches in a forward pass: down0 down1 down2 up0(level 2) up1 up2; the stamps keep the LAST launch that wrote them,
# so stop the network after the wanted block by running a smaller network whose last fused block is the wanted one
want = '%s%d' % (kind, level)
f = m.lib.dnnca_debug_fz_stamps
f.restype = C.c_int; f.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
for _ in range(3):
    m.forward(x, training=True)
n = 1024 * 2 * 8
buf = (C.c_ulonglong * n)()
assert f(buf, n) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 2, 8).astype(np.int64)
nb = int((a[:, 0, 0] > 0).sum())
a = a[:nb]
names = ['conv1', 'conv2', 'commit+issue', 'stores+barrier'] if kind == 'down' else ['tconv', 'conv0', 'commit+issue+up st', 'conv1(+y0 st)', 'store y1+barrier']
print(want, 'blocks', nb)
for t in range(2):
    ok = a[:, t, 0] > 0
    if not ok.any():
        continue
    d = np.diff(a[ok][:, t, :len(names) + 1], axis=1)
    print('tile', t, ' '.join('%s %d' % (nm, np.median(d[:, i])) for i, nm in enumerate(names)), ' total', int(np.median(a[ok][:, t, len(names)] - a[ok][:, t, 0])))
print('kernel span %d ticks; block start spread %d' % (a[a > 0].max() - a[:, 0, 0].min(), a[:, 0, 0].max() - a[:, 0, 0].min()))
