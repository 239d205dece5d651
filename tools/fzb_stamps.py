"""Phase timing of the block-fused backward kernels from s_memtime stamps (tuning build only: DNNCA_TUNING=1 python -m
dnncancerannotator_amd.build --force).    python tools/fzb_stamps.py <up2|up1|down2|down1>"""
import os, sys, ctypes as C
if sys.argv[1] != 'all':
    os.environ['DNNCA_FZB_ONLY'] = sys.argv[1]
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
f = m.lib.dnnca_debug_fzb_stamps
f.restype = C.c_int; f.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
for _ in range(4):
    m.train_step_dev(xb, yb, 8, 1e-3, cfg)
m.sync()
n = 4 * 1024 * 2 * 32
buf = (C.c_ulonglong * n)()
assert f(buf, n) == 0
allk = np.frombuffer(buf, dtype=np.uint64).reshape(4, 1024, 2, 32).astype(np.int64)
for kid, kname in enumerate(('up1', 'up2', 'down1', 'down2')):
  a = allk[kid]
  if not (a[:, 0, 0] > 0).any():
    continue
  print('----', kname)
  nb = int((a[:, 0, 0] > 0).sum())
  a = a[:nb]
  t0 = a[:, :, 0][a[:, :, 0] > 0].min()
  print(sys.argv[1], 'blocks', nb, 'kernel span %d ticks; block start spread %d' % (a[:, :, 31].max() - t0, a[:, 0, 0].max() - t0))
  names = {1: 'prologue(zero,bm)', 2: 'commit0+bar', 3: '-', 4: 'P1', 5: 'bar', 6: 'P2(+ring)', 7: 'bar', 8: 'P3', 9: 'bar', 10: 'commit+bar'}
  for role, rn in ((0, 'dgrad'), (1, 'wgrad')):
      r = a[:, role, :]
      line = ['%s: prologue %d (issue %d zero %d sync %d stores %d; min %d max %d) commit0 %d' % (
          rn, np.median(r[:, 1] - r[:, 0]), np.median(r[:, 27] - r[:, 0]), np.median(r[:, 28] - r[:, 27]), np.median(r[:, 29] - r[:, 28]),
          np.median(r[:, 1] - r[:, 29]), (r[:, 1] - r[:, 0]).min(), (r[:, 1] - r[:, 0]).max(), np.median(r[:, 2] - r[:, 1]))]
      for it in range(3):
          base = 3 + 8 * it
          if not (r[:, base] > 0).all():
              break
          prev = r[:, 2] if it == 0 else r[:, base - 1]
          line.append('| tile%d gap %d P1 %d bar %d P2+epilogue %d' % (it, np.median(r[:, base] - prev), np.median(r[:, base + 1] - r[:, base]),
                      np.median(r[:, base + 2] - r[:, base + 1]), np.median(r[:, base + 5] - r[:, base + 2])))
          if (r[:, base + 7] > 0).all():
              line.append('bar %d commit %d' % (np.median(r[:, base + 6] - r[:, base + 5]), np.median(r[:, base + 7] - r[:, base + 6])))
      if (r[:, 25] > 0).all():
          line.append('| DBG8: entry->issued %d, tile loads landed after %d more, then to stamp27 (operand loads issued+waited) %d' % (np.median(r[:, 25] - r[:, 0]), np.median(r[:, 26] - r[:, 25]), np.median(r[:, 27] - r[:, 26])))
      line.append('| role end->30: at %d, end 31 at %d (from kernel start, median)' % (np.median(r[:, 30] - t0), np.median(r[:, 31] - t0)))
      print(' '.join(line))
  if '--blocks' in sys.argv:
    pr = (a[:, 0, 27] - a[:, 0, 0])
    st = a[:, 0, 0] - a[:, 0, 0].min()
    print('issue-phase ticks by block: rows = XCD (block & 7), columns = block >> 3')
    for x in range(8):
        print('xcd', x, ' '.join('%5d' % pr[j * 8 + x] for j in range(nb // 8)))
    print('block start (ticks after the first block)')
    for x in range(8):
        print('xcd', x, ' '.join('%5d' % st[j * 8 + x] for j in range(nb // 8)))
