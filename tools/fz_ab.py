"""A/B of the block-fused forward kernels: un-instrumented train-step and inference-forward time per variant (GPU box).
Each variant runs in a child process (the library reads DNNCA_NO_FUSED / DNNCA_FZ_ONLY once)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
sys.path.insert(0, %r)
from dnncancerannotator_amd import device as dev
from dnncancerannotator_amd.synthetic import synthetic_batch
dev.init_device(0)
B = 8
m = dev.DeviceModel('unet', 1, 512, 512, B, n_filters_first=3, n_downsample=3, rate=2, kernel_size=3, conv_stride=1, bn=False, padding='same')
m.init_glorot(seed=2)
x, y = synthetic_batch(B, 512, 512, 1)
xb, yb = dev.DeviceBuffer(x), dev.DeviceBuffer(y)
cfg = m.loss_cfg(weight_mul=3.0)
for _ in range(10): m.train_step_dev(xb, yb, B, 1e-3, cfg)
m.sync(); best = 1e9
for rep in range(3):
    m.timer_start()
    for _ in range(50): m.train_step_dev(xb, yb, B, 1e-3, cfg)
    best = min(best, m.timer_stop() / 50)
lib = m.lib
for _ in range(10): lib.dnnca_forward_dev(m.handle, xb.ptr, B, 0)
m.sync(); bf = 1e9
for rep in range(3):
    m.timer_start()
    for _ in range(50): lib.dnnca_forward_dev(m.handle, xb.ptr, B, 0)
    bf = min(bf, m.timer_stop() / 50)
print('RESULT %%.4f %%.4f' %% (best, bf))
''' % ROOT
variants = [('unfused', {'DNNCA_NO_FUSED': '1'})] + [(v, {'DNNCA_FZ_ONLY': v}) for v in ('down0', 'down1', 'down2', 'up0', 'up1', 'up2')] + [('all fused', {})]
for name, env in variants:
    e = dict(os.environ)
    e.pop('DNNCA_NO_FUSED', None); e.pop('DNNCA_FZ_ONLY', None)
    e.update(env)
    r = subprocess.run([sys.executable, '-c', CHILD], env=e, capture_output=True, text=True, timeout=300)
    line = [l for l in r.stdout.splitlines() if l.startswith('RESULT')]
    print('%-10s train step %s ms   inference forward %s ms' % ((name,) + tuple(line[0].split()[1:3])) if line else (name, r.stderr[-500:]), flush=True)
