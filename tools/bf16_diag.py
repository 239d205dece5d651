import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dnncancerannotator_amd import device as dev
from oracle import unet_oracle as O
dev.init_device(0)
B, S = int(sys.argv[1]) if len(sys.argv) > 1 else 1, int(sys.argv[2]) if len(sys.argv) > 2 else 64
full = dict(rate=2, kernel_size=3, conv_stride=1, padding='same', n_filters_first=64, n_downsample=4, bn=True)
spec = O.ModelSpec('unet', 1, **full)
params = O.init_params(spec, seed=2)
x, y = O.synthetic_batch(B, S, S, 1)
cfg = dict(weight_mul=3.0)
p64 = {n: v.astype(np.float64) for n, v in params.items()}
loss, grads, _, _ = O.loss_and_grads(spec, p64, x.astype(np.float64), y, cfg, training=True)
res = {}
for dt in ('gen', 'f32', 'bf16'):
    m = dev.DeviceModel('unet', 1, S, S, B, dtype='f32' if dt == 'gen' else dt, force_generic=(dt == 'gen'), **full)
    m.set_params(O.flatten(spec, params))
    out = m.train_step(x, y, 0.0, m.loss_cfg(**cfg))
    g = m.get_grads()
    res[dt] = (out.loss, g)
    m.close()
print('loss oracle %.6f gen %.6f f32 %.6f bf16 %.6f' % (loss, res['gen'][0], res['f32'][0], res['bf16'][0]))
off = 0
for n, shape, t in O.param_specs(spec):
    if not t: continue
    size = int(np.prod(shape)); ref = grads[n].ravel()
    e = []
    for dt in ('gen', 'f32', 'bf16'):
        g = res[dt][1][off:off + size]
        e.append(np.linalg.norm(g - ref) / (np.linalg.norm(ref) + 1e-30))
    off += size
    if n.endswith('kernel') or n.endswith('gamma'):
        print('%-34s |g| %.3e  relL2 generic %.2e  f32 %.2e  bf16 %.2e' % (n, np.linalg.norm(ref), e[0], e[1], e[2]))
