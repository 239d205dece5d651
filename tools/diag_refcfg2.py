import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import helpers as Hp
from oracle import unet_oracle as O
from dnncancerannotator_amd import device
device.init_device(0)
arch, C, opts, B, size = ('mulmo', 3, dict(n_filters_first=16, n_downsample=4, bn=True), 2, 64)
full = dict(rate=2, kernel_size=3, conv_stride=1, padding='same', **opts)
spec = O.ModelSpec(arch, C, **full)
x, y = O.synthetic_batch(B, size, size, C)
cfg = dict(weight_mul=3.0)
params = O.init_params(spec, seed=2)
p64 = {n: v.astype(np.float64) for n, v in params.items()}
loss, grads, logits, state = O.loss_and_grads(spec, p64, x.astype(np.float64), y, cfg, training=True)
gref = O.flatten(spec, grads)
res = {}
for generic in (True, False):
    m = device.DeviceModel(arch, C, size, size, B, force_generic=generic, **full)
    m.set_params(O.flatten(spec, params))
    out = m.train_step(x, y, 0.0, m.loss_cfg(**cfg))
    res[generic] = m.get_grads().astype(np.float64)
    st = m.get_state()
    sref = O.flatten(spec, dict(p64, **state), trainable=False)
    print('state err', np.abs(st - sref).max())
    m.close()
eg = Hp.per_tensor_err(spec, res[True], gref)
et = Hp.per_tensor_err(spec, res[False], gref)
etg = Hp.per_tensor_err(spec, res[False], res[True])
for n in eg:
    print('%-34s generic %.1e tuned %.1e  tuned-vs-generic %.1e' % (n, eg[n], et[n], etg[n]))
