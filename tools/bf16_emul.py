import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from dnncancerannotator_amd import device as dev
from oracle import unet_oracle as O
import helpers as Hp
def to_bf16(a):
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).astype(np.float64)
dev.init_device(0)
opts = dict(rate=2, kernel_size=3, conv_stride=1, padding='same', n_filters_first=32, n_downsample=2, bn=False)
spec = O.ModelSpec('unet', 32, **opts)
params = Hp.perturbed_params(spec, np.float64)
rng = np.random.default_rng(3)
B, S = 2, 32
x = rng.random((B, S, S, 32)).astype(np.float32)
_, lref, _, acts_ref = O.forward(spec, params, x.astype(np.float64), keep=True)[0], None, None, None
fwd0 = O.conv2d_fwd
def fwd(xx, w, b, padding, alpha=None):
    if w.shape[0] == 1: return fwd0(xx, w, b, padding, alpha)
    return fwd0(to_bf16(xx), to_bf16(w), b, padding, alpha)
l64 = O.forward(spec, params, x.astype(np.float64))[0]
O.conv2d_fwd = fwd
lem = O.forward(spec, params, x.astype(np.float64))[0]
O.conv2d_fwd = fwd0
for dt in ('f32', 'bf16'):
    m = dev.DeviceModel('unet', 32, S, S, B, dtype=dt, **opts)
    m.set_params(O.flatten(spec, params))
    _, lg = m.forward(x, training=False, return_logits=True)
    print(dt, 'max|logit - f64 oracle| %.3e   max|logit - bf16-emulating oracle| %.3e' % (np.abs(lg - l64).max(), np.abs(lg - lem).max()))
    print('   kernels:', sorted(set(r[0] for r in m.plan() if 'conv' in r[0] or 'wgrad' in r[0])))
    m.close()
print('emulated vs f64 oracle %.3e' % np.abs(lem - l64).max())
