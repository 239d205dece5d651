"""unet_big B=4 512x512: bf16-tuned vs fp32-tuned (GPU box) -- evidence for the bounds of the full-size test"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import helpers as Hp
from oracle import unet_oracle as O
from dnncancerannotator_amd import device
from dnncancerannotator_amd.synthetic import synthetic_batch
device.init_device(0)
full = dict(rate=2, kernel_size=3, conv_stride=1, padding='same', n_filters_first=64, n_downsample=4, bn=True)
spec = O.ModelSpec('unet', 1, **full)
B, H, W = 4, 512, 512
x, y = synthetic_batch(B, H, W, 1)
res = {}
for dt in ('f32', 'bf16'):
    m = device.DeviceModel('unet', 1, H, W, B, dtype=dt, **full)
    m.init_glorot(seed=3)
    t = time.time()
    _, lg = m.forward(x, training=False, return_logits=True)
    out = m.train_step(x, y, 0.0, m.loss_cfg(weight_mul=3.0))
    res[dt] = (lg.copy(), out.loss, m.get_grads().astype(np.float64), m.get_state().copy(), sorted(set(r[0] for r in m.plan())))
    print(dt, 'loss', out.loss, 'time', time.time() - t, flush=True)
    m.close()
(l0, loss0, g0, s0, _), (l1, loss1, g1, s1, names) = res['f32'], res['bf16']
print('logit diff max %.3e  median %.3e  logits absmax %.3f' % (np.abs(l1 - l0).max(), np.median(np.abs(l1 - l0)), np.abs(l0).max()))
print('loss rel diff %.3e' % (abs(loss1 - loss0) / abs(loss0)))
print('state diff max %.3e (rel to max %.3e)' % (np.abs(s1 - s0).max(), np.abs(s1 - s0).max() / np.abs(s0).max()))
cos, rel = {}, {}
for n, sl in Hp.tensor_slices(spec):
    a, b = g1[sl], g0[sl]
    cos[n] = float(np.dot(a, b) / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))
    rel[n] = float(np.abs(a - b).max() / (np.abs(b).max() + 1e-300))
deg = Hp.degenerate_tensors(spec)
h = {n: v for n, v in cos.items() if n not in deg}
print('cos: min %.5f' % min(h.values()), sorted(h.items(), key=lambda kv: kv[1])[:6])
print('rel max-norm: max %.3e' % max(v for n, v in rel.items() if n not in deg), sorted(((n, v) for n, v in rel.items() if n not in deg), key=lambda kv: -kv[1])[:6])
print([n for n in names if 'igb' in n])
