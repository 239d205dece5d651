"""Block-fused backward kernels (kernels_fused_bwd.hip) against the per-layer kernels they replace, block by block (development aid;
run on the GPU box).  Same device, same weights: every gradient tensor on its own scale; then the float64 oracle on a small shape.

    python tools/fzb_check.py [--full]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from dnncancerannotator_amd import device as dev                # noqa: E402
from dnncancerannotator_amd.synthetic import synthetic_batch    # noqa: E402
from oracle import unet_oracle as O                              # noqa: E402  (checker only)
import helpers as Hp                                             # noqa: E402

UNET = dict(n_filters_first=3, n_downsample=3, rate=2, kernel_size=3, conv_stride=1, bn=False, padding='same')
dev.init_device(0)
spec = O.ModelSpec('unet', 1, **UNET)


def run(B, H, W, env, leaky=0.0):
    for k in ('DNNCA_NO_FUSED_BWD', 'DNNCA_FZB_ONLY'):
        os.environ.pop(k, None)
    os.environ.update(env)
    x, y = synthetic_batch(B, H, W, 1)
    kw = dict(leaky_alpha=leaky) if leaky else {}
    m = dev.DeviceModel('unet', 1, H, W, B, **UNET, **kw)
    m.init_glorot(seed=2)
    p0 = m.get_params()
    m.set_params((p0 + np.random.default_rng(7).uniform(-0.05, 0.05, p0.shape)).astype(np.float32))
    out = m.train_step(x, y, 0.0, m.loss_cfg(weight_mul=3.0))
    g = m.get_grads().astype(np.float64)
    plan = [r[0] for r in m.plan()]
    m.close()
    return out.loss, g, plan


def per_tensor(g, gref):
    worst = []
    for name, sl in Hp.tensor_slices(spec):
        s = np.abs(gref[sl]).max()
        worst.append((np.abs(g[sl] - gref[sl]).max() / max(s, 1e-30), name))
    return sorted(worst, reverse=True)


shapes = [(2, 64, 256), (3, 32, 128)] + ([(8, 512, 512), (16, 512, 512), (5, 256, 384)] if '--full' in sys.argv else [])
for leaky in (0.0, 0.3):
    for B, H, W in shapes:
        l0, g0, plan0 = run(B, H, W, {'DNNCA_NO_FUSED_BWD': '1'}, leaky)
        for only in ('up2', 'up1', 'down2', 'down1', None):
            env = {'DNNCA_FZB_ONLY': only} if only else {}
            l1, g1, plan1 = run(B, H, W, env, leaky)
            fz = [k for k in plan1 if k.startswith('fzb_')]
            w = per_tensor(g1, g0)
            print('leaky %.1f  %dx%dx%d  only=%-6s  fused: %-40s  worst %.2e (%s), 2nd %.2e (%s)' %
                  (leaky, B, H, W, only, ','.join(fz), w[0][0], w[0][1], w[1][0], w[1][1]), flush=True)

# float64 oracle, every variable on its own scale
B, H, W = 2, 64, 256
params = Hp.perturbed_params(spec, np.float64)
rng = np.random.default_rng(13)
x = rng.random((B, H, W, 1)).astype(np.float32)
y = (rng.random((B, H, W)) < 0.05).astype(np.float32)
cfg = dict(weight_mul=3.0)
loss, grads, logits, _ = O.loss_and_grads(spec, params, x.astype(np.float64), y, cfg, training=True)
for k in ('DNNCA_NO_FUSED_BWD', 'DNNCA_FZB_ONLY'):
    os.environ.pop(k, None)
m = dev.DeviceModel('unet', 1, H, W, B, **UNET)
m.set_params(O.flatten(spec, params))
out = m.train_step(x, y, 0.0, m.loss_cfg(**cfg))
w = per_tensor(m.get_grads().astype(np.float64), O.flatten(spec, grads))
print('oracle 2x64x256: loss %.6f vs %.6f; worst tensors' % (out.loss, loss), ['%.2e %s' % t for t in w[:4]])
print('plan:', [r[0] for r in m.plan()])
m.close()
