"""fp32 by three bf16 planes (kernels_ig3x.hip) against the exact-fp32 MFMA kernels: forward logits and per-tensor gradient error of
small dense networks against the float64 oracle.  Run on the GPU box, once per mode (the switch is read once per process):
    python tools/x3_precision.py            # split-bf16 kernels
    DNNCA_NO_X3=1 python tools/x3_precision.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import helpers as Hp
from oracle import unet_oracle as O
from dnncancerannotator_amd import device

device.init_device(0)
print('mode:', 'exact fp32 MFMA' if os.environ.get('DNNCA_NO_X3') else 'split bf16 x3')
for arch, C, opts, B, S in [('unet', 1, dict(n_filters_first=16, n_downsample=1, bn=False), 2, 64),
                            ('unet', 1, dict(n_filters_first=16, n_downsample=2, bn=False), 2, 64),
                            ('unet', 1, dict(n_filters_first=64, n_downsample=1, bn=False), 2, 32),
                            ('mulmo', 3, dict(n_filters_first=16, n_downsample=2, bn=True), 2, 64)]:
    full = dict(rate=2, kernel_size=3, conv_stride=1, padding='same', **opts)
    spec = O.ModelSpec(arch, C, **full)
    params = Hp.perturbed_params(spec, np.float64)
    rng = np.random.default_rng(7)
    x = rng.random((B, S, S, C)).astype(np.float32)
    y = (rng.random((B, S, S)) < 0.05).astype(np.float32)
    cfg = dict(weight_mul=3.0)
    _, lref = O.predict(spec, params, x.astype(np.float64))
    loss, grads, _, _ = O.loss_and_grads(spec, params, x.astype(np.float64), y, cfg, training=True)
    gref = O.flatten(spec, grads)
    m = device.DeviceModel(arch, C, S, S, B, **full)
    m.set_params(O.flatten(spec, params))
    if m.n_state:
        m.set_state(O.flatten(spec, params, trainable=False))
    _, lg = m.forward(x, training=False, return_logits=True)
    out = m.train_step(x, y, 0.0, m.loss_cfg(**cfg))
    err = Hp.per_tensor_err(spec, m.get_grads(), gref)
    worst = sorted(err.items(), key=lambda kv: -kv[1])[:4]
    print('%-6s f0=%-3d down=%d bn=%d: logits max err %.2e (rel. to max |logit| %.2f)  loss err %.1e  grads: median %.1e max %.1e  worst %s' % (
        arch, opts['n_filters_first'], opts['n_downsample'], opts['bn'], np.abs(lg - lref).max() / np.abs(lref).max(), np.abs(lref).max(),
        abs(out.loss - loss) / abs(loss), np.median(list(err.values())), max(err.values()), ' '.join('%s=%.1e' % w for w in worst)))
    print('   plan:', sorted(set(r[0] for r in m.plan(variants=True) if 'conv' in r[0])))
    m.close()
