import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import helpers as Hp
from oracle import unet_oracle as O
from dnncancerannotator_amd import device
device.init_device(0)
for arch, C, opts, B, size in [('mulmo', 3, dict(n_filters_first=16, n_downsample=4, bn=True), 2, 64),
                               ('unet', 1, dict(n_filters_first=64, n_downsample=4, bn=True), 1, 64)]:
    full = dict(rate=2, kernel_size=3, conv_stride=1, padding='same', **opts)
    spec = O.ModelSpec(arch, C, **full)
    x, y = O.synthetic_batch(B, size, size, C)
    cfg = dict(weight_mul=3.0)
    for pname in ('init', 'perturbed'):
        params = O.init_params(spec, seed=2) if pname == 'init' else Hp.perturbed_params(spec, np.float32)
        p64 = {n: v.astype(np.float64) for n, v in params.items()}
        loss, grads, logits, state = O.loss_and_grads(spec, p64, x.astype(np.float64), y, cfg, training=True)
        gref = O.flatten(spec, grads)
        for generic in (True, False):
            m = device.DeviceModel(arch, C, size, size, B, force_generic=generic, **full)
            m.set_params(O.flatten(spec, params))
            if m.n_state: m.set_state(O.flatten(spec, params, trainable=False))
            _, lg = m.forward(x, training=True, return_logits=True)
            m.set_state(O.flatten(spec, params, trainable=False))
            out = m.train_step(x, y, 0.0, m.loss_cfg(**cfg))
            e = Hp.per_tensor_err(spec, m.get_grads(), gref)
            deg = Hp.degenerate_tensors(spec)
            h = {n: v for n, v in e.items() if n not in deg}
            w = sorted(h.items(), key=lambda kv: -kv[1])[:4]
            print(arch, pname, 'generic' if generic else 'tuned', 'logit err %.2e loss err %.2e' % (np.abs(lg - logits).max(), abs(out.loss - loss)),
                  'median %.1e max %.1e' % (np.median(list(h.values())), max(h.values())), w, flush=True)
            m.close()
