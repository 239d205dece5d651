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
deg = Hp.degenerate_tensors(spec)
cfg = dict(weight_mul=3.0)
for seed in (2, 3, 4, 5):
    for xs in (0, 10):
        x, y = O.synthetic_batch(B, size, size, C, seed_x=xs)
        params = O.init_params(spec, seed=seed)
        p64 = {n: v.astype(np.float64) for n, v in params.items()}
        loss, grads, logits, state = O.loss_and_grads(spec, p64, x.astype(np.float64), y, cfg, training=True)
        gref = O.flatten(spec, grads)
        _, g32, _, _ = O.loss_and_grads(spec, params, x, y, cfg, training=True)
        e32 = Hp.per_tensor_err(spec, O.flatten(spec, g32).astype(np.float64), gref)
        line = 'seed %d x%d numpy32: n>1e-4 %d' % (seed, xs, sum(1 for n, v in e32.items() if n not in deg and v > 1e-4))
        for generic in (True, False):
            m = device.DeviceModel(arch, C, size, size, B, force_generic=generic, **full)
            m.set_params(O.flatten(spec, params))
            m.train_step(x, y, 0.0, m.loss_cfg(**cfg))
            e = Hp.per_tensor_err(spec, m.get_grads(), gref)
            m.close()
            h = {n: v for n, v in e.items() if n not in deg}
            bad = [n for n, v in h.items() if v > 1e-4]
            line += ' | %s: median %.1e n>1e-4 %d max %.1e first-bad(bwd order) %s' % ('gen' if generic else 'tun', np.median(list(h.values())), len(bad), max(h.values()), bad[-1] if bad else '-')
        print(line, flush=True)
