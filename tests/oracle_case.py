"""One train step of a small network on the device against the float64 oracle, in a process of its own: for kernel variants that
the library picks by size (or by a per-process switch such as DNNCA_IG_NW) and that no small shape selects by itself.
    python tests/oracle_case.py '<json: arch, C, opts, B, H, W, alpha>'   ->  one JSON line (loss, per-tensor errors, fp32 floors, plan)
Test infrastructure (the parent test sets the switches in the child's environment and judges the numbers)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from oracle import unet_oracle as O          # noqa: E402
import helpers as Hp                          # noqa: E402


def run(device, arch, C, opts, B, H, W, alpha=0.0, seed=31):
    full = dict(rate=2, kernel_size=3, conv_stride=1, padding='same', **opts)
    kw = dict(activation={'class_name': 'LeakyReLU', 'config': {'alpha': alpha}}) if alpha else {}
    spec = O.ModelSpec(arch, C, **full, **kw)
    params = Hp.perturbed_params(spec, np.float64)
    rng = np.random.default_rng(seed)
    x = rng.random((B, H, W, C)).astype(np.float32)
    y = (rng.random((B, H, W)) < 0.05).astype(np.float32)
    cfg = dict(weight_mul=3.0)
    loss, grads, _, state = O.loss_and_grads(spec, params, x.astype(np.float64), y, cfg, training=True)
    p32 = {n: v.astype(np.float32) for n, v in params.items()}
    _, g32, _, _ = O.loss_and_grads(spec, p32, x, y, cfg, training=True)
    gref, g32 = O.flatten(spec, grads), O.flatten(spec, g32).astype(np.float64)
    m = device.DeviceModel(arch, C, H, W, B, **full, **(dict(leaky_alpha=alpha) if alpha else {}))
    m.set_params(O.flatten(spec, params))
    if m.n_state:
        m.set_state(O.flatten(spec, params, trainable=False))
    out = m.train_step(x, y, 0.0, m.loss_cfg(**cfg))
    g = m.get_grads().astype(np.float64)
    plan = sorted(set(r[0] for r in m.plan(variants=True)))
    m.close()
    errs, floors = {}, {}
    for n, sl in Hp.tensor_slices(spec):
        s = np.abs(gref[sl]).max() + 1e-300
        errs[n] = float(np.abs(g[sl] - gref[sl]).max() / s)
        floors[n] = float(10 * np.abs(g32[sl] - gref[sl]).max() / s)
    return dict(loss=float(out.loss), loss_ref=float(loss), errs=errs, floors=floors, plan=plan)


if __name__ == '__main__':
    from dnncancerannotator_amd import device as dev
    dev.init_device(0)
    print(json.dumps(run(dev, **json.loads(sys.argv[1]))))
