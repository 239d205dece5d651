"""Evidence for the per-tensor gradient tolerance: for every golden case x {generic, tuned} the per-variable error of the
device gradient against the float64 fixture and the run-to-run spread (float atomics).  Run on the GPU box."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import helpers as Hp
from oracle import unet_oracle as O
from dnncancerannotator_amd import device

device.init_device(0)
res = {}
for name in Hp.SMALL_CASES + Hp.BIG_CASES:
    z, spec, loss_cfg = Hp.load_case(name)
    x, y = z['x'], z['y']
    B, H, W, _ = x.shape
    p0, s0 = Hp.case_params(z, spec)
    if 'grads' in z.files:
        gref = z['grads'].astype(np.float64)
    else:
        p = Hp.perturbed_params(spec, np.float64)
        _, grads, _, _ = O.loss_and_grads(spec, p, x.astype(np.float64), y, loss_cfg, training=True)
        gref = O.flatten(spec, grads)
    for generic in (True, False):
        m = device.DeviceModel(**Hp.device_kwargs(spec, H, W, B, force_generic=generic))
        runs = []
        for r in range(3):
            m.set_params(p0)
            if m.n_state:
                m.set_state(s0)
            m.train_step(x, y, 0.0, m.loss_cfg(**loss_cfg))
            runs.append(m.get_grads().astype(np.float64))
        m.close()
        err = Hp.per_tensor_err(spec, runs[0], gref)
        spread = Hp.per_tensor_err(spec, runs[1], runs[0])
        spread2 = Hp.per_tensor_err(spec, runs[2], runs[0])
        worst = sorted(err.items(), key=lambda kv: -kv[1])[:5]
        print('%-24s %-7s max err %.2e  max spread %.2e   worst: %s' % (
            name, 'generic' if generic else 'tuned', max(err.values()), max(max(spread.values()), max(spread2.values())),
            ' '.join('%s=%.1e' % w for w in worst)), flush=True)
        res['%s/%s' % (name, 'generic' if generic else 'tuned')] = dict(err=err, spread=spread)
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, 'gpurun_out', 'grad_spread.json'), 'w'), indent=1)
