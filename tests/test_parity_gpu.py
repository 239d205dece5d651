"""-m gpu: the HIP path (through the C ABI) against the committed golden vectors and the numpy oracle.

Tolerances (fp32 arithmetic, different summation order than the float64 oracle; the fixtures record what plain
float32 numpy costs in `fp32_logit_err` / `fp32_grad_err`):
  logits   |d| <= 2e-4 * max(1, |logits|max)      (BN networks amplify rounding: measured fp32-numpy error 4e-5)
  loss     rel  <= 1e-4
  grads    max|d| / max|g| <= 2e-3 per flat vector (atomics: order differs run to run)
  weights after one Adam step: |d| <= 2e-4 * lr-normalised step (Adam's first step is +-lr for every weight whose
           gradient sign is resolved, so the comparison is made on weights with |g| above the noise floor)
  masks    bit-exact wherever the oracle logit is farther than the logit tolerance from the threshold
"""

import numpy as np
import pytest

import helpers as Hp
from oracle import unet_oracle as O

pytestmark = pytest.mark.gpu

LOGIT_TOL = 2e-4
GRAD_TOL = 2e-3


def _run_case(gpu, name, force_generic):
    z, spec, loss_cfg = Hp.load_case(name)
    x, y = z['x'], z['y']
    B, H, W, _ = x.shape
    p0, s0 = Hp.case_params(z, spec)
    m = gpu.DeviceModel(**Hp.device_kwargs(spec, H, W, B, force_generic=force_generic))
    try:
        infos = m.param_infos()
        want = O.param_specs(spec)
        assert [(n, tuple(s), t) for n, s, t, _ in infos] == [(n, tuple(s), t) for n, s, t in want]
        m.set_params(p0)
        if m.n_state:
            m.set_state(s0)
        cfg = m.loss_cfg(**loss_cfg)

        # inference mode first (does not touch the state)
        prob, logits = m.forward(x, training=False, return_logits=True)
        tol = LOGIT_TOL * max(1.0, float(np.abs(z['logits_eval']).max()))
        assert np.abs(logits - z['logits_eval']).max() <= tol
        assert np.abs(prob - z['prob_eval']).max() <= tol
        for thr, key in ((0.5, 'mask05'), (0.8, 'mask08')):
            t_logit = np.log(thr / (1 - thr))
            decided = np.abs(z['logits_eval'] - t_logit) > tol
            assert np.array_equal((prob > thr)[decided], z[key][decided]), 'mask flip away from the threshold'
        out = m.eval_step(x, y, cfg)
        assert abs(out.loss - (float(z['loss_eval']) + O.l2_penalty(spec, O.unflatten(spec, p0)))) <= 1e-4 * max(1, abs(out.loss))

        # one training step
        out = m.train_step(x, y, float(z['lr']), cfg)
        assert abs(out.loss - float(z['loss_train'])) <= 1e-4 * max(1.0, abs(float(z['loss_train'])))
        assert abs(out.positive_rate - float(y.mean())) < 1e-6
        g = m.get_grads()
        pa = m.get_params()
        if 'grads' in z.files:
            gref, paref = z['grads'], z['params_after']
        else:
            st = int(z['sample_stride'])
            g, pa, gref, paref = g[::st], pa[::st], z['grads_sample'], z['params_after_sample']
            p0 = p0[::st]
        err = Hp.rel_err(g, gref)
        assert err <= GRAD_TOL, 'gradient max-norm relative error %.3e' % err
        # Adam's first step: |delta| = lr * |g| / (|g| + eps); compare where the gradient is well above rounding noise
        lr = float(z['lr'])
        big = np.abs(gref) > 100 * GRAD_TOL * np.abs(gref).max() * 1e-2
        assert np.abs((pa - paref)[big]).max() <= 0.05 * lr
        assert np.abs(pa - paref).max() <= 2.1 * lr      # never farther than a full step apart in either direction
        if m.n_state:
            assert np.abs(m.get_state() - z['state_after']).max() <= 1e-5
    finally:
        m.close()


@pytest.mark.parametrize('name', Hp.SMALL_CASES + Hp.BIG_CASES)
def test_golden_generic_kernels(gpu, name):
    _run_case(gpu, name, force_generic=True)


@pytest.mark.parametrize('name', Hp.SMALL_CASES + Hp.BIG_CASES)
def test_golden_tuned_kernels(gpu, name):
    _run_case(gpu, name, force_generic=False)
