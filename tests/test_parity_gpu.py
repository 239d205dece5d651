"""-m gpu: the HIP path (through the C ABI) against the committed golden vectors and the numpy oracle.

Tolerances (fp32 arithmetic, different summation order than the float64 oracle; the fixtures record what plain
float32 numpy costs in `fp32_logit_err` / `fp32_grad_err`):
  logits   |d| <= 2e-4 * max(1, |logits|max)      (BN networks amplify rounding: measured fp32-numpy error 4e-5)
  loss     rel  <= 1e-4
  grads    PER VARIABLE: max|g - gref| over a tensor <= 2e-5 * max|gref| over THAT tensor + 10 x what plain float32 numpy
           costs on that tensor (fixture `fp32_grad_abs_err_t`; ~3e-7 of the tensor's scale for a healthy variable, the whole
           value for the analytically-zero ones).  Evidence (tools/grad_spread.py on MI355X, profiles/r02_grad_spread.txt):
           device error <= 5e-7 per tensor on the configs/unet.yaml cases, <= 5 x the numpy float32 error on every tensor
           of every case; run-to-run spread of the float atomics <= 7e-7 (generic), 0 (tuned, small cases).
  weights after one Adam step: (w1 - w0) / lr against the fixture within 2e-3 wherever |g| > 1000 eps (Adam's first step is
           -lr * g / (|g| + eps): a sign test there), and k_adam itself against oracle.adam_step on the DEVICE's gradient
           with random non-trivial slots m, v at iteration 7: |dw - dw_ref| <= 1e-5 * lr
  masks    bit-exact wherever the oracle logit is farther than the logit tolerance from the threshold
"""

import numpy as np
import pytest

import helpers as Hp
from oracle import unet_oracle as O

pytestmark = pytest.mark.gpu

LOGIT_TOL = 2e-4
GRAD_TOL = 2e-5          # per tensor, on top of 10 x the float32-numpy noise of that tensor
ADAM_EPS = 1e-7


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
        gref = z['grads']
        Hp.assert_grads_per_tensor(spec, g, gref, GRAD_TOL, floor=10 * z['fp32_grad_abs_err_t'])
        # Adam's first step: delta = -lr * g / (|g| + eps)
        lr = float(z['lr'])
        if 'params_after' in z.files:
            paref = z['params_after']
        else:
            new = O.adam_step(O.unflatten(spec, p0.astype(np.float64)), O.unflatten(spec, gref.astype(np.float64)), {}, {}, 1, lr)
            paref = O.flatten(spec, new)
        # ... a sign test wherever the gradient is well away from zero: |g| > 1000 eps AND > 20 x the error the gradient check above
        # allows that variable (its own scale x GRAD_TOL + the float32-numpy floor: on the deep BatchNorm fixtures one ReLU / max-pool
        # decision that float32 takes differently from float64 moves a variable by 1e-4 of its scale -- whichever float32 summation
        # order is used -- and an entry of that size must not be asked for its sign)
        big = np.abs(gref) > 1000 * ADAM_EPS
        for i, (n, sl) in enumerate(Hp.tensor_slices(spec)):
            allowed = GRAD_TOL * np.abs(gref[sl]).max() + 10 * float(z['fp32_grad_abs_err_t'][i])
            big[sl] &= np.abs(gref[sl]) > 20 * allowed
        assert np.abs(((pa - p0) - (paref - p0))[big]).max() <= 2e-3 * lr
        assert np.abs(pa - paref).max() <= 2.0 * lr + 1e-7   # the rest: never farther than a full step each way
        if m.n_state:
            assert np.abs(m.get_state() - z['state_after']).max() <= 1e-5
        if not force_generic:
            Hp.record_oracle_plan(m, 'test_golden_tuned_kernels')
    finally:
        m.close()


@pytest.mark.parametrize('name,force_generic', [('unet_yaml_2x32', False), ('unet_yaml_2x32', True), ('unet_big_f8_2x64', False)])
def test_adam_kernel_with_history(gpu, name, force_generic):
    """engine.py:276-284 Keras Adam: the fused update against oracle.adam_step on the device's own gradient, starting from
    random slots m, v at iteration 6 (so the step depends on gradient MAGNITUDES, not only on signs)."""
    z, spec, loss_cfg = Hp.load_case(name)
    x, y = z['x'], z['y']
    B, H, W, _ = x.shape
    p0, s0 = Hp.case_params(z, spec)
    rng = np.random.default_rng(21)
    gscale = np.abs(z['grads']) + 1e-6
    m0 = (rng.standard_normal(p0.shape) * gscale).astype(np.float32)
    v0 = (rng.random(p0.shape) * gscale ** 2).astype(np.float32)
    m = gpu.DeviceModel(**Hp.device_kwargs(spec, H, W, B, force_generic=force_generic))
    try:
        m.set_params(p0)
        if m.n_state:
            m.set_state(s0)
        m.set_opt_state(m0, v0, 6)
        lr = 3e-3
        m.train_step(x, y, lr, m.loss_cfg(**loss_cfg))
        g = m.get_grads().astype(np.float64)
        m1, v1, it = m.get_opt_state()
        assert it == 7
        mm, vv = O.unflatten(spec, m0.astype(np.float64)), O.unflatten(spec, v0.astype(np.float64))
        new = O.adam_step(O.unflatten(spec, p0.astype(np.float64)), O.unflatten(spec, g), mm, vv, 7, lr)
        want = O.flatten(spec, new)
        assert np.abs((m.get_params() - p0) - (want - p0)).max() <= 1e-5 * lr + 1.5e-7   # 1.5e-7: the float32 rounding of weights in [1, 2) and of the difference
        assert np.abs(m1 - O.flatten(spec, mm)).max() <= 1e-6 * np.abs(O.flatten(spec, mm)).max()
        assert np.abs(v1 - O.flatten(spec, vv)).max() <= 1e-6 * np.abs(O.flatten(spec, vv)).max()
    finally:
        m.close()


@pytest.mark.parametrize('name', Hp.SMALL_CASES + Hp.BIG_CASES)
def test_golden_generic_kernels(gpu, name):
    _run_case(gpu, name, force_generic=True)


@pytest.mark.parametrize('name', Hp.SMALL_CASES + Hp.BIG_CASES)
def test_golden_tuned_kernels(gpu, name):
    _run_case(gpu, name, force_generic=False)
