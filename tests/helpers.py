"""Shared helpers of the parity tests: fixture loading and oracle <-> device-model plumbing."""

import json
import os
from collections import OrderedDict

import numpy as np

from oracle import unet_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
SMALL_CASES = ['unet_yaml_2x32', 'unet_yaml_1x64_nopos', 'unet_bn_leaky_l2_2x16', 'mulmo_small_2x16', 'unet_fixedw_2x16']
BIG_CASES = ['mulmo_yaml_2x64', 'unet_big_f8_2x64']


PARAM_SEED = 2

# ---- which kernels have met the oracle?  Every -m gpu test that compares a train step with the float64 / bf16-emulating oracle
# (or with a golden fixture made by it) registers the launch names of the model it checked; tests/test_zz_kernel_coverage.py
# closes the loop: every launch of the three BASELINE configurations at full size must be in this set.
ORACLE_KERNELS = set()
ORACLE_TESTS = set()


def record_oracle_plan(model_or_names, test):
    """`model_or_names`: a DeviceModel (its plan() -- the launch schedule of one train step under the current switches) or an
    iterable of launch names (a child process's); `test`: the registering test function's name."""
    names = model_or_names if isinstance(model_or_names, (list, tuple, set)) else [r[0] for r in model_or_names.plan(variants=True)]
    ORACLE_KERNELS.update(names)
    ORACLE_TESTS.add(test)



def perturbed_params(spec, dtype):
    """glorot kernels (seed 2) + small non-zero biases / BN parameters so every term is exercised."""
    params = O.init_params(spec, seed=PARAM_SEED, dtype=np.float64)
    rng = np.random.default_rng(PARAM_SEED + 1)
    for n in params:
        if n.endswith('.kernel'):
            continue
        params[n] = params[n] + rng.uniform(-0.1, 0.1, params[n].shape)
        if n.endswith('moving_variance') or n.endswith('gamma'):
            params[n] = np.abs(params[n])
    # round through float32 so the fp32 product path and the float64 oracle see identical inputs
    return {n: v.astype(np.float32).astype(dtype) for n, v in params.items()}


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    kw = json.loads(str(z['spec']))
    spec = O.ModelSpec(**kw)
    loss_cfg = json.loads(str(z['loss_cfg']))
    return z, spec, loss_cfg


def case_params(z, spec):
    """float32 trainable / state flat vectors of a fixture (regenerated from the seed when not stored)."""
    if 'params' in z.files:
        return z['params'], z['state']
    p = perturbed_params(spec, np.float64)
    return O.flatten(spec, p).astype(np.float32), O.flatten(spec, p, trainable=False).astype(np.float32)


def device_kwargs(spec, H, W, max_batch, **extra):
    kw = dict(arch=spec.arch, in_channels=spec.in_channels, height=H, width=W, max_batch=max_batch,
              n_filters_first=spec.f0, n_downsample=spec.n_down, rate=spec.rate, kernel_size=spec.k, conv_stride=1,
              bn=spec.bn, padding=spec.padding, leaky_alpha=spec.alpha, l2=spec.l2, reference_index=spec.reference_index,
              n_conv=spec.n_conv)
    kw.update(extra)
    return kw


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def tensor_slices(spec, trainable=True):
    """[(name, slice)] of every variable inside the flat trainable (or state) vector, Keras creation order."""
    out, off = [], 0
    for n, shape, t in O.param_specs(spec):
        if t != trainable:
            continue
        size = int(np.prod(shape))
        out.append((n, slice(off, off + size)))
        off += size
    return out


def per_tensor_err(spec, g, gref):
    """{name: max|g - gref| / max|gref|} with BOTH maxima taken inside that variable's own slice -- a gradient tensor that
    is small next to the head's gradient is judged against its own scale (components.py:46-52,118-127: every variable
    gets its gradient from GradientTape).  A tensor whose reference gradient is exactly zero must be exactly zero-ish:
    its error is reported against the global scale times 1e-6."""
    g = np.asarray(g, np.float64)
    gref = np.asarray(gref, np.float64)
    assert g.shape == gref.shape, (g.shape, gref.shape)
    gmax = np.abs(gref).max() + 1e-300
    out = OrderedDict()
    for n, sl in tensor_slices(spec):
        scale = np.abs(gref[sl]).max()
        if scale < 1e-12 * gmax:
            scale = 1e-6 * gmax
        out[n] = float(np.abs(g[sl] - gref[sl]).max() / scale)
    return out


def assert_grads_per_tensor(spec, g, gref, tol, floor=None, what='gradient'):
    """Every variable's gradient within `tol` of the reference RELATIVE TO THAT VARIABLE'S OWN largest entry:
        max|g - gref| over the tensor  <=  tol * max|gref| over the tensor  (+ floor[tensor])
    `floor` (absolute, one entry per trainable variable in creation order) is the noise floor of variables whose gradient is
    analytically zero -- a bias that feeds a BatchNorm, the last BatchNorm of a mulmo encoder whose skips are unused --
    and which therefore consist of fp32 rounding noise in ANY fp32 implementation: the golden fixtures store what plain
    float32 numpy costs per variable (`fp32_grad_abs_err_t`) and the tests pass 10x that.  Returns {name: relative error}."""
    g = np.asarray(g, np.float64)
    gref = np.asarray(gref, np.float64)
    assert g.shape == gref.shape, (g.shape, gref.shape)
    assert np.isfinite(g).all(), what + ': non-finite entries'
    errs, bad = OrderedDict(), []
    for i, (n, sl) in enumerate(tensor_slices(spec)):
        scale = np.abs(gref[sl]).max()
        d = np.abs(g[sl] - gref[sl]).max()
        errs[n] = float(d / (scale + 1e-300))
        if not d <= tol * scale + (0.0 if floor is None else float(floor[i])):
            bad.append((n, errs[n]))
    assert not bad, '%s: %d of %d tensors off (tol %.1e): %s' % (
        what, len(bad), len(errs), tol, ', '.join('%s %.2e' % b for b in bad[:8]))
    return errs


def degenerate_tensors(spec):
    """Names of the trainable variables whose gradient is analytically (almost) zero, i.e. rounding noise in fp32:
      * `*.tconv.bias` under bn: the transposed conv feeds a BatchNorm directly (components.py:118-120,131), which removes
        any per-channel constant;
      * `encoderE.downI.bn{last}.{beta,gamma}` of the mulmo encoders whose skips are unused (E != reference_index,
        unet.py:188): the BatchNorm output only reaches max-pool -> pool BatchNorm (components.py:54,59), which commutes
        with a per-channel shift (exactly) and positive scale (up to the BatchNorm epsilon).
    Tests that have no float32-noise fixture (tuned vs generic at full size) judge these against the scale of a healthy
    sibling instead (`sibling_of`)."""
    out = set()
    if not spec.bn:
        return out
    for n, _ in tensor_slices(spec):
        if n.endswith('.tconv.bias'):
            out.add(n)
    if spec.arch == 'mulmo':
        last = 'bn%d' % (spec.n_conv - 1)
        for e in range(spec.n_encoders()):
            if e == spec.reference_index:
                continue
            for i in range(spec.n_down):
                out.add('encoder%d.down%d.%s.beta' % (e, i, last))
                out.add('encoder%d.down%d.%s.gamma' % (e, i, last))
    return out


def sibling_of(name):
    """a healthy variable fed by the same upstream gradient as the degenerate variable `name`"""
    if name.endswith('.tconv.bias'):
        return name[:-len('bias')] + 'kernel'
    head, _ = name.rsplit('.', 1)                 # encoderE.downI.bnJ
    block, bn = head.rsplit('.', 1)
    return '%s.conv%s.bias' % (block, bn[2:])


def assert_grads_per_tensor_nofixture(spec, g, gref, tol, degenerate_tol=None, what='gradient'):
    """assert_grads_per_tensor for comparisons without a float32-noise fixture: degenerate variables (see
    degenerate_tensors) are measured against their sibling's scale with `degenerate_tol` (default: tol)."""
    g = np.asarray(g, np.float64)
    gref = np.asarray(gref, np.float64)
    sl = dict(tensor_slices(spec))
    deg = degenerate_tensors(spec)
    floor = [((degenerate_tol or tol) * np.abs(gref[sl[sibling_of(n)]]).max() if n in deg else 0.0) for n in sl]
    return assert_grads_per_tensor(spec, g, gref, tol, floor=floor, what=what)
