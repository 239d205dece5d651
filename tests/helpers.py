"""Shared helpers of the parity tests: fixture loading and oracle <-> device-model plumbing."""

import json
import os

import numpy as np

from oracle import unet_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
SMALL_CASES = ['unet_yaml_2x32', 'unet_yaml_1x64_nopos', 'unet_bn_leaky_l2_2x16', 'mulmo_small_2x16', 'unet_fixedw_2x16']
BIG_CASES = ['mulmo_yaml_2x64', 'unet_big_f8_2x64']


PARAM_SEED = 2


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
