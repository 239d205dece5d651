"""Generates tests/golden/*.npz: small input/expected-output vectors for the train/evaluate hot path.

Expected values come from the numpy oracle (oracle/unet_oracle.py) evaluated in float64 and are
cross-checked here against an independent torch-CPU autograd statement (tests/torch_ref.py) before
being written.  The reference itself cannot be run (TensorFlow is not installed), so
these vectors are the pin: "parity unpinned" by the reference, pinned by two independent restatements.

Run from the repo root:  python tests/golden/make_golden.py
"""

import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..'))
sys.path.insert(0, os.path.join(HERE, '..'))

from oracle import unet_oracle as O   # noqa: E402
import torch_ref                      # noqa: E402
from helpers import perturbed_params, tensor_slices, PARAM_SEED   # noqa: E402

LEAKY = {'class_name': 'LeakyReLU', 'config': {'alpha': 0.3}}
L2 = {'class_name': 'L2', 'config': {'l2': 0.01}}

# name -> (spec kwargs, B, H, W, loss_cfg, empty_first, store_params)
CASES = {
    # configs/unet.yaml hyper-parameters, tiny image
    'unet_yaml_2x32': (dict(arch='unet', in_channels=1, n_filters_first=3, n_downsample=3, bn=False, padding='same'),
                       2, 32, 32, dict(weight_mul=3.0), False, True),
    # configs/unet.yaml, one 64x64 slice with NO positives -> weight = 1 branch (losses.py:27)
    'unet_yaml_1x64_nopos': (dict(arch='unet', in_channels=1, n_filters_first=3, n_downsample=3, bn=False, padding='same'),
                             1, 64, 64, dict(weight_mul=3.0), True, True),
    # BN + LeakyReLU + L2 + weight_add, two input channels
    'unet_bn_leaky_l2_2x16': (dict(arch='unet', in_channels=2, n_filters_first=4, n_downsample=2, bn=True, padding='same',
                                   activation=LEAKY, kernel_regularizer=L2),
                              2, 16, 16, dict(weight_mul=3.0, weight_add=0.5), False, True),
    # mulmo structure, small filters
    'mulmo_small_2x16': (dict(arch='mulmo', in_channels=3, n_filters_first=4, n_downsample=2, bn=True, padding='same'),
                         2, 16, 16, dict(weight_mul=3.0), False, True),
    # configs/mulmo_unet.yaml hyper-parameters (f0 16, 4 levels), 2x64x64x3 (weights regenerated from the seed);
    # 64x64 keeps 32 samples per channel in the deepest BatchNorm (a 2x2 bottleneck makes BN ill-conditioned)
    'mulmo_yaml_2x64': (dict(arch='mulmo', in_channels=3, n_filters_first=16, n_downsample=4, bn=True, padding='same'),
                        2, 64, 64, dict(weight_mul=3.0), False, False),
    # unet_big.yaml structure (4 levels, BN) at f0 = 8 so the fixture stays small
    'unet_big_f8_2x64': (dict(arch='unet', in_channels=1, n_filters_first=8, n_downsample=4, bn=True, padding='same'),
                         2, 64, 64, dict(weight_mul=3.0), False, False),
    # fixed weight (losses.py:24 weight is not None)
    'unet_fixedw_2x16': (dict(arch='unet', in_channels=1, n_filters_first=3, n_downsample=2, bn=False, padding='same'),
                         2, 16, 16, dict(weight=5.0, weight_mul=1.0, weight_add=0.0), False, True),
}

LR = 1e-3


def make_case(name):
    kw, B, H, W, loss_cfg, empty_first, store_params = CASES[name]
    spec = O.ModelSpec(**kw)
    x, y = O.synthetic_batch(B, H, W, spec.in_channels, empty_first=empty_first)
    p64 = perturbed_params(spec, np.float64)
    x64 = x.astype(np.float64)

    # float64 oracle, training mode, one Adam step (t = 1)
    m, v = {}, {}
    loss, new_params, grads, logits = O.train_step(spec, p64, m, v, 1, x64, y, LR, loss_cfg)
    # independent check
    ref = torch_ref.run(spec, p64, x64, y, loss_cfg, training=True)
    assert abs(loss - ref['loss']) < 1e-10 * max(1, abs(loss)), (loss, ref['loss'])
    assert np.abs(logits - ref['logits']).max() < 1e-10
    for n, g in grads.items():
        assert np.abs(g - ref['grads'][n]).max() <= 1e-9 * np.abs(ref['grads'][n]).max() + 1e-13, n
    for n, s in ref['state'].items():
        assert np.abs(new_params[n] - s).max() < 1e-10, n

    # inference mode (moving statistics)
    prob_eval, logits_eval = O.predict(spec, p64, x64)
    ref_eval = torch_ref.run(spec, p64, x64, y, loss_cfg, training=False)
    assert np.abs(logits_eval - ref_eval['logits']).max() < 1e-10
    per_eval, _ = O.weighted_crossentropy(y, logits_eval, **loss_cfg)

    # float32 oracle: what plain fp32 arithmetic costs (sets the test tolerances)
    p32 = {n: a.astype(np.float32) for n, a in p64.items()}
    loss32, grads32, logits32, _ = O.loss_and_grads(spec, p32, x, y, loss_cfg, training=True)
    g64 = O.flatten(spec, grads)
    g32 = O.flatten(spec, grads32)

    out = dict(
        spec=json.dumps(kw), loss_cfg=json.dumps(loss_cfg), param_seed=PARAM_SEED, lr=LR,
        x=x, y=y,
        logits_train=logits.astype(np.float32), loss_train=np.float64(loss),
        grad_norms=np.array([np.sqrt((np.asarray(grads[n], np.float64) ** 2).sum())
                             for n, _, t in O.param_specs(spec) if t]),
        state_after=O.flatten(spec, new_params, trainable=False).astype(np.float32),
        logits_eval=logits_eval.astype(np.float32), prob_eval=prob_eval.astype(np.float32),
        loss_eval=np.float64(per_eval.mean()),
        mask05=(prob_eval > 0.5), mask08=(prob_eval > 0.8),
        fp32_logit_err=np.float64(np.abs(logits32 - logits).max()),
        fp32_grad_err=np.float64(np.abs(g32 - g64).max() / np.abs(g64).max()),
        torch_logit_err=np.float64(np.abs(logits - ref['logits']).max()),
    )
    pa = O.flatten(spec, new_params).astype(np.float32)
    # full gradients for every case (the two big ones too: every variable is compared on its own scale)
    out['grads'] = g64.astype(np.float32)
    if len(g64) <= 50000:
        out['params_after'] = pa          # big cases: the test re-derives them with O.adam_step from `grads` (halves the fixture)
    # what plain float32 numpy costs PER VARIABLE (max |g32 - g64| inside each tensor, absolute): the evidence the per-tensor
    # tolerance of the GPU tests is derived from.  Variables whose gradient is analytically zero (a bias feeding a
    # BatchNorm, the last BatchNorm of a mulmo encoder whose skips are unused) are pure rounding noise in any fp32
    # implementation; this is their noise floor.
    out['fp32_grad_abs_err_t'] = np.array([np.abs(g32[sl].astype(np.float64) - g64[sl]).max() for _, sl in tensor_slices(spec)])
    if store_params:
        out['params'] = O.flatten(spec, p64).astype(np.float32)
        out['state'] = O.flatten(spec, p64, trainable=False).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print('%-24s loss %.6f  n_params %d  fp32 logit err %.2e  grad err %.2e' % (
        name, loss, len(g64), out['fp32_logit_err'], out['fp32_grad_err']))


if __name__ == '__main__':
    for name in (sys.argv[1:] or CASES):
        make_case(name)
