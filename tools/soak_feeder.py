#!/usr/bin/env python3
"""Soak of the threaded input pipeline: many train steps through BatchFeeder + staging ring with validation on the evaluation
slots in between, then repeated staged evaluations; compares the loss history with the synchronous loop's (DNNCA_NO_FEEDER=1) on
the same data.  A slot handed over too early, or an output read from the wrong step, shows up as a diverging history.

    python tools/soak_feeder.py [--steps 20000] [--size 64] [--batch 4]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dnncancerannotator_amd import data, engine      # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=20000)
ap.add_argument('--size', type=int, default=64)
ap.add_argument('--batch', type=int, default=4)
a = ap.parse_args()
config = {'model': 'UNetAnnotator',
          'model_options': dict(n_filters_first=3, n_downsample=3, rate=2, kernel_size=3, conv_stride=1, bn=False, padding='same'),
          'deploy_options': {'optimizer': 'adam', 'loss': {'class_name': 'WeightedCrossentropy', 'config': {'weight_mul': 3.0}},
                             'LearningRateScheduler': 'lambda epoch, current_lr: 0.001 * 0.96 ** (epoch // 1000)', 'enable_multigpu': False,
                             'metrics': [{'AUC': {'curve': 'ROC', 'name': 'pixel/AUROC', 'num_thresholds': 150}},
                                         {'FBetaScore': {'thresholds': 0.5, 'beta': 1.0, 'name': 'pixel/F1-score'}}]}}


def run(no_feeder):
    if no_feeder:
        os.environ['DNNCA_NO_FEEDER'] = '1'
    else:
        os.environ.pop('DNNCA_NO_FEEDER', None)
    ds = data.SyntheticDataset(a.batch, a.size, a.size, 1, n_batches=7, seed=3)
    val = data.SyntheticDataset(2 * a.batch, a.size, a.size, 1, n_batches=3, seed=9, repeat=False)
    m = engine.TFKerasModel(config)
    t0 = time.perf_counter()
    res = m.train(ds, val_data=val, max_steps=a.steps, save_freq=max(a.steps // 40, 1))
    dt = time.perf_counter() - t0
    evs = [m._evaluate(val, staged=not no_feeder) for _ in range(20)]
    assert all(e == evs[0] for e in evs), 'repeated evaluations differ'
    return res, evs[0], dt


ra, ea, ta = run(False)
rb, eb, tb = run(True)
rc, ec, tc = run(True)          # control: the synchronous loop against itself (float atomics make large steps differ run to run)
la, lb, lc = (np.array(r.history['loss']) for r in (ra, rb, rc))
assert len(la) == len(lb) == a.steps and ra.epoch == rb.epoch
k = min(a.steps, 200)
rel = lambda u, v, n: float(np.abs(u[:n] - v[:n]).max() / np.abs(v[:n]).max())      # noqa: E731
print('steps %d: feeder %.2f s, synchronous %.2f s' % (a.steps, ta, tb))
for n in (10, 50, k):
    print('first %3d losses: feeder vs synchronous %.2e   synchronous vs synchronous %.2e' % (n, rel(la, lb, n), rel(lc, lb, n)))
print('last-100 mean: feeder %.6f, synchronous %.6f / %.6f' % (la[-100:].mean(), lb[-100:].mean(), lc[-100:].mean()))
print('val_loss rows %d / %d, first %.6f vs %.6f, last %.6f vs %.6f (control %.6f)' % (
    len(ra.history['val_loss']), len(rb.history['val_loss']), ra.history['val_loss'][0], rb.history['val_loss'][0],
    ra.history['val_loss'][-1], rb.history['val_loss'][-1], rc.history['val_loss'][-1]))
print('final evaluation', dict(ea), dict(eb))
# the pipeline must not add to the run-to-run spread: within 10 x the control over the first steps (identical when the control is)
assert rel(la, lb, 10) <= max(10 * rel(lc, lb, 10), 1e-6), 'feeder diverges from the synchronous loop faster than the loop from itself'
print('ok')
