#!/usr/bin/env python3
"""e2e_rate.py -- train-step rate when every batch starts in HOST memory (what `annotator train` sees), next to the
device-resident rate of bench.py.  Four distinct synthetic batches are cycled so that no copy can be elided.

    python tools/e2e_rate.py [--batch 8] [--steps 300]

  resident     x, y already in HBM (bench.py's timed region)
  host-sync    dnnca_train_step(host pointers) + step outputs read every step (engine.train before the input pipeline)
  ring         StagingRing from one thread: upload on the copy stream, enqueue the step, read the scalars of the step before
  feeder       BatchFeeder: the uploads on a second host thread
  engine       TFKerasModel.train on a generator dataset (checkpoints / validation off; DNNCA_NO_FEEDER=1: the old loop)"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dnncancerannotator_amd import device as dev                # noqa: E402
from dnncancerannotator_amd.synthetic import synthetic_batch    # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--steps', type=int, default=300)
ap.add_argument('--size', type=int, default=512)
a = ap.parse_args()
B, S, N = a.batch, a.size, a.steps
dev.init_device(0)
m = dev.DeviceModel('unet', 1, S, S, B, n_filters_first=3, n_downsample=3, rate=2, kernel_size=3, conv_stride=1, bn=False, padding='same')
m.init_glorot(seed=2)
cfg = m.loss_cfg(weight_mul=3.0)
batches = [synthetic_batch(B, S, S, 1, seed_x=10 + i, seed_y=20 + i) for i in range(4)]
mb = sum(x.nbytes + y.nbytes for x, y in batches[:1]) / 1e6


def report(name, dt, extra=''):
    print('%-10s %8.3f ms/step  %9.1f slices/s %s' % (name, dt / N * 1e3, B * N / dt, extra), flush=True)


xb, yb = dev.DeviceBuffer(batches[0][0]), dev.DeviceBuffer(batches[0][1])
for _ in range(20):
    m.train_step_dev(xb, yb, B, 1e-3, cfg)
m.sync()
t0 = time.perf_counter()
for _ in range(N):
    m.train_step_dev(xb, yb, B, 1e-3, cfg)
m.sync()
report('resident', time.perf_counter() - t0)

for _ in range(5):
    m.train_step(batches[0][0], batches[0][1], 1e-3, cfg)
t0 = time.perf_counter()
for i in range(N):
    x, y = batches[i % 4]
    out = m.train_step(x, y, 1e-3, cfg)
dt = time.perf_counter() - t0
report('host-sync', dt, '(%.1f MB per step from pageable numpy arrays)' % mb)

ring = m.staging()
t0 = time.perf_counter()
prev = None
for i in range(N):
    x, y = batches[i % 4]
    slot = i % ring.slots
    px, py = ring.upload(slot, x, y, wait=False)        # (the arrays live on in `batches`)
    ring.train_step(slot, px, py, B, 1e-3, cfg)
    if prev is not None:
        out = ring.out(prev)
    prev = slot
out = ring.out(prev)
report('ring', time.perf_counter() - t0, '(one thread: upload, enqueue, read the step before; last loss %.6f)' % out.loss)

from dnncancerannotator_amd.feeder import BatchFeeder           # noqa: E402
feeder = BatchFeeder(m, iter(batches[i % 4] for i in range(N)), lambda a, b=None: (a, b))
t0 = time.perf_counter()
prev = None
for kind, slot, px, py, n in feeder:
    ring.train_step(slot, px, py, n, 1e-3, cfg)
    if prev is not None:
        out = ring.out(prev)
        feeder.release(prev)
    prev = slot
out = ring.out(prev)
report('feeder', time.perf_counter() - t0, '(uploads on a second thread; last loss %.6f)' % out.loss)
feeder.close()

from dnncancerannotator_amd.engine import TFKerasModel          # noqa: E402
m.close()
config = dict(model='UNetAnnotator',
              model_options=dict(n_filters_first=3, n_downsample=3, rate=2, kernel_size=3, conv_stride=1, bn=False, padding='same'),
              deploy_options=dict(optimizer='adam', loss=dict(class_name='WeightedCrossentropy', config=dict(weight_mul=3.0)),
                                  enable_multigpu=False))


class Gen:
    def __init__(self, n):
        self.n = n

    def __iter__(self):
        for i in range(self.n):
            yield batches[i % 4]


eng = TFKerasModel(config)
eng.train(Gen(20), max_steps=20, auto_resume=False)
t0 = time.perf_counter()
eng.train(Gen(N), max_steps=20 + N, auto_resume=False)
report('engine', time.perf_counter() - t0)


# `annotator evaluate`: keras Model.evaluate with the pixel metrics of configs/additionals/metrics.yaml (302 thresholds)
config['deploy_options']['metrics'] = [{'Precision': {'thresholds': 0.8, 'name': 'pixel/precision'}},
                                       {'Recall': {'thresholds': 0.8, 'name': 'pixel/recall'}},
                                       {'AUC': {'curve': 'PR', 'name': 'pixel/AUPRC', 'num_thresholds': 150}},
                                       {'AUC': {'curve': 'ROC', 'name': 'pixel/AUROC', 'num_thresholds': 150}},
                                       {'FBetaScore': {'thresholds': 0.8, 'beta': 1.0, 'name': 'pixel/F1-score'}},
                                       {'FBetaScore': {'thresholds': 0.8, 'beta': 2.0, 'name': 'pixel/F2-score'}}]
ev = TFKerasModel(config)
ev._build(Gen(1))
NE = max(N // 4, 8)
ev._evaluate(Gen(4), staged=True)
for staged in (False, True):
    t0 = time.perf_counter()
    r = ev._evaluate(Gen(NE), staged=staged)
    dt = time.perf_counter() - t0
    print('%-10s %8.3f ms/batch %9.1f slices/s (loss %.6f, F1 %.6f)' % ('eval' + ('-ring' if staged else '-sync'), dt / NE * 1e3, B * NE / dt,
                                                                      r['loss'], r['pixel/F1-score']), flush=True)
