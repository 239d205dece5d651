#!/usr/bin/env python3
"""One-GPU rehearsal of the data-parallel path: with DNNCA_FORCE_RCCL=1 a one-rank RCCL communicator is created and
the gradient all-reduce, weight broadcast, state average and host all-reduce all go through RCCL.  A sum over one rank is
the identity, so every result must equal the run without a communicator up to the run-to-run noise of the float
atomics in the weight-gradient reduction.  Prints JSON."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dnncancerannotator_amd import device as dev                      # noqa: E402
from dnncancerannotator_amd.synthetic import synthetic_batch         # noqa: E402


def run(force):
    if force:
        os.environ['DNNCA_FORCE_RCCL'] = '1'
    else:
        os.environ.pop('DNNCA_FORCE_RCCL', None)
    m = dev.DeviceModel('unet', 1, 64, 64, 4, n_filters_first=4, n_downsample=2, rate=2, kernel_size=3, conv_stride=1,
                        bn=True, padding='same')
    m.init_glorot(seed=2)
    m.comm_init(0, 1, dev.DeviceModel.comm_unique_id() if force else None)
    m.comm_broadcast_weights(0)
    x, y = synthetic_batch(4, 64, 64, 1)
    losses = [float(m.train_step(x, y, 1e-3, m.loss_cfg(weight_mul=3.0)).loss) for _ in range(3)]
    # the same steps fed through the staging ring (copy stream + scalars one step late) with the communicator in the step
    ring, ring_losses, prev = m.staging(slots=2), [], None
    for i in range(3):
        px, py = ring.upload(i % 2, x, y)
        ring.train_step(i % 2, px, py, 4, 1e-3, m.loss_cfg(weight_mul=3.0))
        if prev is not None:
            ring_losses.append(float(ring.out(prev).loss))
        prev = i % 2
    ring_losses.append(float(ring.out(prev).loss))
    losses += ring_losses
    m.comm_average_state()
    red = m.comm_allreduce([1.5, -2.0], op='max').tolist()
    # metric counts of a 150-threshold AUC (600 counters, more than one staging chunk would be 2050) and counts beyond 2^24
    big = np.arange(5000, dtype=np.float64) * 7.0 + 2.0 ** 40 + 1.0
    red_big = bool(np.array_equal(m.comm_allreduce(big), big))
    out = dict(losses=losses, params=m.get_params(), state=m.get_state(), grads=m.get_grads(), red=red, red_big=red_big)
    m.close()
    return out


def run_big(force, bucket_bytes=None):
    """a 4 MB gradient vector (dense-channel kernels, BatchNorm): the bucketed all-reduce on the second stream"""
    if force:
        os.environ['DNNCA_FORCE_RCCL'] = '1'
    else:
        os.environ.pop('DNNCA_FORCE_RCCL', None)
    # (no BatchNorm: a deep BatchNorm network at this size amplifies the run-to-run rounding of the float atomics through
    #  ReLU / max-pool flips -- see profiles/r02_mask_flip_evidence.txt -- which would drown the comparison)
    m = dev.DeviceModel('unet', 1, 32, 32, 2, n_filters_first=32, n_downsample=3, rate=2, kernel_size=3, conv_stride=1,
                        bn=False, padding='same')
    m.init_glorot(seed=4)
    m.comm_init(0, 1, dev.DeviceModel.comm_unique_id() if force else None)
    x, y = synthetic_batch(2, 32, 32, 1)
    losses, calls = [], []
    for _ in range(3):
        losses.append(float(m.train_step(x, y, 1e-3, m.loss_cfg(weight_mul=3.0)).loss))
        calls.append(m.comm_collectives())
    out = dict(losses=losses, calls=calls, params=m.get_params(), grads=m.get_grads(), n=int(m.n_trainable))
    m.close()
    return out


def main():
    dev.init_device(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'big':
        # DNNCA_BUCKET_BYTES is read once per process: this process runs with the value the parent chose
        p, q, r = run_big(False), run_big(True), run_big(False)
        rel = lambda u, v: float(np.abs(u - v).max() / (np.abs(u).max() + 1e-30))       # noqa: E731
        print(json.dumps(dict(n=q['n'], calls=q['calls'], calls_plain=p['calls'], diff_params=rel(p['params'], q['params']),
                              diff_grads=rel(p['grads'], q['grads']), noise_params=rel(p['params'], r['params']),
                              noise_grads=rel(p['grads'], r['grads']), losses_plain=p['losses'], losses_rccl=q['losses'])))
        return
    a, b = run(False), run(True)
    c = run(False)       # run-to-run noise floor (weight gradients are accumulated with float atomics)
    diff = {k: float(np.abs(a[k] - b[k]).max() / (np.abs(a[k]).max() + 1e-30)) for k in ('params', 'state', 'grads')}
    noise = {k: float(np.abs(a[k] - c[k]).max() / (np.abs(a[k]).max() + 1e-30)) for k in ('params', 'state', 'grads')}
    print(json.dumps(dict(diff=diff, noise=noise, losses_plain=a['losses'], losses_rccl=b['losses'], red=b['red'], red_big=b['red_big'])))


if __name__ == '__main__':
    main()
