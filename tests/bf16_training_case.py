"""Does dtype bf16 (BASELINE configs[2]: bf16 MFMA operands, bf16-stored BatchNorm inputs / gradients) TRAIN like fp32?

unet_big's widths at a size the test suite can afford -- n_filters_first 64, two levels (64 .. 256 channels), BatchNorm on, 64 x 64
images, batch 4 -- trained for `steps` Adam steps from the same initial weights on the same stream of synthetic batches (discs on
a noisy background; the image carries the disc at half contrast, so there is something to learn), once per arithmetic:
    f32            the reference's arithmetic (all-fp32)
    bf16           dtype bf16 as the benchmark runs it (operands rounded to bf16, tensors whose readers round them stored as bf16,
                   BatchNorm inputs and the gradients arriving at a BatchNorm stored as bf16: ig_plan_half)
    bf16_f32act    dtype bf16 with DNNCA_NO_HALF_Z=1 DNNCA_NO_HALF_DY=1: only the operand rounding, activations stay fp32
Reported per run: the training-loss curve (every 10th step), the loss on held-out batches (batch statistics) and the masks at 0.5;
between runs: Dice of the masks.  Test infrastructure and a profiles/ generator (python tests/bf16_training_case.py > profiles/...).
"""
import json
import os
import sys

import numpy as np

OPTS = dict(n_filters_first=64, n_downsample=2, rate=2, kernel_size=3, conv_stride=1, bn=True, padding='same')
B, S = 4, 64


def batch(seed):
    rng = np.random.default_rng(seed)
    y = np.zeros((B, S, S), np.float32)
    yy, xx = np.mgrid[0:S, 0:S]
    for b in range(B):
        for _ in range(int(rng.integers(1, 3))):
            r = rng.uniform(4, 11)
            cy, cx = rng.uniform(8, S - 8, 2)
            y[b][(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = 1.0
    x = (0.25 + 0.35 * y + 0.4 * rng.random((B, S, S))).astype(np.float32)[..., None]       # disc at half contrast under noise
    return x, y


def dice(a, b):
    return float(2.0 * np.logical_and(a, b).sum() / max(1, a.sum() + b.sum()))


def train(device, dtype, steps, env=()):
    for k in ('DNNCA_NO_HALF_Z', 'DNNCA_NO_HALF_DY'):
        os.environ.pop(k, None)
    for k in env:
        os.environ[k] = '1'
    m = device.DeviceModel('unet', 1, S, S, B, dtype=dtype, **OPTS)
    for k in env:
        os.environ.pop(k, None)
    m.init_glorot(seed=5)
    cfg = m.loss_cfg(weight_mul=3.0)
    train_set = [batch(1000 + i) for i in range(16)]
    held_out = [batch(5000 + i) for i in range(8)]
    curve, last = [], []
    for step in range(steps):
        x, y = train_set[step % len(train_set)]
        lr = 1e-3 if step < 2 * steps // 3 else 1e-4          # the last third anneals: the runs settle instead of bouncing
        out = m.train_step(x, y, lr, cfg)
        if step % 10 == 0 or step == steps - 1:
            curve.append(round(float(out.loss), 5))
        if step >= steps - 3 * len(train_set):
            last.append(float(out.loss))                  # the last three passes over the training batches
    ev, masks, truth = [], [], []
    # held-out batches with BATCH statistics (training=True, learning rate 0): after a few hundred steps the moving statistics of
    # Keras' BatchNormalization (momentum 0.99, components.py:57) still lag the weights, whatever the arithmetic
    for x, y in held_out:
        masks.append(m.forward(x, training=True)[..., 0] > 0.5)
        ev.append(float(m.train_step(x, y, 0.0, cfg).loss))
        truth.append(y > 0.5)
    plan = sorted(set(r[0] for r in m.plan()))
    m.close()
    masks, truth = np.stack(masks), np.stack(truth)
    return dict(curve=curve, eval_loss=float(np.mean(ev)), train_loss_tail=float(np.mean(last)), dice_truth=dice(masks, truth), plan=plan), masks


def run(device, steps=300):
    out, masks = {}, {}
    # (f32_again: the same fp32 run a second time -- the float atomics of the weight gradients make two runs of ONE arithmetic drift
    #  apart as well; that drift is the yardstick for the bf16 differences)
    for name, dtype, env in (('f32', 'f32', ()), ('f32_again', 'f32', ()), ('bf16', 'bf16', ()),
                             ('bf16_f32act', 'bf16', ('DNNCA_NO_HALF_Z', 'DNNCA_NO_HALF_DY'))):
        out[name], masks[name] = train(device, dtype, steps, env)
    for name in ('f32_again', 'bf16', 'bf16_f32act'):
        out[name]['dice_vs_f32'] = dice(masks[name], masks['f32'])
        out[name]['eval_loss_rel'] = (out[name]['eval_loss'] - out['f32']['eval_loss']) / out['f32']['eval_loss']
        out[name]['train_tail_rel'] = (out[name]['train_loss_tail'] - out['f32']['train_loss_tail']) / out['f32']['train_loss_tail']
    return out


if __name__ == '__main__':
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from dnncancerannotator_amd import device as dev
    dev.init_device(0)
    res = run(dev, int(sys.argv[1]) if len(sys.argv) > 1 else 300)
    for name, r in res.items():
        print('%-12s held-out loss %.5f  mean training loss of the last 48 steps %.5f  Dice vs truth %.4f%s' % (
              name, r['eval_loss'], r['train_loss_tail'], r['dice_truth'],
              '' if name == 'f32' else '  | vs f32: Dice of the masks %.4f, held-out loss %+.2f %%, training tail %+.2f %%' % (
                  r['dice_vs_f32'], 100 * r['eval_loss_rel'], 100 * r['train_tail_rel'])))
        print('             training loss every 10th step:', ' '.join('%.4f' % v for v in r['curve']))
    print(json.dumps({n: {k: v for k, v in r.items() if k != 'plan'} for n, r in res.items()}))
