"""Does the split-bf16 fp32 arithmetic (csrc/kernels_ig3x.hip) TRAIN like the fp32-MFMA kernels?  mulmo_unet's widths at an affordable
size (16 .. 64 channels x 3 encoders, BatchNorm, 64 x 64 x 3 images, batch 4), 300 Adam steps from the same initial weights on the same
stream of synthetic batches, once per arithmetic (child processes: the switch is read once per process):
    x3, x3_again     the default: every fp32 product of the dense 3x3 convs from six bf16 products (two runs: the float summation order
                     of ONE arithmetic already makes two runs drift -- that drift is the yardstick)
    f32mfma          DNNCA_NO_X3=1: v_mfma_f32_16x16x4_f32
Reported: training-loss curve, loss on held-out batches (batch statistics), Dice of the masks at 0.5 against the truth and between runs.
    python tests/x3_training_case.py [steps] > profiles/r04_x3_training.txt   (also run by tests/test_engine_gpu.py)"""
import json, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
OPTS = dict(n_filters_first=16, n_downsample=3, rate=2, kernel_size=3, conv_stride=1, bn=True, padding='same')
B, S, C = 4, 64, 3


def batch(seed):
    rng = np.random.default_rng(seed)
    y = np.zeros((B, S, S), np.float32)
    yy, xx = np.mgrid[0:S, 0:S]
    for b in range(B):
        for _ in range(int(rng.integers(1, 3))):
            r = rng.uniform(4, 11)
            cy, cx = rng.uniform(8, S - 8, 2)
            y[b][(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = 1.0
    x = np.stack([0.25 + a * y + 0.4 * rng.random((B, S, S)) for a in (0.35, 0.2, -0.15)], -1).astype(np.float32)   # three "modalities"
    return x, y


def dice(a, b):
    return float(2.0 * np.logical_and(a, b).sum() / max(1, a.sum() + b.sum()))


def child(steps, out):
    from dnncancerannotator_amd import device
    device.init_device(0)
    m = device.DeviceModel('mulmo', C, S, S, B, **OPTS)
    m.init_glorot(seed=5)
    cfg = m.loss_cfg(weight_mul=3.0)
    train_set = [batch(1000 + i) for i in range(16)]
    held_out = [batch(5000 + i) for i in range(8)]
    curve, last = [], []
    for step in range(steps):
        x, y = train_set[step % len(train_set)]
        o = m.train_step(x, y, 1e-3 if step < 2 * steps // 3 else 1e-4, cfg)
        if step % 10 == 0 or step == steps - 1:
            curve.append(round(float(o.loss), 5))
        if step >= steps - 48:
            last.append(float(o.loss))
    ev, masks, truth = [], [], []
    for x, y in held_out:
        masks.append(m.forward(x, training=True)[..., 0] > 0.5)
        ev.append(float(m.train_step(x, y, 0.0, cfg).loss))
        truth.append(y > 0.5)
    plan = sorted(set(r[0] for r in m.plan() if 'conv' in r[0] or 'wgrad' in r[0]))
    m.close()
    masks, truth = np.stack(masks), np.stack(truth)
    np.save(out + '.npy', masks)
    json.dump(dict(curve=curve, eval_loss=float(np.mean(ev)), tail=float(np.mean(last)), dice_truth=dice(masks, truth), plan=plan), open(out + '.json', 'w'))


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[1] == '--child':
        child(int(sys.argv[2]), sys.argv[3])
        sys.exit(0)
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    tmp = os.environ.get('TMPDIR', '/tmp')
    res, masks = {}, {}
    for name, env in (('x3', {}), ('x3_again', {}), ('f32mfma', {'DNNCA_NO_X3': '1'})):
        out = os.path.join(tmp, 'x3train_' + name)
        subprocess.run([sys.executable, os.path.abspath(__file__), '--child', str(steps), out], env=dict(os.environ, **env), check=True)
        res[name], masks[name] = json.load(open(out + '.json')), np.load(out + '.npy')
    print(__doc__.split('Reported')[0])
    for name, r in res.items():
        extra = '' if name == 'x3' else '  | vs x3: Dice of the masks %.4f, held-out loss %+.2f %%, training tail %+.2f %%' % (
            dice(masks[name], masks['x3']), 100 * (r['eval_loss'] / res['x3']['eval_loss'] - 1), 100 * (r['tail'] / res['x3']['tail'] - 1))
        print('%-9s held-out loss %.5f  mean training loss of the last 48 steps %.5f  Dice vs truth %.4f%s' % (name, r['eval_loss'], r['tail'], r['dice_truth'], extra))
        print('          kernels:', ' '.join(r['plan']))
        print('          training loss every 10th step:', ' '.join('%.4f' % v for v in r['curve']))
    print(json.dumps({n: dict(eval_loss=r['eval_loss'], tail=r['tail'], dice_truth=r['dice_truth'], dice_vs_x3=dice(masks[n], masks['x3']), plan=r['plan']) for n, r in res.items()}))
