"""Which input seeds of tests/test_engine_gpu.py::test_dense_configs_at_real_widths_against_oracle are free of max-pool winner flips
for the arithmetic the library runs?  Per case and seed: tensors beyond 1e-4 of their own scale (+ 10 x float32-numpy noise), the
worst and the median per-tensor error.  The set of clean seeds changes with ANY float32 summation order (run with DNNCA_NO_X3=1 for
the exact-fp32 MFMA kernels); the NUMBER of clean seeds and the clean seeds' errors are what compares two arithmetics.
    python tools/seed_scan.py [first_seed] [n_seeds]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import helpers as Hp
from oracle import unet_oracle as O
from dnncancerannotator_amd import device

device.init_device(0)
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
CASES = [('unet', 1, dict(n_filters_first=64, n_downsample=4, bn=True), 1, 64),
         ('unet', 1, dict(n_filters_first=64, n_downsample=4, bn=True), 2, 64),
         ('mulmo', 3, dict(n_filters_first=16, n_downsample=4, bn=True), 2, 64),
         ('unet', 1, dict(n_filters_first=512, n_downsample=1, bn=True), 2, 32)]
print('mode:', 'exact fp32 MFMA' if os.environ.get('DNNCA_NO_X3') else 'split bf16 x3', flush=True)
for arch, C, opts, B, size in CASES:
    full = dict(rate=2, kernel_size=3, conv_stride=1, padding='same', **opts)
    alpha = 0.99
    spec = O.ModelSpec(arch, C, activation={'class_name': 'LeakyReLU', 'config': {'alpha': alpha}}, **full)
    params = Hp.perturbed_params(spec, np.float64)
    p32 = {k: v.astype(np.float32) for k, v in params.items()}
    m = device.DeviceModel(arch, C, size, size, B, leaky_alpha=alpha, **full)
    clean = []
    for seed in range(first, first + n):
        rng = np.random.default_rng(seed)
        x = rng.random((B, size, size, C)).astype(np.float32)
        y = (rng.random((B, size, size)) < 0.05).astype(np.float32)
        cfg = dict(weight_mul=3.0)
        loss, grads, _, state = O.loss_and_grads(spec, params, x.astype(np.float64), y, cfg, training=True)
        _, g32, _, _ = O.loss_and_grads(spec, p32, x, y, cfg, training=True)
        gref, g32 = O.flatten(spec, grads), O.flatten(spec, g32).astype(np.float64)
        m.set_params(O.flatten(spec, params))
        m.set_state(O.flatten(spec, params, trainable=False))
        m.train_step(x, y, 0.0, m.loss_cfg(**cfg))
        g = m.get_grads().astype(np.float64)
        errs, off, off32 = [], 0, 0
        for _, sl in Hp.tensor_slices(spec):
            s = np.abs(gref[sl]).max()
            fl = 10 * np.abs(g32[sl] - gref[sl]).max()
            d = np.abs(g[sl] - gref[sl]).max()
            errs.append(d / (s + 1e-300))
            off += not d <= 1e-4 * s + fl
            off32 += not np.abs(g32[sl] - gref[sl]).max() <= 1e-4 * s
        print('%-5s f0=%-3d B=%d seed %d: %2d tensors off (float32 numpy: %2d), max %.1e median %.1e' % (
            arch, opts['n_filters_first'], B, seed, off, off32, max(errs), np.median(errs)), flush=True)
        if not off:
            clean.append(seed)
    print('   clean seeds:', clean, flush=True)
    m.close()
