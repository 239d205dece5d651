"""CPU: the numpy oracle against the committed golden vectors, against an independent torch-autograd statement, and
against hand-computed known answers.  (The reference holds no fixtures for this path: SURVEY.md 8c "parity unpinned".)"""

import numpy as np
import pytest

import helpers as Hp
from oracle import unet_oracle as O


@pytest.mark.parametrize('name', Hp.SMALL_CASES + Hp.BIG_CASES)
def test_oracle_reproduces_golden(name):
    z, spec, loss_cfg = Hp.load_case(name)
    p32, s32 = Hp.case_params(z, spec)
    params = O.unflatten(spec, p32.astype(np.float64))
    O.unflatten(spec, s32.astype(np.float64), trainable=False, into=params)
    x = z['x'].astype(np.float64)
    loss, new_params, grads, logits = O.train_step(spec, params, {}, {}, 1, x, z['y'], float(z['lr']), loss_cfg)
    assert abs(loss - float(z['loss_train'])) < 1e-9
    assert np.abs(logits - z['logits_train']).max() < 1e-5
    g = O.flatten(spec, grads)
    # every variable on its own scale; the fixture stores float32 (6e-8 relative), analytically-zero variables are 1e-17
    # noise in float64 and are held to the float32 rounding of the fixture's storage of that noise
    Hp.assert_grads_per_tensor(spec, g, z['grads'], 1e-6, floor=[1e-12] * len(Hp.tensor_slices(spec)))
    norms = np.array([np.sqrt((g[sl] ** 2).sum()) for _, sl in Hp.tensor_slices(spec)])
    assert np.allclose(norms, z['grad_norms'], rtol=1e-9, atol=1e-15)
    if 'params_after' in z.files:
        assert np.abs(O.flatten(spec, new_params) - z['params_after']).max() < 1e-6
    prob, logits_eval = O.predict(spec, params, x)
    assert np.abs(logits_eval - z['logits_eval']).max() < 1e-5
    assert np.array_equal(prob > 0.5, z['mask05'])


@pytest.mark.parametrize('kw, B, H, W, cfg', [
    (dict(arch='unet', in_channels=1, n_filters_first=3, n_downsample=2, bn=False, padding='same'), 2, 16, 16, dict(weight_mul=3.0)),
    (dict(arch='mulmo', in_channels=2, n_filters_first=3, n_downsample=2, bn=True, padding='same',
          activation={'class_name': 'LeakyReLU', 'config': {'alpha': 0.3}}), 2, 8, 8, dict(weight_mul=2.0, weight_add=0.25)),
])
def test_oracle_matches_torch_autograd(kw, B, H, W, cfg):
    torch_ref = pytest.importorskip('torch_ref')
    spec = O.ModelSpec(**kw)
    params = Hp.perturbed_params(spec, np.float64)
    x, y = O.synthetic_batch(B, H, W, spec.in_channels)
    x = x.astype(np.float64)
    loss, grads, logits, state = O.loss_and_grads(spec, params, x, y, cfg, training=True)
    ref = torch_ref.run(spec, params, x, y, cfg, training=True)
    assert abs(loss - ref['loss']) < 1e-10
    assert np.abs(logits - ref['logits']).max() < 1e-10
    for n, g in grads.items():
        assert np.abs(g - ref['grads'][n]).max() <= 1e-9 * np.abs(ref['grads'][n]).max() + 1e-13, n
    for n, s in ref['state'].items():
        assert np.abs(state[n] - s).max() < 1e-10, n


def test_valid_padding_forward_matches_torch():
    torch_ref = pytest.importorskip('torch_ref')
    spec = O.ModelSpec('unet', 1, 3, 2, bn=False, padding='valid')
    params = Hp.perturbed_params(spec, np.float64)
    x, _ = O.synthetic_batch(1, 44, 44, 1)
    logits, _, _, _ = O.forward(spec, params, x.astype(np.float64))
    assert logits.shape == (1, 20, 20, 1)            # real centre crop of the skips (components.py:161-163)
    ref = torch_ref.run(spec, params, x.astype(np.float64), np.zeros((1, 20, 20), np.float32), dict(weight_mul=1.0), False)
    assert np.abs(logits - ref['logits']).max() < 1e-10


def test_param_order_is_keras_creation_order():
    spec = O.ModelSpec('mulmo', 3, 16, 4, bn=True, padding='same')
    names = [n for n, _, _ in O.param_specs(spec)]
    assert names[0] == 'encoder0.down0.conv0.kernel' and names[1] == 'encoder0.down0.conv0.bias'
    assert names[2:6] == ['encoder0.down0.bn0.' + s for s in ('gamma', 'beta', 'moving_mean', 'moving_variance')]
    assert names.index('encoder2.down3.pool_bn.gamma') < names.index('decoder.up0.tconv.kernel') < names.index('head.kernel')
    assert sum(int(np.prod(s)) for _, s, t in O.param_specs(spec) if t) == 1713329
    assert sum(int(np.prod(s)) for _, s, t in O.param_specs(O.ModelSpec('unet', 1, 3, 3, padding='same')) if t) == 8686
    big = O.ModelSpec('unet', 1, 64, 4, bn=True, padding='same')
    assert sum(int(np.prod(s)) for _, s, t in O.param_specs(big) if t) == 15835713
    shapes = dict((n, s) for n, s, _ in O.param_specs(spec))
    assert shapes['decoder.up0.tconv.kernel'] == (2, 2, 128, 384)     # [kh, kw, Cout, Cin], bottleneck concat of 3 x 128


def test_loss_known_answers():
    # utils/losses.py:17-37 on a 1x1x2 "image": logits (0, 2), labels (1, 0), weight_mul 3
    y = np.array([[[1.0, 0.0]]], np.float32)
    logits = np.array([[[[0.0], [2.0]]]], np.float64)
    per, d = O.weighted_crossentropy(y, logits, weight_mul=3.0)
    w = 3.0 * (1 / 0.5)                                   # positive_rate 0.5
    expect = (np.log(2.0) * w + (2.0 + np.log1p(np.exp(-2.0))) * 1.0) / 2
    assert abs(per[0] - expect) < 1e-12
    assert abs(d[0, 0, 0, 0] - w * (0.5 - 1.0) / 2) < 1e-12
    # no positives -> weight 1 branch (losses.py:27); weight_mul still applies (losses.py:29)
    assert O.loss_weight(np.zeros((1, 4, 4), np.float32), weight_mul=3.0) == 3.0
    # fixed weight
    assert O.loss_weight(y, weight=5.0, weight_add=0.5, weight_mul=2.0) == 10.5
    # empty batch (losses.py:22-23)
    per, _ = O.weighted_crossentropy(np.zeros((0, 4, 4), np.float32), np.zeros((0, 4, 4, 1)))
    assert per.shape == (0,)
    with pytest.raises(AssertionError):                   # assert_on_max (losses.py:91)
        O.positive_rate(np.full((1, 2, 2), 1.5, np.float32))
    with pytest.raises(AssertionError):                   # assert_on_weight (losses.py:30)
        O.loss_weight(y, weight=1.0, weight_mul=-1.0)


def test_adam_and_schedule_known_answers():
    p = {'w': np.array([1.0, -2.0])}
    g = {'w': np.array([0.5, -0.25])}
    m, v = {}, {}
    out = O.adam_step(p, g, m, v, 1, 1e-3)
    # first Keras-Adam step: lr * sqrt(1-b2)/(1-b1) * (1-b1) g / (sqrt((1-b2) g^2) + eps) ~= lr * sign(g)
    assert np.allclose(out['w'], [1.0 - 1e-3, -2.0 + 1e-3], atol=1e-8)
    assert O.lr_schedule(0) == 0.001 and O.lr_schedule(999) == 0.001
    assert abs(O.lr_schedule(1000) - 0.00096) < 1e-12 and abs(O.lr_schedule(2500) - 0.001 * 0.96 ** 2) < 1e-12
    f = eval('lambda epoch, current_lr: 0.001 * 0.96 ** (epoch // 1000)')      # deploy_options.yaml:3 verbatim
    assert all(abs(f(s, None) - O.lr_schedule(s)) < 1e-15 for s in (0, 1, 999, 1000, 54321))


def test_layer_properties():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 8, 8, 3))
    w = rng.standard_normal((3, 3, 3, 4))
    b = rng.standard_normal(4)
    y1, _ = O.conv2d_fwd(x, w, b, 'same')
    y2, _ = O.conv2d_fwd(2 * x, w, 0 * b, 'same')
    y0, _ = O.conv2d_fwd(x, w, 0 * b, 'same')
    assert np.allclose(y2, 2 * y0) and np.allclose(y1, y0 + b)                 # linearity
    p, cache = O.maxpool_fwd(x, 2)
    assert p.shape == (2, 4, 4, 3) and np.all(p >= x[:, ::2, ::2, :])
    d = O.maxpool_bwd(cache, np.ones_like(p))
    assert d.sum() == p.size and set(np.unique(d)) <= {0.0, 1.0}               # exactly one winner per window
    t, _ = O.tconv_fwd(x, rng.standard_normal((2, 2, 5, 3)), np.zeros(5))
    assert t.shape == (2, 16, 16, 5)
    # BN train: zero mean / unit variance output for gamma 1, beta 0
    yb, _, (nm, nv) = O.bn_fwd(x, np.ones(3), np.zeros(3), np.zeros(3), np.ones(3), True)
    assert np.allclose(yb.mean((0, 1, 2)), 0, atol=1e-12) and np.allclose(yb.var((0, 1, 2)), x.var((0, 1, 2)) / (x.var((0, 1, 2)) + 1e-3))
    n = 2 * 8 * 8
    assert np.allclose(nv, 0.99 + 0.01 * x.var((0, 1, 2)) * n / (n - 1))        # unbiased variance into the moving average


def test_label_smoothing_kernel_and_padding():
    """tfa.image.gaussian_filter2d as used at utils/losses.py:64-66 (filter 6, sigma 3): the even filter's taps sit at
    u = -2 .. 3, the kernel sums to one (a constant image stays constant) and REFLECT padding does not repeat the edge."""
    const = np.full((1, 9, 11), 0.25)
    assert np.allclose(O.gaussian_filter2d(const), 0.25, atol=1e-15)
    u = np.arange(-2, 4)
    g = np.exp(-u ** 2 / 18.0)
    g /= g.sum()
    imp = np.zeros((1, 16, 16))
    imp[0, 8, 8] = 1.0
    out = O.gaussian_filter2d(imp)
    # out[y, x] = sum g[i] g[j] imp[y + i - 2, x + j - 2]  ->  the impulse response is g mirrored: out[8 - u, 8 - v] = g(u) g(v)
    for a, ua in enumerate(u):
        for b, ub in enumerate(u):
            assert abs(out[0, 8 - ua, 8 - ub] - g[a] * g[b]) < 1e-15
    ramp = np.arange(8.0)[None, None, :].repeat(4, 1)                           # reflect: index -1 -> 1, -2 -> 2
    left = O.gaussian_filter2d(ramp)[0, 0, 0]
    assert abs(left - sum(g[j] * abs(j - 2) for j in range(6))) < 1e-12
    logits = np.zeros((1, 8, 8, 1))
    y = np.zeros((1, 8, 8)); y[0, 2:5, 3:6] = 1.0
    per, _ = O.weighted_crossentropy(y, logits, label_smoothing=True)
    per0, _ = O.weighted_crossentropy(O.gaussian_filter2d(y), logits)
    assert np.allclose(per, per0)
