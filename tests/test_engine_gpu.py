"""-m gpu: the engine facade end to end (train / checkpoints / resume / evaluate / predict) and full-size properties."""

import os

import numpy as np
import pytest

import helpers as Hp
from oracle import unet_oracle as O

pytestmark = pytest.mark.gpu

FULL_TOL = 1e-4          # per tensor, 8 x 512 x 512: float32 sums of 2M terms in different orders
BF16_SMALL_COS = 0.7     # per tensor: measured min 0.775, median 0.90 (one 64 x 64 slice, 23 BatchNorm layers at random init); what
                         # this arithmetic does to TRAINING is test_bf16_trains_like_fp32's subject
DENSE_FULL_TOL = 0.15    # per tensor, deep BatchNorm networks at 512 x 512, tuned vs generic: two float32 paths flip different
                         # ReLU masks / pool winners (see test_reference_configs_against_oracle); measured <= 6.4e-2 on
                         # 6 of 90 tensors, median 3e-3; an all-zero / mis-indexed tensor is off by 1.0

UNET = dict(n_filters_first=3, n_downsample=3, rate=2, kernel_size=3, conv_stride=1, bn=False, padding='same')
CONFIG = {
    'model': 'UNetAnnotator', 'model_options': UNET,
    'deploy_options': {'optimizer': 'adam', 'LearningRateScheduler': 'lambda epoch, current_lr: 0.001 * 0.96 ** (epoch // 1000)',
                       'loss': {'class_name': 'WeightedCrossentropy', 'config': {'weight_mul': 3.0}}, 'enable_multigpu': False,
                       'metrics': [{'Precision': {'thresholds': 0.8, 'name': 'pixel/precision'}},
                                   {'FBetaScore': {'thresholds': 0.5, 'beta': 1.0, 'name': 'pixel/F1-score'}},
                                   {'AUC': {'curve': 'ROC', 'name': 'pixel/AUROC', 'num_thresholds': 50}}]},
}


def test_engine_train_checkpoint_resume_eval(gpu, tmp_path):
    from dnncancerannotator_amd import data, engine
    save = str(tmp_path / 'run')
    ds = data.SyntheticDataset(4, 64, 64, 1, n_batches=2, seed=3)
    m = engine.TFKerasModel(CONFIG)
    res = m.train(ds, save_path=save, max_steps=30, save_freq=10, val_data=data.SyntheticDataset(4, 64, 64, 1, n_batches=1, seed=9, repeat=False))
    assert res.epoch == list(range(30)) and len(res.history['loss']) == 30
    assert res.history['loss'][-1] < res.history['loss'][0]                 # it learns the synthetic discs
    assert list(m.get_ckpts(os.path.join(save, 'checkpoints')).keys()) == [10, 20, 30]
    assert 'val_loss' in res.history and len(res.history['val_loss']) == 3  # validation every save_freq steps
    w30 = m.device_model.get_params()

    # auto-resume: a fresh engine continues from ckpt-30 with the Adam slots and the step counter
    m2 = engine.TFKerasModel(CONFIG)
    res2 = m2.train(ds, save_path=save, max_steps=32, save_freq=10)
    assert res2.epoch == [30, 31] and m2.current_step == 32
    m3 = engine.TFKerasModel(CONFIG)
    m3._build(ds)
    m3.load(os.path.join(save, 'checkpoints', 'ckpt-30'))
    assert np.array_equal(m3.device_model.get_params(), w30)
    assert m3.device_model.get_opt_state()[2] == 30

    # evaluate every checkpoint in a step range, with pixel metrics computed from device-side confusion counts
    ev = data.SyntheticDataset(4, 64, 64, 1, n_batches=2, seed=11, repeat=False)
    rows = m3.eval(ev, save_path=save, tag='val', export_csv=True, step_range=(10, 20))
    assert list(rows.keys()) == [10, 20]
    assert set(rows[10]) == {'loss', 'pixel/precision', 'pixel/F1-score', 'pixel/AUROC'}
    assert os.path.exists(os.path.join(save, 'tfevents', 'val', 'results.csv'))
    with pytest.raises(ValueError):
        m3.eval(ev, save_path=save, tag='val')                              # tag already exists (engine.py:160-163)

    # predict == oracle forward with the trained weights
    x, _ = next(iter(ev))
    prob = m3.predict([(x,)])
    spec = O.ModelSpec('unet', 1, **UNET)
    params = O.unflatten(spec, m3.device_model.get_params())
    pref, _ = O.predict(spec, params, x)
    assert np.abs(prob - pref).max() < 1e-4


def test_label_assertion_and_batch_errors(gpu):
    m = gpu.DeviceModel('unet', 1, 32, 32, 2, **UNET)
    x = np.zeros((2, 32, 32, 1), np.float32)
    y = np.full((2, 32, 32), 1.5, np.float32)
    with pytest.raises(Exception) as e:                                     # assert_on_max (utils/losses.py:91)
        m.train_step(x, y, 1e-3, m.loss_cfg(weight_mul=3.0))
    assert 'label outside' in str(e.value)
    with pytest.raises(ValueError):
        m.forward(np.zeros((3, 32, 32, 1), np.float32))                     # more than max_batch
    with pytest.raises(ValueError):
        m.forward(np.zeros((2, 16, 16, 1), np.float32))                     # not the built input shape
    out = m.train_step(x, np.zeros((2, 32, 32), np.float32), 1e-3, m.loss_cfg(weight_mul=3.0))
    assert out.positive_rate == 0.0 and out.weight == 3.0                   # no positives: weight = mul * 1 + add
    m.close()
    with pytest.raises(Exception):
        gpu.DeviceModel('unet', 1, 30, 30, 2, **UNET)                       # 30 not divisible by 2^3
    with pytest.raises(Exception):
        gpu.DeviceModel('unet', 1, 32, 32, 2, **dict(UNET, padding='valid'))


def test_full_size_tuned_vs_generic_and_directional_derivative(gpu):
    """BASELINE size (8 x 512 x 512 x 1): the tuned MFMA path against the generic kernels, and the analytic gradient
    against a central finite difference of the loss along a random direction (size-independent property)."""
    from dnncancerannotator_amd.synthetic import synthetic_batch
    B, H, W = 8, 512, 512
    x, y = synthetic_batch(B, H, W, 1)
    tuned = gpu.DeviceModel('unet', 1, H, W, B, **UNET)
    generic = gpu.DeviceModel('unet', 1, H, W, B, force_generic=True, **UNET)
    tuned.init_glorot(seed=2)
    p0 = tuned.get_params()
    rng = np.random.default_rng(7)
    p0 = (p0 + rng.uniform(-0.05, 0.05, p0.shape)).astype(np.float32)      # non-zero biases
    tuned.set_params(p0)
    generic.set_params(p0)
    cfg = tuned.loss_cfg(weight_mul=3.0)
    pt, lt = tuned.forward(x, training=False, return_logits=True)
    pg, lg = generic.forward(x, training=False, return_logits=True)
    tol = 2e-4 * max(1.0, float(np.abs(lg).max()))
    assert np.abs(lt - lg).max() <= tol
    for thr in (0.5, 0.8):                                                  # Keras convention: prob > threshold
        decided = np.abs(lg - np.log(thr / (1 - thr))) > tol
        mt, mg = (pt > thr), (pg > thr)
        assert np.array_equal(mt[decided], mg[decided])
        inter = np.logical_and(mt, mg).sum()
        dice = 2.0 * inter / max(mt.sum() + mg.sum(), 1)
        assert dice > 0.9999 or mt.sum() + mg.sum() == 0
    ot = tuned.train_step(x, y, 0.0, cfg)                                   # lr 0: gradients without moving the weights
    og = generic.train_step(x, y, 0.0, cfg)
    assert abs(ot.loss - og.loss) <= 1e-5 * max(1.0, abs(og.loss))
    gt, gg = tuned.get_grads(), generic.get_grads()
    spec = O.ModelSpec('unet', 1, **UNET)
    # every variable on its own scale; 2M-term float32 sums in two different orders (generic: atomics; tuned: MFMA chains +
    # bucket slabs): measured <= 2e-5 per tensor, run-to-run spread of the generic path 1e-5 (profiles/r02_grad_spread.txt)
    Hp.assert_grads_per_tensor(spec, gt, gg, FULL_TOL, what='tuned vs generic')
    # directional derivative: (L(w + e d) - L(w - e d)) / 2e  ==  g . d
    d = rng.standard_normal(p0.shape).astype(np.float32)
    d /= np.linalg.norm(d)
    eps = 2e-3
    tuned.set_params(p0 + eps * d)
    lp = tuned.eval_step(x, y, cfg).loss
    tuned.set_params(p0 - eps * d)
    lm = tuned.eval_step(x, y, cfg).loss
    fd = (lp - lm) / (2 * eps)
    an = float(np.dot(gt.astype(np.float64), d.astype(np.float64)))
    assert abs(fd - an) <= 0.05 * max(abs(an), 1e-3), (fd, an)
    tuned.close()
    generic.close()


def test_full_size_label_statistics_ride_in_the_first_block(gpu):
    """At the BASELINE size the train step has no label-statistics launch and no head launch: the first encoder block's fused
    kernel reduces the labels (utils/losses.py:87-102) into a partials table and the conv that feeds the head runs head + loss +
    head backward in its epilogue.  The reference's assertions and the no-positives branch must still work through that path."""
    from dnncancerannotator_amd.synthetic import synthetic_batch
    B, H, W = 8, 512, 512
    x, y = synthetic_batch(B, H, W, 1)
    m = gpu.DeviceModel('unet', 1, H, W, B, **UNET)
    m.init_glorot(seed=2)
    names = [r[0] for r in m.plan()]
    assert 'tail3_3x1_3' in names and 'first3_fwd' in names and 'label_stats4' not in names and 'head_train_3' not in names
    cfg = m.loss_cfg(weight_mul=3.0, weight_add=0.25)
    out = m.train_step(x, y, 0.0, cfg)
    pr = float(y.astype(np.float64).mean())
    assert abs(out.positive_rate - pr) <= 1e-6 * pr and out.label_min == 0.0 and out.label_max == 1.0
    assert abs(out.weight - (3.0 / pr + 0.25)) <= 1e-4 * out.weight
    out0 = m.train_step(x, np.zeros_like(y), 0.0, cfg)                      # no positives: weight = mul * 1 + add (losses.py:27)
    assert out0.positive_rate == 0.0 and abs(out0.weight - 3.25) < 1e-6
    bad = y.copy()
    bad[5, 300, 17] = 1.5
    with pytest.raises(Exception) as e:                                     # assert_on_max (utils/losses.py:91)
        m.train_step(x, bad, 0.0, cfg)
    assert 'label outside' in str(e.value)
    ok = m.train_step(x, y, 0.0, cfg)                                       # and the model recovers
    assert abs(ok.loss - out.loss) <= 1e-6 * abs(out.loss)
    m.close()


def test_full_size_batch_permutation_and_determinism(gpu):
    """BASELINE size (8 x 512 x 512 x 1), size-independent properties of the tuned path: the forward pass is bit-reproducible;
    permuting the slices of the batch permutes the logits exactly (no BatchNorm in configs/unet.yaml: slices are independent)
    and leaves loss and gradients unchanged up to the summation order of the float atomics; the per-slice results do not
    depend on which tile / block / XCD processed them."""
    from dnncancerannotator_amd.synthetic import synthetic_batch
    B, H, W = 8, 512, 512
    x, y = synthetic_batch(B, H, W, 1)
    m = gpu.DeviceModel('unet', 1, H, W, B, **UNET)
    m.init_glorot(seed=2)
    cfg = m.loss_cfg(weight_mul=3.0)
    _, l0 = m.forward(x, training=False, return_logits=True)
    _, l1 = m.forward(x, training=False, return_logits=True)
    assert np.array_equal(l0, l1)
    perm = np.array([5, 2, 7, 0, 3, 6, 1, 4])
    _, lp = m.forward(np.ascontiguousarray(x[perm]), training=False, return_logits=True)
    assert np.array_equal(lp, l0[perm])
    o0 = m.train_step(x, y, 0.0, cfg)
    g0 = m.get_grads().copy()
    o1 = m.train_step(np.ascontiguousarray(x[perm]), np.ascontiguousarray(y[perm]), 0.0, cfg)
    g1 = m.get_grads()
    assert abs(o0.loss - o1.loss) <= 1e-6 * max(1.0, abs(o0.loss))
    Hp.assert_grads_per_tensor(O.ModelSpec('unet', 1, **UNET), g1, g0, FULL_TOL, what='permuted batch')
    m.close()


@pytest.mark.parametrize('arch, C, opts, B, size, seed_x', [
    ('unet', 1, dict(n_filters_first=64, n_downsample=4, bn=True), 1, 64, 3),        # configs/unet_big.yaml
    ('mulmo', 3, dict(n_filters_first=16, n_downsample=4, bn=True), 2, 64, 5),       # configs/mulmo_unet.yaml
])
def test_reference_configs_against_oracle(gpu, arch, C, opts, B, size, seed_x):
    """The real unet_big / mulmo_unet hyper-parameters against the float64 oracle on one small batch, ReLU at the Keras default
    initialisation.  (Input seeds: of seeds 0 .. 7 the worst tensor is off by 3e-4 .. 1.6e-1 on the device -- with the split-bf16
    conv kernels and with the exact-fp32 ones alike, each on different seeds -- see the comment below; these two sit at 4e-3 / 1e-4
    under both arithmetics.)"""
    full = dict(rate=2, kernel_size=3, conv_stride=1, padding='same', **opts)
    spec = O.ModelSpec(arch, C, **full)
    params = O.init_params(spec, seed=2)
    x, y = O.synthetic_batch(B, size, size, C, seed_x=seed_x)
    m = gpu.DeviceModel(arch, C, size, size, B, **full)
    m.set_params(O.flatten(spec, params))
    cfg = dict(weight_mul=3.0)
    out = m.train_step(x, y, 1e-3, m.loss_cfg(**cfg))
    p64 = {n: v.astype(np.float64) for n, v in params.items()}
    loss, grads, _, state = O.loss_and_grads(spec, p64, x.astype(np.float64), y, cfg, training=True)
    assert abs(out.loss - loss) <= 2e-4 * max(1.0, abs(loss))
    # Keras-default initialisation (zero biases, beta 0) on a deep BatchNorm network: any float32 implementation flips a few
    # ReLU masks / max-pool winners against float64 (pre-activations within rounding of zero), and one flip at a level with N
    # positions moves that layer's gradient -- and everything upstream of it -- by O(1/N): 32 positions in the deepest
    # BatchNorm here.  Plain float32 numpy shows the same jumps on the same inputs (profiles/r02_mask_flip_evidence.txt:
    # 1e-3 on ~140 of 170 tensors for 3 of 8 seeds, the device 1e-3 .. 8e-2 for 4 of 8).  So: every variable within 15 % of
    # its own scale (an all-zero or mis-indexed gradient is off by 100 %), half of them within 2e-3.  The flip-free, tight
    # per-tensor bound on these hyper-parameters is tests/test_parity_gpu.py (mulmo_yaml_2x64, unet_big_f8_2x64).
    gref = O.flatten(spec, grads)
    errs = Hp.assert_grads_per_tensor_nofixture(spec, m.get_grads(), gref, 0.15)
    healthy = [e for n, e in errs.items() if n not in Hp.degenerate_tensors(spec)]
    assert np.median(healthy) <= 2e-3, np.median(healthy)
    assert np.abs(m.get_state() - O.flatten(spec, dict(p64, **state), trainable=False)).max() <= 1e-4
    m.close()


@pytest.mark.parametrize('arch, C, opts, B, size, seed', [
    ('unet', 1, dict(n_filters_first=64, n_downsample=4, bn=True), 1, 64, 121),      # configs/unet_big.yaml: 64 .. 1024 channels
    ('unet', 1, dict(n_filters_first=64, n_downsample=4, bn=True), 2, 64, 121),
    ('mulmo', 3, dict(n_filters_first=16, n_downsample=4, bn=True), 2, 64, 121),     # configs/mulmo_unet.yaml: 3 x (16 .. 128) + 384
    ('unet', 1, dict(n_filters_first=512, n_downsample=1, bn=True), 2, 32, 121),     # one level, 512 -> 512 and 1024 -> 512 channels
])
def test_dense_configs_at_real_widths_against_oracle(gpu, arch, C, opts, B, size, seed):
    """The dense fp32 kernels (k_ig_conv3, k_ig_wgrad2, k_ig_tconv_*, k_first_*, the tuned BatchNorm and pooling passes) at the REAL
    widths of configs/unet_big.yaml and configs/mulmo_unet.yaml -- K loops over up to 1 024 input channels -- against the float64
    oracle: every variable within 1e-4 of its own scale (+ 10 x what plain float32 numpy costs on that variable).

    How the flip lottery is kept out (measured on MI355X with the seed scan this test's inputs come from): with ReLU, one
    pre-activation that float32 and float64 put on different sides of zero changes a BatchNorm channel's batch statistics and
    with them EVERY tensor by O(1 / positions of that level) -- the device and plain float32 numpy both show medians of 1e-3 on
    most random inputs at these widths, each on different ones.  So the activation here is LeakyReLU(0.99): the kernels run the
    same code (act = v > 0 ? v : alpha v, act' = y > 0 ? 1 : alpha, the masks read the same pixels -- a mis-indexed mask is still
    off by 0.5 %, fifty times the bound) but a sign flip moves a derivative by 1 %, not 100 %.  What remains are max-pool winner
    flips, independent of alpha: input seed 121 has none (others fail on a handful of tensors by 1e-4 .. 3e-2, in float32
    numpy just as often; any change of the arithmetic order reshuffles which.  Round 4, tools/seed_scan.py, profiles/r04_x3_seed_scan.txt:
    of seeds 100 .. 123 the split-bf16 conv kernels of kernels_ig3x.hip are clean on 9 / 9 / 21 / 15 for the four cases, the exact-fp32
    MFMA kernels on 14 / 9 / 23 / 15, on different seeds; 121 is clean for both in all four cases.  The seed was 108 until the
    transposed convs' batch statistics moved into their own epilogue (k_ig_tconv_fwd2: another summation order, another lottery).)
    test_reference_configs_against_oracle keeps ReLU at the Keras default initialisation, loosely."""
    full = dict(rate=2, kernel_size=3, conv_stride=1, padding='same', **opts)
    alpha = 0.99
    spec = O.ModelSpec(arch, C, activation={'class_name': 'LeakyReLU', 'config': {'alpha': alpha}}, **full)
    params = Hp.perturbed_params(spec, np.float64)
    rng = np.random.default_rng(seed)
    x = rng.random((B, size, size, C)).astype(np.float32)
    y = (rng.random((B, size, size)) < 0.05).astype(np.float32)
    cfg = dict(weight_mul=3.0)
    loss, grads, _, state = O.loss_and_grads(spec, params, x.astype(np.float64), y, cfg, training=True)
    p32 = {n: v.astype(np.float32) for n, v in params.items()}
    _, g32, _, _ = O.loss_and_grads(spec, p32, x, y, cfg, training=True)
    gref, g32 = O.flatten(spec, grads), O.flatten(spec, g32).astype(np.float64)
    floor = [10 * np.abs(g32[sl] - gref[sl]).max() for _, sl in Hp.tensor_slices(spec)]
    m = gpu.DeviceModel(arch, C, size, size, B, leaky_alpha=alpha, **full)
    m.set_params(O.flatten(spec, params))
    m.set_state(O.flatten(spec, params, trainable=False))
    out = m.train_step(x, y, 0.0, m.loss_cfg(**cfg))
    assert abs(out.loss - loss) <= 1e-5 * max(1.0, abs(loss))
    errs = Hp.assert_grads_per_tensor(spec, m.get_grads(), gref, 1e-4, floor=floor)
    assert np.median(list(errs.values())) <= 2e-5, np.median(list(errs.values()))       # measured 2.4e-6 .. 6.6e-6
    assert np.abs(m.get_state() - O.flatten(spec, dict(params, **state), trainable=False)).max() <= 1e-5
    launches = [r[0] for r in m.plan()]
    plan = set(launches)
    assert any(k.startswith('ig3x_conv') for k in plan) and any(k.startswith('ig3x_wgrad') for k in plan), plan
    # the BatchNorms whose every reader is a 3x3 conv have no apply pass: the convs (k_ig_conv3 forward, k_ig_wgrad2) read the
    # BatchNorm's input and apply scale / shift while they stage it (Op::elided) -- this comparison is what pins that path
    n_bn = sum(1 for n, _ in Hp.tensor_slices(spec) if n.endswith('.gamma'))
    n_apply = launches.count('bn_apply') + launches.count('bn_apply_pool')
    assert n_apply <= n_bn // 2, (n_apply, n_bn)
    # inference at the same widths (engine.py:198-203: evaluate runs training=False): the elision is decided regardless of `training`,
    # so the convs then apply the MOVING-statistics coefficients while they stage their input.  The step above (learning rate 0) has
    # left the weights alone and updated the moving statistics: logits against oracle.predict on exactly that state, masks at 0.5 /
    # 0.8 bit-exact wherever the oracle logit is farther from the threshold than the logit tolerance; the launches of that forward
    # pass (HIP-event profile) must show the elided apply passes missing.
    p_after = dict(params, **state)
    prob_ref, logit_ref = O.predict(spec, p_after, x.astype(np.float64))
    m.profile_reset()
    m.profile_enable(1)
    prob, logits = m.forward(x, training=False, return_logits=True)
    eval_launches = {r[0]: r[1] for r in m.profile()}
    m.profile_enable(0)
    tol = 2e-4 * max(1.0, float(np.abs(logit_ref).max()))
    assert np.abs(logits - logit_ref).max() <= tol, np.abs(logits - logit_ref).max()
    assert np.abs(prob - prob_ref).max() <= tol
    for thr in (0.5, 0.8):
        decided = np.abs(logit_ref - np.log(thr / (1 - thr))) > tol
        assert np.array_equal((prob > thr)[decided], (prob_ref > thr)[decided]), 'mask flip away from the threshold'
    n_apply_eval = eval_launches.get('bn_apply', 0) + eval_launches.get('bn_apply_pool', 0)
    assert any(k.startswith('ig3x_conv_fwd') for k in eval_launches) and n_apply_eval <= n_bn // 2, eval_launches
    Hp.record_oracle_plan(m, 'test_dense_configs_at_real_widths_against_oracle')
    m.close()


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_weight_gradient_slab_folds_per_launch_and_batched(gpu, monkeypatch, dtype):
    """The atomics-free weight gradients (WgArgs::plain: every block stores its sums into its own slab) reach the gradient vector
    through one fold launch behind each weight-gradient launch (default) or through ONE fold launch for the whole backward pass
    (DNNCA_FOLD_BATCH=1, opt-in: k_wg_fold_batch finds its segment by a ballot over the segment table).  Same sums in the same order
    either way: the gradients must agree to the noise of the BatchNorm statistics' double atomics and the transposed convs' float
    atomics (1e-5 of each tensor's scale); the default arm is compared with the float64 oracle by the dense tests."""
    opts = dict(n_filters_first=64, n_downsample=2, rate=2, kernel_size=3, conv_stride=1, bn=True, padding='same')
    B, S = 2, 64
    spec = O.ModelSpec('unet', 1, **opts)
    rng = np.random.default_rng(11)
    x = rng.random((B, S, S, 1)).astype(np.float32)
    y = (rng.random((B, S, S)) < 0.05).astype(np.float32)
    grads, plans = {}, {}
    for arm in ('per_launch', 'batched'):
        monkeypatch.delenv('DNNCA_FOLD_BATCH', raising=False)
        if arm == 'batched':
            monkeypatch.setenv('DNNCA_FOLD_BATCH', '1')
        m = gpu.DeviceModel('unet', 1, S, S, B, dtype=dtype, **opts)
        m.init_glorot(seed=4)
        cfg = m.loss_cfg(weight_mul=3.0)
        m.train_step(x, y, 0.0, cfg)
        m.train_step(x, y, 0.0, cfg)          # the second step reuses the device-side segment table of the first
        grads[arm] = m.get_grads().astype(np.float64)
        plans[arm] = [r[0] for r in m.plan()]
        m.close()
    assert plans['per_launch'].count('wg_fold') >= 4 and 'wg_fold_all' not in plans['per_launch'], plans['per_launch']
    assert plans['batched'].count('wg_fold_all') == 1 and 'wg_fold' not in plans['batched'], plans['batched']
    ref = grads['per_launch']
    assert np.isfinite(ref).all()
    for name, sl in Hp.tensor_slices(spec):
        if name in Hp.degenerate_tensors(spec):          # (biases in front of a BatchNorm: gradient = rounding noise around zero)
            continue
        scale = np.abs(ref[sl]).max()
        err = np.abs(grads['batched'][sl] - ref[sl]).max()
        assert scale > 0 and err <= 1e-5 * scale, (name, err / scale)


def test_batchnorm_reductions_fold_themselves_repeatably(gpu):
    """The BatchNorm reductions add their partial sums to a small table with double atomics and the last block of the same launch
    folds it and leaves it zeroed (csrc/bn_dev.h).  A hundred steps at learning rate 0 on one model: every step finds the table zeroed (a
    leftover would show up in the very next BatchNorm's statistics), so every step reproduces the first one's loss, BatchNorm
    gradients and batch statistics up to rounding -- the order the blocks arrive in moves a double sum of partials by double
    rounding only (~1e-16 relative; what moves visibly is the float32 weight-gradient atomics of the convs).  At this size every
    BatchNorm pass has <= 16 blocks: one member per ticket group, one adder per bucket row; the many-block case is the next test."""
    opts = dict(n_filters_first=64, n_downsample=2, rate=2, kernel_size=3, conv_stride=1, bn=True, padding='same')
    B, S = 2, 64
    spec = O.ModelSpec('unet', 1, **opts)
    rng = np.random.default_rng(5)
    x = rng.random((B, S, S, 1)).astype(np.float32)
    y = (rng.random((B, S, S)) < 0.05).astype(np.float32)
    for dtype in ('f32', 'bf16'):
        m = gpu.DeviceModel('unet', 1, S, S, B, dtype=dtype, **opts)
        m.init_glorot(seed=4)
        s0 = m.get_state()
        cfg = m.loss_cfg(weight_mul=3.0)
        first = None
        for step in range(100):
            m.set_state(s0)
            out = m.train_step(x, y, 0.0, cfg)
            g, st = m.get_grads().astype(np.float64), m.get_state().astype(np.float64)
            bn = np.concatenate([g[sl] for n, sl in Hp.tensor_slices(spec) if n.endswith('.gamma') or n.endswith('.beta')])
            if first is None:
                first = (out.loss, bn, st)
                assert np.isfinite(bn).all() and np.abs(bn).max() > 0
                continue
            assert abs(out.loss - first[0]) <= 1e-6 * max(1.0, abs(first[0])), (dtype, step)
            assert np.abs(st - first[2]).max() <= 1e-6 * np.abs(first[2]).max(), (dtype, step)          # moving statistics
            assert np.abs(bn - first[1]).max() <= 1e-4 * np.abs(first[1]).max(), (dtype, step)          # (their inputs carry the conv atomics)
        assert 'bn_bwd_reduce' in set(r[0] for r in m.plan())
        m.close()


def _bn_table(m):
    """(sum |bucket rows|, sum of ticket counters, allocated?) of the self-folding BatchNorm table (debug_tools.hip; synchronises)"""
    import ctypes as C
    fn = m.lib.dnnca_debug_bn_table
    fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint), C.POINTER(C.c_int)]
    a, t, al = C.c_double(-1.0), C.c_uint(99), C.c_int(0)
    assert fn(m.handle, C.byref(a), C.byref(t), C.byref(al)) == 0
    return a.value, t.value, bool(al.value)


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_batchnorm_self_fold_with_hundreds_of_blocks(gpu, dtype):
    """The two-level ticket of csrc/bn_dev.h as configs/unet_big.yaml and configs/mulmo_unet.yaml run it: 2 x 512 x 512 x 64 gives
    every stand-alone BatchNorm pass 512 blocks (16 members per ticket group, 32 adders per bucket row) and the conv epilogues one
    block per CU, all finishing together.  (i) After every step the table and all ticket counters read back as ZERO (a block that
    drew the last ticket before somebody's add had landed would leave that add behind).  (ii) The first BatchNorm's batch statistics
    -- Conv2D(1 -> 64) + ReLU computed here in float64 numpy, mean and unbiased variance per channel (components.py:46-58, Keras
    BatchNormalization's moving-statistics update with momentum 0.99) -- match to 2e-6.  (iii) Twenty steps at learning rate 0
    reproduce the first one's loss, moving statistics and BatchNorm gradients."""
    opts = dict(n_filters_first=64, n_downsample=1, rate=2, kernel_size=3, conv_stride=1, bn=True, padding='same')
    B, S = 2, 512
    spec = O.ModelSpec('unet', 1, **opts)
    rng = np.random.default_rng(11)
    x = rng.random((B, S, S, 1)).astype(np.float32)
    y = (rng.random((B, S, S)) < 0.02).astype(np.float32)
    m = gpu.DeviceModel('unet', 1, S, S, B, dtype=dtype, **opts)
    m.init_glorot(seed=6)
    named = m.named_params()
    s0 = np.zeros_like(m.get_state())          # moving statistics start at 0: afterwards they are 0.01 x the batch statistics
    cfg = m.loss_cfg(weight_mul=3.0)
    # (ii) the reference for the first BatchNorm, float64
    w = named['encoder.down0.conv0.kernel'].astype(np.float64).reshape(9, 64)
    b = named['encoder.down0.conv0.bias'].astype(np.float64)
    xp = np.pad(x[..., 0].astype(np.float64), ((0, 0), (1, 1), (1, 1)))
    z = np.zeros((B, S, S, 64))
    for t in range(9):
        z += xp[:, t // 3:t // 3 + S, t % 3:t % 3 + S, None] * w[t]
    z = np.maximum(z + b, 0.0).reshape(-1, 64)
    n = z.shape[0]
    mean_ref, var_ref = z.mean(0), z.var(0) * n / (n - 1)
    del z
    st_slices = dict(Hp.tensor_slices(spec, trainable=False))
    first = None
    for step in range(20):
        m.set_state(s0)
        out = m.train_step(x, y, 0.0, cfg)
        leftover, tickets, allocated = _bn_table(m)
        assert allocated and leftover == 0.0 and tickets == 0, (dtype, step, leftover, tickets)
        g, st = m.get_grads().astype(np.float64), m.get_state().astype(np.float64)
        bn = np.concatenate([g[sl] for n_, sl in Hp.tensor_slices(spec) if n_.endswith('.gamma') or n_.endswith('.beta')])
        if first is None:
            first = (out.loss, bn, st)
            assert np.isfinite(bn).all() and np.abs(bn).max() > 0
            mm, mv = st[st_slices['encoder.down0.bn0.moving_mean']] / 0.01, st[st_slices['encoder.down0.bn0.moving_variance']] / 0.01
            assert np.abs(mm - mean_ref).max() <= 2e-6 * np.abs(mean_ref).max(), np.abs(mm - mean_ref).max()
            assert np.abs(mv - var_ref).max() <= 2e-6 * np.abs(var_ref).max(), np.abs(mv - var_ref).max()
            continue
        assert abs(out.loss - first[0]) <= 1e-6 * max(1.0, abs(first[0])), (dtype, step)
        assert np.abs(st - first[2]).max() <= 1e-6 * np.abs(first[2]).max(), (dtype, step)
        assert np.abs(bn - first[1]).max() <= 2e-4 * np.abs(first[1]).max(), (dtype, step)          # (their inputs carry the conv atomics)
    plan = set(r[0] for r in m.plan())
    assert 'bn_bwd_reduce' in plan and ('bn_apply_pool' in plan or 'bn_stats' in plan), plan
    m.close()


@pytest.mark.parametrize('x3', [1, 0])
@pytest.mark.parametrize('alpha', [0.0, 0.99])
def test_fp32_eight_wave_conv_kernels_against_oracle(gpu, alpha, x3):
    """The eight-wave fp32 conv kernels of the 16- and 32-channel levels, forward and data gradient -- what configs/mulmo_unet.yaml
    runs at 8 x 512 x 512 (the launchers pick them where a layer has >= 256 units of 32 x 16 pixels) and what no small shape selects:
    the kernel-coverage test found them missing.  x3 = 1: ig3x::k_ig3x_conv3<1|2, MODE, 8>, the split-bf16 kernels the benchmark
    runs (kernels_ig3x.hip: fp32 operands as three bf16 planes on the bf16 matrix pipe, fp32-accurate); x3 = 0 (DNNCA_NO_X3=1):
    ig::k_ig_conv3<1|2, MODE, 8> on the fp32 matrix pipe, the exact-fp32 twin.  A mulmo network of those widths on a 2 x 40 x 48
    batch in a child process with DNNCA_IG_NW=8 (partial tiles in both directions), against the float64 oracle: every variable within
    2e-5 of its own scale + 10 x the float32-numpy noise -- the SAME bound for both (ReLU, perturbed weights; LeakyReLU(0.99) as the
    flip-free companion, see test_dense_configs_at_real_widths_against_oracle)."""
    import json
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    # x3: three levels -- 16-, 32- and 64-channel tiles all have an eight-wave variant; exact fp32: two (its 64-channel tile has none).
    # (Seed 31 has a max-pool winner flip under LeakyReLU; so has seed 32 for the exact-fp32 kernels on three levels.)
    case = dict(arch='mulmo', C=3, opts=dict(n_filters_first=16, n_downsample=3 if x3 else 2, bn=True), B=2, H=40, W=48, alpha=alpha, seed=32)
    env = dict(os.environ, DNNCA_IG_NW='8')
    if not x3:
        env['DNNCA_NO_X3'] = '1'
    r = subprocess.run([sys.executable, os.path.join(here, 'oracle_case.py'), json.dumps(case)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    o = json.loads(r.stdout.strip().splitlines()[-1])
    assert abs(o['loss'] - o['loss_ref']) <= 1e-4 * max(1.0, abs(o['loss_ref']))
    bad = {n: e for n, e in o['errs'].items() if not e <= (2e-5 if not alpha else 1e-4) + o['floors'][n]}
    assert not bad, bad
    if x3:
        # (16-channel tiles on eight waves run the double-buffered variant: x3n1w8d)
        want = {'ig3x_conv_fwd#x3n1w8d', 'ig3x_conv_fwd#x3n2w8', 'ig3x_conv_fwd#x3n4w8', 'ig3x_conv_dgrad#x3n1w8d', 'ig3x_conv_dgrad#x3n2w8', 'ig3x_conv_dgrad#x3n4w8'}
    else:
        want = {'ig_conv_fwd#3n1w8', 'ig_conv_fwd#3n2w8', 'ig_conv_dgrad#3n1w8', 'ig_conv_dgrad#3n2w8'}
    assert want <= set(o['plan']), o['plan']
    Hp.record_oracle_plan(set(o['plan']) | set(k.split('#')[0] for k in o['plan']), 'test_fp32_eight_wave_conv_kernels_against_oracle')


def test_split_bf16_trains_like_fp32_mfma(gpu):
    """300 Adam steps of a mulmo network (3 encoders, 16 .. 64 channels, BatchNorm) from the same weights on the same batches: the default
    arithmetic (fp32 products from six bf16 products, kernels_ig3x.hip) twice, and the fp32-MFMA kernels (DNNCA_NO_X3=1) once, in child
    processes (tests/x3_training_case.py; profiles/r04_x3_training.txt).  A property, not parity: the two arithmetics must differ no
    more than two runs of ONE arithmetic do (measured: Dice of the masks 0.975 vs 0.973, held-out loss +0.07 % vs +0.04 %)."""
    import json
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, 'x3_training_case.py'), '300'], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    o = json.loads(r.stdout.strip().splitlines()[-1])
    assert any(k.startswith('ig3x_conv') for k in o['x3']['plan']) and not any(k.startswith('ig3x') for k in o['f32mfma']['plan']), o
    assert o['x3']['dice_truth'] >= 0.75 and o['f32mfma']['dice_truth'] >= 0.75          # both learn the task
    yard = 1.0 - o['x3_again']['dice_vs_x3']                                              # one arithmetic's own run-to-run drift
    assert 1.0 - o['f32mfma']['dice_vs_x3'] <= max(2.0 * yard, 0.05), (o['f32mfma']['dice_vs_x3'], o['x3_again']['dice_vs_x3'])
    assert abs(o['f32mfma']['eval_loss'] / o['x3']['eval_loss'] - 1.0) <= 0.02
    assert abs(o['f32mfma']['tail'] / o['x3']['tail'] - 1.0) <= 0.02


def test_exact_fp32_conv_kernels_behind_the_switch(gpu):
    """DNNCA_NO_X3=1 takes the 3x3 convs of the fp32 dense path back to ig::k_ig_conv3 on the fp32 matrix pipe (the kernels the
    split-bf16 ones of kernels_ig3x.hip replaced in round 4; kept as the exact-fp32 twin): a 64 / 128-channel level in a child
    process against the float64 oracle, the same bound as the default path (1e-4 per tensor + float32-numpy noise, LeakyReLU(0.99))."""
    import json
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    case = dict(arch='unet', C=1, opts=dict(n_filters_first=64, n_downsample=1, bn=True), B=2, H=32, W=32, alpha=0.99, seed=108)
    r = subprocess.run([sys.executable, os.path.join(here, 'oracle_case.py'), json.dumps(case)], env=dict(os.environ, DNNCA_NO_X3='1'),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    o = json.loads(r.stdout.strip().splitlines()[-1])
    assert abs(o['loss'] - o['loss_ref']) <= 1e-4 * max(1.0, abs(o['loss_ref']))
    bad = {n: e for n, e in o['errs'].items() if not e <= 1e-4 + o['floors'][n]}
    assert not bad, bad
    assert {'ig_conv_fwd#3n4w4', 'ig_conv_dgrad#3n4w4'} <= set(o['plan']) and not any(k.startswith('ig3x') for k in o['plan']), o['plan']
    Hp.record_oracle_plan(set(o['plan']) | set(k.split('#')[0] for k in o['plan']), 'test_exact_fp32_conv_kernels_behind_the_switch')


def test_first_generation_transposed_conv_forward_behind_the_switch(gpu):
    """DNNCA_TCONV_FWD1=1 takes the Conv2DTranspose forward back to the first-generation kernels (one block per output parity, no
    fused statistics: ig::k_ig_tconv_fwd -- still the fallback for outputs beyond 2^31 elements): a BatchNorm decoder level in a child
    process against the float64 oracle, the same bound as the default path."""
    import json
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    case = dict(arch='unet', C=1, opts=dict(n_filters_first=32, n_downsample=2, bn=True), B=2, H=32, W=48, alpha=0.99, seed=121)
    r = subprocess.run([sys.executable, os.path.join(here, 'oracle_case.py'), json.dumps(case)], env=dict(os.environ, DNNCA_TCONV_FWD1='1'),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    o = json.loads(r.stdout.strip().splitlines()[-1])
    assert abs(o['loss'] - o['loss_ref']) <= 1e-4 * max(1.0, abs(o['loss_ref']))
    bad = {n: e for n, e in o['errs'].items() if not e <= 1e-4 + o['floors'][n]}
    assert not bad, bad
    assert any(k.startswith('ig_tconv_fwd#n') for k in o['plan']) and 'bn_stats' in set(k.split('#')[0] for k in o['plan']), o['plan']
    Hp.record_oracle_plan(set(o['plan']) | set(k.split('#')[0] for k in o['plan']), 'test_first_generation_transposed_conv_forward_behind_the_switch')


def _per_tensor_cosine(spec, g, gref):
    out = {}
    for n, sl in Hp.tensor_slices(spec):
        a, b = np.asarray(g[sl], np.float64), np.asarray(gref[sl], np.float64)
        out[n] = float(np.dot(a, b) / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))
    return out


@pytest.mark.parametrize('nw', [4, 8])
@pytest.mark.parametrize('f0,S,B,bn,n_down', [(64, 24, 2, 0, 2), (64, 16, 2, 0, 2), (32, 24, 2, 0, 2), (64, 24, 2, 1, 1), (64, 16, 2, 1, 1),
                                              (64, 16, 2, 1, 2), (32, 24, 2, 1, 2)])
def test_bf16_kernels_against_bf16_emulating_oracle(gpu, f0, S, B, bn, n_down, nw):
    """dtype bf16 (BASELINE configs[2]): the implicit-GEMM kernels round their operands to bf16 while staging and accumulate in
    fp32; the oracle is made to do exactly that (tests/bf16_emul_case.py), so the comparison isolates the kernels' indexing
    from bf16 noise.  Runs in a child process with DNNCA_IGB_NW = 4 / 8: NW = 8 (eight waves, 32 x 16-pixel tiles) is the
    k_igb_conv3 variant the unet_big benchmark runs and that no small shape selects by itself; S = 24 / 16 leave partial
    tiles in both directions for both variants; the BatchNorm case runs the bf16-STORED operand variants (_a16, wgrad64w).

    Tolerance.  Rounding to bf16 is a chaotic map: a value that lands on the other side of a rounding boundary on the device
    (fp32 accumulation) than in the oracle (float64) is off by 2^-8, which moves everything in its receptive field by ~1e-3
    relative and flips a quarter of THOSE roundings.  So the agreement is only far below the bf16 noise itself (per tensor
    4e-2 .. 1e-1 between the emulating and the exact oracle on these networks) while few such cones exist -- at small
    images: measured per-tensor 1.4e-3 .. 5.2e-3 here, identical for NW = 4 and 8 (profiles/r02_bf16_emulation_evidence.txt;
    at 40 x 40 it is 5e-2 already, with BatchNorm over two levels 3e-1).  Bound: 2e-2 per tensor, every tensor.

    Round 4: a second size for the bf16-STORED operand kernels (BatchNorm, one level, S = 16: measured 1.4e-2 per tensor, same bound), and
    BatchNorm over TWO levels at f0 = 64 / S = 16 and f0 = 32 / S = 24 (the launch mix of three resolutions with stored-bf16 tensors
    between them).  There the rounding cones cover the image: measured per-tensor error 8e-2 .. 1.4e-1 in the median, 0.22 .. 0.44 at
    worst, err_l2 0.08 .. 0.14 -- the level of bf16 noise itself (emulating against exact oracle: 4e-2 .. 1e-1 per tensor on the
    one-level networks), identical for NW = 4 and 8 to seven digits.  Those two cases are held to median <= 0.25, err_l2 <= 0.25,
    worst tensor <= 0.8: a mis-indexed or dropped tensor is off by 1.0 and takes err_l2 with it; what they pin beyond that is that
    both wave variants agree bit-for-bit in the loss and that every launch of the mix runs.  The tight bound stays with the cases above."""
    import json
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, 'bf16_emul_case.py'), str(f0), str(S), str(B), str(bn), '32', str(n_down)],
                       env=dict(os.environ, DNNCA_IGB_NW=str(nw)), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    o = json.loads(r.stdout.strip().splitlines()[-1])
    if not bn:      # (with BatchNorm the inference logits use the moving statistics: not the oracle's training logits)
        assert o['dl_max'] <= 2e-3 and o['dl_median'] <= 5e-4, (o['dl_max'], o['dl_median'])
    # (two BatchNorm levels: the LOSS moves with the summation order of the batch statistics too -- 4.3e-3 with the channel-major
    #  epilogue of k_igb_conv3, 1.6e-3 with the pixel-major one before it, the same value for NW = 4 and 8 either time: bound 1e-2)
    assert abs(o['loss'] - o['loss_ref']) <= (1e-2 if bn and n_down > 1 else 1e-3) * max(1.0, abs(o['loss_ref']))
    if bn and n_down > 1:      # two BatchNorm levels: bf16 rounding is chaotic there (docstring)
        pt = sorted(o['per_tensor'].values())
        assert pt[len(pt) // 2] <= 0.25 and pt[-1] <= 0.8 and o['err_l2'] <= 0.25, (pt[len(pt) // 2], pt[-1], o['err_l2'])
    else:
        bad = {n: e for n, e in o['per_tensor'].items() if not e <= 2e-2}
        assert not bad and o['err_l2'] <= 1e-2, (o['err_l2'], bad)
    plan = set(o['plan'])
    Hp.record_oracle_plan(plan, 'test_bf16_kernels_against_bf16_emulating_oracle')
    suffix = '_w%d' % nw
    if bn and f0 == 32:        # (mixed 32 / 64-channel network: only its 64-channel level stores bf16 tensors)
        assert any(n.startswith('igb_conv_fwd' + suffix) for n in plan) and any(n.startswith('igb_conv_dgrad' + suffix) for n in plan), plan
    else:
        assert {'igb_conv_fwd' + suffix, 'igb_conv_dgrad' + suffix + ('_a16' if bn else '')} <= plan, plan   # the variant under test ran
    assert not any(n.startswith('igb_conv') and n.endswith('_w%d' % (12 - nw)) for n in plan)
    if f0 == 64 and not bn:
        assert {'igb_wgrad64', 'igb_tconv_fwd', 'igb_tconv_wgrad', 'igb_tconv_dgrad', 'igb_wgrad'} <= plan
        assert 'ig_tconv_fwd' not in plan           # f0 = 64: every transposed conv contracts in bf16


def test_unet_big_bf16_full_size_against_fp32(gpu):
    """BASELINE configs[2] at its own shape -- configs/unet_big.yaml, batch 4, 512 x 512, dtype bf16 -- against the fp32 tuned
    path on the same weights and batch: logits, loss, BatchNorm moving statistics, and every gradient tensor.  This is the
    launch configuration the benchmark runs (eight-wave k_igb_conv3 for forward and data gradient, bf16-stored operands).
    Bounds = 3 x what was measured (profiles/r02_bf16_fullsize_evidence.txt): logits 6.6e-4 of |logit| <= 0.11, loss 5.1e-4
    relative, state 2.8e-5; per-tensor gradient cosine >= 0.86 -- a freshly initialised 23-layer BatchNorm network amplifies
    the 2^-8 operand rounding (tests above: the kernels themselves agree with a bf16-emulating oracle to 5e-3 per tensor)."""
    from dnncancerannotator_amd.synthetic import synthetic_batch
    full = dict(rate=2, kernel_size=3, conv_stride=1, padding='same', n_filters_first=64, n_downsample=4, bn=True)
    spec = O.ModelSpec('unet', 1, **full)
    B, H, W = 4, 512, 512
    x, y = synthetic_batch(B, H, W, 1)
    res = {}
    for dt in ('f32', 'bf16'):
        m = gpu.DeviceModel('unet', 1, H, W, B, dtype=dt, **full)
        m.init_glorot(seed=3)
        _, lg = m.forward(x, training=False, return_logits=True)
        out = m.train_step(x, y, 0.0, m.loss_cfg(weight_mul=3.0))
        res[dt] = (lg.copy(), out.loss, m.get_grads().astype(np.float64), m.get_state().copy(), set(r[0] for r in m.plan()))
        m.close()
    (l0, loss0, g0, s0, _), (l1, loss1, g1, s1, names) = res['f32'], res['bf16']
    assert np.abs(l1 - l0).max() <= 2e-3 and np.median(np.abs(l1 - l0)) <= 3e-4
    assert abs(loss1 - loss0) <= 2e-3 * abs(loss0)
    assert np.abs(s1 - s0).max() <= 1e-4 * max(1.0, float(np.abs(s0).max()))
    cos = _per_tensor_cosine(spec, g1, g0)
    deg = Hp.degenerate_tensors(spec)
    low = {n: c for n, c in cos.items() if n not in deg and not c >= 0.75}
    assert not low, low
    assert np.median([c for n, c in cos.items() if n not in deg]) >= 0.9
    assert {'igb_conv_fwd_w8_a16', 'igb_conv_dgrad_w8_a16', 'igb_wgrad64', 'igb_tconv_fwd'} <= names, names


def test_unet_big_bf16_contraction_against_oracle(gpu):
    """configs/unet_big.yaml with dtype bf16 end to end against the float64 oracle.  One 64x64 slice leaves the deepest
    BatchNorm 16 samples per channel and the random-init network is ill-conditioned (fp32 numpy itself shows ~1e-3 .. 2e-2
    per-tensor gradient error there, profiles/r02_mask_flip_evidence.txt), so the forward quantities are held to a tolerance
    (logits 6e-2 absolute, loss 2e-2 relative) and every gradient tensor must point the right way (per-tensor cosine; the
    arithmetic itself is pinned by test_bf16_kernels_against_bf16_emulating_oracle)."""
    full = dict(rate=2, kernel_size=3, conv_stride=1, padding='same', n_filters_first=64, n_downsample=4, bn=True)
    spec = O.ModelSpec('unet', 1, **full)
    params = O.init_params(spec, seed=2)
    x, y = O.synthetic_batch(1, 64, 64, 1)
    m = gpu.DeviceModel('unet', 1, 64, 64, 1, dtype='bf16', **full)
    m.set_params(O.flatten(spec, params))
    cfg = dict(weight_mul=3.0)
    p64 = {n: v.astype(np.float64) for n, v in params.items()}
    _, lref = O.predict(spec, p64, x.astype(np.float64))
    _, lg = m.forward(x, training=False, return_logits=True)
    assert np.abs(lg - lref).max() <= 6e-2
    out = m.train_step(x, y, 1e-3, m.loss_cfg(**cfg))
    loss, grads, _, _ = O.loss_and_grads(spec, p64, x.astype(np.float64), y, cfg, training=True)
    assert abs(out.loss - loss) <= 2e-2 * max(1.0, abs(loss))
    g, gref = m.get_grads().astype(np.float64), O.flatten(spec, grads)
    assert np.isfinite(g).all()
    cos = _per_tensor_cosine(spec, g, gref)
    deg = Hp.degenerate_tensors(spec)
    low = {n: c for n, c in cos.items() if n not in deg and not c >= BF16_SMALL_COS}
    assert not low, low
    assert np.median([c for n, c in cos.items() if n not in deg]) >= 0.85
    m.close()


def test_bf16_trains_like_fp32(gpu):
    """dtype bf16 (BASELINE configs[2]) as a TRAINING arithmetic, not only as a kernel: unet_big's widths (64 .. 256 channels, two
    levels, BatchNorm), 64 x 64, batch 4, 300 Adam steps (the last 100 at a tenth of the learning rate, so that the runs settle) on
    synthetic discs from the same initial weights, bf16 against fp32 (tests/bf16_training_case.py; curves in
    profiles/r03_bf16_training.txt).  The yardstick is fp32 against ITSELF: the float atomics of the weight gradients make two fp32
    runs drift apart too.  Measured over three repetitions: fp32 vs fp32 -- Dice of the held-out masks 0.989 .. 0.994, held-out loss
    within 0.4 %; bf16 vs fp32 -- Dice 0.988 .. 0.989, held-out loss -0.2 .. -1.9 %; bf16 with fp32 activations (DNNCA_NO_HALF_Z /
    DNNCA_NO_HALF_DY: only the MFMA operands are rounded) -- 0.987 .. 0.988, -0.4 .. -2.3 %: storing BatchNorm inputs and gradients as
    bf16 costs nothing measurable on top of the operand rounding.  Bounds: masks Dice >= 0.98 against the fp32 run's, held-out loss
    within 5 %, every run segments the held-out discs (Dice >= 0.95 against the truth; measured 0.963 .. 0.970 for all of them), and
    the first 100 steps -- before the drift has compounded -- within 2 % step by step."""
    import bf16_training_case as T
    res = T.run(gpu, steps=300)
    assert {'igb_conv_fwd_w4_a16', 'igb_wgrad64'} & set(res['bf16']['plan']) and not any(k.startswith('igb_') for k in res['f32']['plan'])
    for name in ('f32', 'f32_again', 'bf16', 'bf16_f32act'):
        assert res[name]['dice_truth'] >= 0.95, (name, res[name]['dice_truth'])
        assert res[name]['curve'][-1] <= 0.02 * res[name]['curve'][0]              # the loss fell by two orders of magnitude
    for name in ('bf16', 'bf16_f32act'):
        r = res[name]
        assert r['dice_vs_f32'] >= 0.98, (name, r['dice_vs_f32'], res['f32_again']['dice_vs_f32'])
        assert abs(r['eval_loss_rel']) <= 0.05, (name, r['eval_loss_rel'], res['f32_again']['eval_loss_rel'])
        early = np.abs(np.array(r['curve'][:10]) / np.array(res['f32']['curve'][:10]) - 1.0)
        assert early.max() <= 0.02, (name, early)


@pytest.mark.parametrize('arch,C,opts,size', [
    ('unet', 1, dict(n_filters_first=64, n_downsample=2), 64),
    ('mulmo', 3, dict(n_filters_first=16, n_downsample=4), 128),       # 16/32-channel levels stay f32, 64/128 go bf16: mixed plan
])
def test_bf16_storage_of_rounded_tensors_is_transparent(gpu, monkeypatch, arch, C, opts, size):
    """dtype bf16 keeps the tensors whose every reader rounds to bf16 anyway (BatchNorm outputs feeding the 64-channel conv
    kernels, the conv-output gradients from the BatchNorm backward) as bf16 in HBM (ig_plan_half).  That must not change a
    single bit of the forward pass; gradients may differ by the summation order of the float atomics only.
    (The BatchNorm inputs and the gradients arriving at a BatchNorm are stored as bf16 too -- "bf16 activations", which rounds
    them and is not transparent: switched off here, covered by test_bf16_kernels_against_bf16_emulating_oracle.)"""
    monkeypatch.setenv('DNNCA_NO_HALF_Z', '1')
    monkeypatch.setenv('DNNCA_NO_HALF_DY', '1')
    opts = dict(rate=2, kernel_size=3, conv_stride=1, padding='same', bn=True, **opts)
    x, y = O.synthetic_batch(2, size, size, C)
    res = []
    for no_half in (False, True):
        if no_half:
            monkeypatch.setenv('DNNCA_NO_HALF', '1')
        m = gpu.DeviceModel(arch, C, size, size, 2, dtype='bf16', **opts)
        m.init_glorot(seed=2)
        plan_bytes = sum(b for k, b, f in m.plan() if k.startswith('bn_'))
        _, lg = m.forward(x, training=True, return_logits=True)
        m.train_step(x, y, 1e-3, m.loss_cfg(weight_mul=3.0))
        res.append((plan_bytes, lg.copy(), m.get_grads().copy()))
        m.close()
    (bh, lh, gh), (bf, lf, gf) = res
    assert bh < bf                       # the BatchNorm passes really write 2-byte elements
    assert np.array_equal(lh, lf)
    assert np.abs(gh - gf).max() <= 1e-5 * np.abs(gf).max()


@pytest.mark.parametrize('arch, C, dtype, opts, size', [('unet', 1, 'bf16', dict(n_filters_first=64, n_downsample=2), 64),
                                                       ('mulmo', 3, 'f32', dict(n_filters_first=16, n_downsample=2), 64)])
def test_weight_gradients_on_the_side_stream_change_nothing(gpu, monkeypatch, arch, C, dtype, opts, size):
    """The dense convs' weight gradients run on a second stream beside the main chain (Model::wg_side_begin): same kernels, same
    operands, so loss, BatchNorm state and every gradient tensor equal the single-stream run up to the summation order of the
    float atomics.  Three steps on three different batches at learning rate 0 (the weights stay put, so the steps do not amplify
    that noise): a missing fork / join dependency -- a weight gradient still running when the next step zeroes the gradient
    vector or rewrites the activations -- would show in the third step's gradient."""
    opts = dict(rate=2, kernel_size=3, conv_stride=1, padding='same', bn=True, **opts)
    spec = O.ModelSpec(arch, C, **opts)
    batches = [O.synthetic_batch(2, size, size, C, seed_x=10 + k, seed_y=20 + k) for k in range(3)]
    res = []
    for single in (False, True):
        if single:
            monkeypatch.setenv('DNNCA_NO_WG_STREAM', '1')
        m = gpu.DeviceModel(arch, C, size, size, 2, dtype=dtype, **opts)
        m.init_glorot(seed=4)
        losses = [m.train_step(x, y, 0.0, m.loss_cfg(weight_mul=3.0)).loss for x, y in batches]
        res.append((losses, m.get_grads().astype(np.float64), m.get_state().copy()))
        m.close()
    (l1, g1, s1), (l0, g0, s0) = res
    np.testing.assert_allclose(l1, l0, rtol=1e-6)
    Hp.assert_grads_per_tensor_nofixture(spec, g1, g0, 2e-5, what='side stream vs one stream (third step)')
    assert np.abs(s1 - s0).max() <= 1e-6 * max(1.0, np.abs(s0).max())


def test_cli_train_then_evaluate_on_tfrecords(gpu, tmp_path):
    """`python3 -m annotator train|evaluate` with the reference's YAML surface and .tfrecords exam files, in a child
    process (the documented drop-in invocation)."""
    import subprocess
    import sys
    import yaml
    from dnncancerannotator_amd import tfrecord as T
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rng = np.random.default_rng(5)
    types = ['TRA', 'ADC', 'DWI', 'label']
    exams = []
    for i in range(2):
        s = rng.integers(0, 256, (4, 48, 48, 4), dtype=np.uint8)     # 8 px of margin: the training crop jitters by up to 6
        s[..., 3] = 0
        s[:, 14:24, 16:26, 3] = 255
        exams.append(T.make_example(s, i, i, '/e/%d' % i, 'cancer', types))
    rec = str(tmp_path / 'exams.tfrecords')
    T.write_records(rec, exams)
    cfgs = {
        'unet.yaml': dict(model='UNetAnnotator', model_options=dict(n_filters_first=3, n_downsample=3, rate=2, kernel_size=3,
                                                                     conv_stride=1, bn=False, padding='same')),
        'deploy.yaml': {'deploy_options': {'optimizer': 'adam',
                                           'LearningRateScheduler': 'lambda epoch, current_lr: 0.001 * 0.96 ** (epoch // 1000)',
                                           'loss': {'class_name': 'WeightedCrossentropy', 'config': {'weight_mul': 3.0}},
                                           'enable_multigpu': False}},
        'data.yaml': {'data_options': {'train': {'batch_size': 4, 'output_size': [32, 32], 'slice_types': types},
                                       'eval': {'batch_size': 8, 'output_size': [32, 32], 'slice_types': types}}},
    }
    paths = []
    for name, obj in cfgs.items():
        p = tmp_path / name
        p.write_text(yaml.safe_dump(obj))
        paths.append(str(p))
    save = str(tmp_path / 'run')
    env = dict(os.environ, PYTHONPATH=root)
    r = subprocess.run([sys.executable, '-m', 'annotator', 'train', '--config'] + paths +
                       ['--save_path', save, '--data_path', rec, '--max_steps', '12', '--save_freq', '6'],
                       cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert sorted(f for f in os.listdir(os.path.join(save, 'checkpoints')) if f.endswith('.index')) == ['ckpt-12.index', 'ckpt-6.index']
    assert os.path.exists(os.path.join(save, 'options.yaml')) and os.path.exists(os.path.join(save, 'results.pkl'))
    r = subprocess.run([sys.executable, '-m', 'annotator', 'evaluate', '--save_path', save, '--data_path', rec, '--tag', 'val',
                        '--export_csv', '--skip_visualization'], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    rows = open(os.path.join(save, 'tfevents', 'val', 'results.csv')).read().strip().splitlines()
    assert rows[0].startswith('step,loss') and [l.split(',')[0] for l in rows[1:]] == ['6', '12']


@pytest.mark.parametrize('arch, C, opts', [
    ('unet', 1, dict(n_filters_first=3, n_downsample=3, bn=False)),                      # configs/unet.yaml + overlays
    ('mulmo', 2, dict(n_filters_first=16, n_downsample=2, bn=True)),                      # dense-channel kernels
    ('unet', 1, dict(n_filters_first=16, n_downsample=1, bn=True, rate=4)),               # rate 4: 16 -> 64 channel convs, 4x4 pool / transposed conv
])
def test_non_square_leaky_l2_tuned_kernels(gpu, arch, C, opts):
    """Tuned kernels on a non-square image (48 x 80: partial tiles in both directions) with the LeakyReLU and L2-regulariser
    overlays of the reference (configs/additionals/leakyReLU.yaml, kernel_regularizer.yaml) against the float64 oracle."""
    full = dict(rate=2, kernel_size=3, conv_stride=1, padding='same',
                activation={'class_name': 'LeakyReLU', 'config': {'alpha': 0.3}},
                kernel_regularizer={'class_name': 'L2', 'config': {'l2': 0.01}})
    full.update(opts)
    spec = O.ModelSpec(arch, C, **full)
    params = Hp.perturbed_params(spec, np.float64)
    B, H, W = 3, 48, 80
    rng = np.random.default_rng(11)
    x = rng.random((B, H, W, C)).astype(np.float32)
    y = (rng.random((B, H, W)) < 0.05).astype(np.float32)
    cfg = dict(weight_mul=3.0, weight_add=0.25)
    loss, grads, logits, state = O.loss_and_grads(spec, params, x.astype(np.float64), y, cfg, training=True)
    dev_opts = {k: v for k, v in full.items() if k not in ('activation', 'kernel_regularizer')}
    m = gpu.DeviceModel(arch, C, H, W, B, leaky_alpha=0.3, l2=0.01, **dev_opts)
    m.set_params(O.flatten(spec, params))
    if m.n_state:
        m.set_state(O.flatten(spec, params, trainable=False))
    out = m.train_step(x, y, 0.0, m.loss_cfg(**cfg))
    assert abs(out.loss - loss) <= 1e-4 * max(1.0, abs(loss))
    p32 = {n: v.astype(np.float32) for n, v in params.items()}
    _, g32, _, _ = O.loss_and_grads(spec, p32, x, y, cfg, training=True)
    gref, g32 = O.flatten(spec, grads), O.flatten(spec, g32).astype(np.float64)
    floor = [10 * np.abs(g32[sl] - gref[sl]).max() for _, sl in Hp.tensor_slices(spec)]
    Hp.assert_grads_per_tensor(spec, m.get_grads(), gref, 2e-5, floor=floor)
    plan = set(r[0] for r in m.plan())
    assert any(k.startswith('pgbwd_') or k.startswith('ig_') or k.startswith('ig3x_') for k in plan)              # the tuned kernels are the ones planned
    Hp.record_oracle_plan(m, 'test_non_square_leaky_l2_tuned_kernels')
    m.close()


@pytest.mark.parametrize('B, H, W, strip', [(2, 40, 128, True), (8, 16, 256, True), (2, 40, 128, False), (8, 16, 256, False),
                                            (3, 24, 200, True), (1, 72, 64, True), (2, 64, 256, True), (2, 64, 256, 'per-layer backward'),
                                            (2, 64, 256, 'fz_up2')])
def test_vector_alu_kernels_of_the_3_channel_level_against_oracle(gpu, monkeypatch, B, H, W, strip):
    """configs/unet.yaml, the vector-ALU kernels of the full-resolution 3-channel level against the float64 oracle, every variable on
    its own scale.  strip: the column-strip kernels (strip_dev.h; strips of 60 columns, row chunks; shapes with partial strips, one
    strip, chunk counts not divisible by 8) -- k_tail3: the conv that feeds the head runs forward + head + loss + its whole backward
    in one launch; k_first3: the backward of the first encoder block (second conv with the folded max-pool backward + the first
    conv's weight gradient) in one launch.  Without them and on images made of whole 128 x 8 tiles those convs' backward is the
    tile kernel k_bwd3v (plain / with the pool fold); the (8, 16, 256) shape has a tile count divisible by 8 (XCD-aware order).
    (2, 64, 256) is made of whole tiles at every level: there the transposed convs' backward rides in the launch of the two-source
    conv behind them on all three levels (k_pgbwd TCF / TCM) and the 12 -> 6 transposed conv's forward in the fused 12-channel
    decoder block (k_fz_up); by default the 6- and 12-channel blocks' backward passes are the block-fused launches of
    kernels_fused_bwd.hip (k_fzb), so the shape runs a second time with DNNCA_NO_FUSED_BWD for the per-layer TCM kernels."""
    fz_up2 = strip == 'fz_up2'          # the 256^2-level decoder block's two convs as ONE forward launch (k_fz_up<..., NOTC>; opt-in, DNNCA_FZ_UP2)
    if fz_up2:
        monkeypatch.setenv('DNNCA_FZ_UP2', '1')
        strip = True
    per_layer_bwd = strip == 'per-layer backward'
    if per_layer_bwd:
        monkeypatch.setenv('DNNCA_NO_FUSED_BWD', '1')
        strip = True
    if not strip:
        monkeypatch.setenv('DNNCA_NO_TAIL3', '1')
        monkeypatch.setenv('DNNCA_NO_FIRST3', '1')
    spec = O.ModelSpec('unet', 1, **UNET)
    params = Hp.perturbed_params(spec, np.float64)
    rng = np.random.default_rng(13)
    x = rng.random((B, H, W, 1)).astype(np.float32)
    y = (rng.random((B, H, W)) < 0.05).astype(np.float32)
    cfg = dict(weight_mul=3.0)
    loss, grads, logits, _ = O.loss_and_grads(spec, params, x.astype(np.float64), y, cfg, training=True)
    m = gpu.DeviceModel('unet', 1, H, W, B, **UNET)
    m.set_params(O.flatten(spec, params))
    out = m.train_step(x, y, 0.0, m.loss_cfg(**cfg))
    assert abs(out.loss - loss) <= 1e-4 * max(1.0, abs(loss))
    p32 = {n: v.astype(np.float32) for n, v in params.items()}
    _, g32, _, _ = O.loss_and_grads(spec, p32, x, y, cfg, training=True)
    gref, g32 = O.flatten(spec, grads), O.flatten(spec, g32).astype(np.float64)
    floor = [10 * np.abs(g32[sl] - gref[sl]).max() for _, sl in Hp.tensor_slices(spec)]
    Hp.assert_grads_per_tensor(spec, m.get_grads(), gref, 2e-5, floor=floor)
    plan = set(r[0] for r in m.plan())
    if (B, H, W) == (2, 64, 256):
        bwd = {'pgbwd_tc_6x2_6', 'pgbwd_tc_12x2_12', 'pgbwd_pool_12x1_12', 'pgbwd_pool_6x1_6'} if per_layer_bwd else \
              {'fzb_up_6', 'fzb_up_12', 'fzb_down_6_12', 'fzb_down_3_6'}
        assert ({'pgbwd_tc_3x2_3', 'fz_up_tc_12_12', 'fz_down_3_6', 'fz_down_6_12', 'pg_fold'} | bwd) <= plan and not any(k.startswith('tconv') for k in plan), plan
        assert per_layer_bwd or not any(k.startswith('pgbwd_') and k != 'pgbwd_tc_3x2_3' for k in plan), plan
        assert ('fz_up2_6' in plan and 'pgfwd_6x2_6' not in plan) if fz_up2 else ('pgfwd_6x2_6' in plan and 'fz_up2_6' not in plan), plan
    assert ('tail3_3x1_3' if strip else 'bwd3v_3x1_3') in plan, plan
    if 'fz_down_1_3' in plan or 'first3_fwd' in plan:          # the fused first block records the pool's window positions
        assert ('first3_bwd' if strip else 'bwd3v_pool_3x1_3') in plan and 'pgbwd_w_1x1_3' not in plan or not strip, plan
    Hp.record_oracle_plan(m, 'test_vector_alu_kernels_of_the_3_channel_level_against_oracle')
    m.close()


@pytest.mark.parametrize('leaky', [0.0, 0.3])
@pytest.mark.parametrize('B, H, W', [(2, 64, 256), (3, 32, 128), (1, 96, 128), (9, 32, 256)])
def test_block_fused_backward_against_oracle(gpu, leaky, B, H, W):
    """kernels_fused_bwd.hip: the whole backward of a Downsample / Upsample block (components.py:77-81, 158-166) of the 6- and
    12-channel levels in one launch (k_fzb: second conv, first conv, transposed conv / max-pool; data-gradient and weight-gradient
    waves) against the float64 oracle, every variable on its own scale.  Shapes: several tiles per block and image; one tile row per
    image at the 12-channel level (every tile touches the top and the bottom edge); one tile column at the 6-channel level; more
    images than XCDs with a tile count that is not a multiple of 8 (plain tile order).  LeakyReLU: the gradient of the first conv's
    output is masked with a non-zero slope, so the ring of the LDS tile outside the image is zeroed explicitly; the pool fold then
    keeps the per-layer kernels (its ReLU-only tie rule), so only the decoder blocks fuse."""
    full = dict(UNET)
    if leaky:
        full['activation'] = {'class_name': 'LeakyReLU', 'config': {'alpha': leaky}}
    spec = O.ModelSpec('unet', 1, **full)
    params = Hp.perturbed_params(spec, np.float64)
    # (seed: on some random inputs two values of a pooling window agree to float32 precision and the float64 oracle routes that
    #  window's gradient to the other one -- with seed 17 the (9, 32, 256) LeakyReLU case is off by 2.9e-3 on the down2 tensors for
    #  the generic kernels and for every tuned variant alike; this seed has no such window in any of the eight cases)
    rng = np.random.default_rng(23)
    x = rng.random((B, H, W, 1)).astype(np.float32)
    y = (rng.random((B, H, W)) < 0.05).astype(np.float32)
    cfg = dict(weight_mul=3.0)
    loss, grads, logits, _ = O.loss_and_grads(spec, params, x.astype(np.float64), y, cfg, training=True)
    m = gpu.DeviceModel('unet', 1, H, W, B, **UNET, **(dict(leaky_alpha=leaky) if leaky else {}))
    m.set_params(O.flatten(spec, params))
    out = m.train_step(x, y, 0.0, m.loss_cfg(**cfg))
    assert abs(out.loss - loss) <= 1e-4 * max(1.0, abs(loss))
    p32 = {n: v.astype(np.float32) for n, v in params.items()}
    _, g32, _, _ = O.loss_and_grads(spec, p32, x, y, cfg, training=True)
    gref, g32 = O.flatten(spec, grads), O.flatten(spec, g32).astype(np.float64)
    floor = [10 * np.abs(g32[sl] - gref[sl]).max() for _, sl in Hp.tensor_slices(spec)]
    Hp.assert_grads_per_tensor(spec, m.get_grads(), gref, 2e-5, floor=floor)
    plan = set(r[0] for r in m.plan())
    want = {'fzb_up_6', 'fzb_up_12'} | (set() if leaky else {'fzb_down_6_12', 'fzb_down_3_6'})
    assert want <= plan, plan
    Hp.record_oracle_plan(m, 'test_block_fused_backward_against_oracle')
    m.close()


@pytest.mark.parametrize('B, H, W', [(1, 8, 64), (2, 24, 120), (5, 32, 248), (3, 88, 504), (1, 512, 512), (11, 64, 64), (3, 96, 384)])
def test_strip_kernels_match_the_per_layer_kernels(gpu, monkeypatch, B, H, W):
    """The column-strip kernels (k_tail3, k_first3, k_first3_fwd, k_up3_fwd) and the ride-along launches (transposed-conv backward in
    the two-source conv's launch, Adam in the slab fold, operand preparation in the first strip launch) against the per-layer / tile
    kernels and separate launches they replace, on
    shapes that stress their bookkeeping: one chunk, partial last strips, odd chunk heights, one image, more images than XCDs.
    Same device, same weights: loss, probabilities and every gradient tensor (float32 sums in two different orders)."""
    from dnncancerannotator_amd.synthetic import synthetic_batch
    x, y = synthetic_batch(B, H, W, 1)
    spec = O.ModelSpec('unet', 1, **UNET)
    rng = np.random.default_rng(5)

    def run():
        m = gpu.DeviceModel('unet', 1, H, W, B, **UNET)
        m.init_glorot(seed=2)
        p0 = m.get_params()
        m.set_params((p0 + np.random.default_rng(7).uniform(-0.05, 0.05, p0.shape)).astype(np.float32))
        out = m.train_step(x, y, 0.0, m.loss_cfg(weight_mul=3.0))
        g = m.get_grads()
        prob = m.forward(x, training=False)
        plan = set(r[0] for r in m.plan())
        m.close()
        return out.loss, g, prob, plan

    l1, g1, p1, plan1 = run()
    for k in ('DNNCA_NO_TAIL3', 'DNNCA_NO_FIRST3', 'DNNCA_NO_FIRST3F', 'DNNCA_NO_UP3F', 'DNNCA_NO_TCF', 'DNNCA_NO_FOLD_ADAM', 'DNNCA_NO_PREP_RIDE',
              'DNNCA_NO_TCONV_RIDE', 'DNNCA_NO_TCM', 'DNNCA_NO_FUSED_BWD'):
        monkeypatch.setenv(k, '1')
    l0, g0, p0, plan0 = run()
    fused = {'tail3_3x1_3', 'first3_fwd', 'up3_fwd'} | ({'fz_up_tc_12_12', 'fzb_up_6', 'fzb_up_12', 'fzb_down_6_12', 'fzb_down_3_6'} if W % 128 == 0 and H % 32 == 0 else set())
    assert fused <= plan1 and not ((fused | {'first3_bwd'}) & plan0), (plan1, plan0)
    assert abs(l1 - l0) <= 1e-5 * max(1.0, abs(l0))
    assert np.abs(np.asarray(p1) - np.asarray(p0)).max() <= 2e-5
    # two float32 paths sum the convolutions in different orders: a pooling window whose two largest values agree to the last bit or
    # two routes its gradient differently (one pixel of 256 K at the first level: measured 1.6e-4 of encoder.down1.conv0.kernel's
    # scale on the one-image shape, <= 1e-4 everywhere else) -- 3 x FULL_TOL; a mis-indexed strip, chunk or halo is off by >= 1e-2
    Hp.assert_grads_per_tensor(spec, g1, g0, 3 * FULL_TOL, what='strip vs per-layer kernels')
    del rng


def test_rccl_one_rank_rehearsal(gpu):
    """The RCCL calls of the DP path (unique id, communicator, gradient all-reduce on the step's stream, broadcast,
    state average, host all-reduce) on a one-rank communicator: a sum over one rank is the identity, so weights, BN
    state, gradients and losses equal the run without a communicator (up to the float-atomic run-to-run noise)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'rccl_one_rank.py')], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    # float atomics make the weight gradients differ run to run in the last bits; three Adam steps (update ~ g / |g|)
    # amplify that to ~1e-4 of the weights, so the bound on the weights is the larger of a fixed 1e-3 and the noise sample
    tol = dict(params=1e-3, state=1e-5, grads=1e-4)
    for key, d in out['diff'].items():
        assert d <= max(tol[key], 4 * out['noise'][key]), out
    # three steps from host buffers, then three through the staging ring (copy stream, scalars one step late) with the communicator
    assert len(out['losses_rccl']) == 6
    np.testing.assert_allclose(out['losses_plain'][:3], out['losses_rccl'][:3], rtol=1e-6)
    np.testing.assert_allclose(out['losses_plain'][3:], out['losses_rccl'][3:], rtol=1e-4)
    assert out['red'] == [1.5, -2.0]
    assert out['red_big'] is True          # 5000 doubles > 2^40 through the chunked host all-reduce, exact


def test_bench_n_gt_1_sequence_on_one_rank(gpu):
    """bench.py's N > 1 branch on one GPU (DNNCA_FORCE_COMM=1): the RCCL unique id through the rendezvous file, a one-rank
    communicator, barriers and the max-reduction of the region's wall time through it, the data-parallel launch sequence (slab
    fold, ncclAllReduce of [gradients, loss], Adam with 1/world) in every timed step -- and the line still parses."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DNNCA_FORCE_COMM='1')
    env.pop('HSA_ENABLE_IPC_MODE_LEGACY', None)          # _lib.load() must provide it
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--steps', '5', '--warmup', '2', '--no-cpu-baseline'],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line['n_gpus'] == 1 and line['steps'] == 5 and line['config']['communicator'] is True
    assert line['repeats'] == 3 and len(line['region_ms']) == 3
    assert line['value'] > 1000 and abs(line['value'] - 8 * 5 / (line['ms_per_step'] * 5e-3)) <= 0.01 * line['value']
    names = [k['kernel'] for k in line['roofline_all']]
    assert 'pg_fold' in names and 'g_adam' in names and 'pg_fold_adam' not in names, names      # the data-parallel sequence ran
    assert 'other_workloads' not in line and 'from_host_memory' not in line


def test_bucketed_gradient_allreduce_rehearsal(gpu):
    """Gradient vectors above 1 MB (unet_big 63 MB, mulmo_unet 6.9 MB) are all-reduced in buckets on a second stream while
    the backward pass runs (reverse layer order: the backward pass finalises the flat gradient vector from its end).  On a
    one-rank communicator every sum is the identity, so training with 256 KB buckets must equal training without a
    communicator (up to the run-to-run noise of the float atomics), and the step must really have issued several collectives."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'rccl_one_rank.py'), 'big'], capture_output=True, text=True,
                       timeout=300, env=dict(os.environ, DNNCA_BUCKET_BYTES='262144'))
    assert r.returncode == 0, r.stdout + r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out['n'] * 4 > (1 << 20)
    assert min(out['calls']) >= 5 and out['calls_plain'] == [0, 0, 0], out       # ~4 MB in 256 KB buckets + the remainder + the loss
    assert out['diff_grads'] <= max(1e-4, 4 * out['noise_grads']), out
    assert out['diff_params'] <= max(1e-3, 4 * out['noise_params']), out
    np.testing.assert_allclose(out['losses_plain'], out['losses_rccl'], rtol=1e-5)


@pytest.mark.parametrize('force_generic', [False, True])
def test_label_smoothing_loss_matches_oracle(gpu, force_generic):
    """deploy_options.loss.config.label_smoothing (utils/losses.py:62-67): the labels are blurred on the device (filter 6, sigma 3,
    REFLECT padding) before the positive rate and the loss; loss, weight and gradients against the oracle, 40 x 24 image."""
    spec = O.ModelSpec('unet', 1, **UNET)
    params = Hp.perturbed_params(spec, np.float64)
    B, H, W = 2, 40, 24
    rng = np.random.default_rng(8)
    x = rng.random((B, H, W, 1)).astype(np.float32)
    y = np.zeros((B, H, W), np.float32)
    y[0, 5:15, 4:12] = 1.0
    y[1, 30:40, 0:6] = 1.0                                    # touches two borders: the reflect padding matters
    cfg = dict(weight_mul=3.0, label_smoothing=True, label_smoothing_filter_size=6, label_smoothing_sigma=3)
    loss, grads, _, _ = O.loss_and_grads(spec, params, x.astype(np.float64), y, cfg, training=True)
    m = gpu.DeviceModel('unet', 1, H, W, B, force_generic=force_generic, **UNET)
    m.set_params(O.flatten(spec, params))
    out = m.train_step(x, y, 0.0, m.loss_cfg(**cfg))
    ys = O.gaussian_filter2d(y)
    assert abs(out.positive_rate - ys.mean()) <= 1e-6 and abs(out.label_max - ys.max()) <= 1e-6
    assert abs(out.loss - loss) <= 1e-4 * max(1.0, abs(loss))
    Hp.assert_grads_per_tensor(spec, m.get_grads(), O.flatten(spec, grads), 2e-5)
    plain = m.train_step(x, y, 0.0, m.loss_cfg(weight_mul=3.0))
    assert abs(plain.loss - out.loss) > 1e-3                  # and it is not a no-op
    if not force_generic:
        Hp.record_oracle_plan(m, 'test_label_smoothing_loss_matches_oracle')
    m.close()


@pytest.mark.parametrize('arch,C,B,opts', [
    ('mulmo', 3, 8, dict(n_filters_first=16, n_downsample=4, bn=True)),       # configs/mulmo_unet.yaml at the BASELINE batch (configs[3])
    ('unet', 1, 1, dict(n_filters_first=64, n_downsample=4, bn=True)),        # configs/unet_big.yaml (fp32 contraction)
])
def test_dense_configs_full_resolution_tuned_vs_generic(gpu, arch, C, B, opts):
    """The dense-channel configurations at the BASELINE resolution (512 x 512): the second-generation kernels (persistent conv,
    tiled weight gradients, fused BatchNorm statistics / pool, first-layer kernels, transposed-conv kernels) against the generic
    kernels on the same weights -- logits of an inference pass, and loss / gradients / BatchNorm moving statistics of a training
    step.  Tolerances: the networks are deep and freshly initialised (fp32 accumulation order differs between the paths)."""
    from dnncancerannotator_amd.synthetic import synthetic_batch
    H = W = 512
    full = dict(rate=2, kernel_size=3, conv_stride=1, padding='same', **opts)
    x, y = synthetic_batch(B, H, W, C)
    tuned = gpu.DeviceModel(arch, C, H, W, B, **full)
    generic = gpu.DeviceModel(arch, C, H, W, B, force_generic=True, **full)
    tuned.init_glorot(seed=3)
    p0 = tuned.get_params()
    generic.set_params(p0)
    _, lt = tuned.forward(x, training=False, return_logits=True)
    _, lg = generic.forward(x, training=False, return_logits=True)
    assert np.abs(lt - lg).max() <= 1e-3 * max(1.0, float(np.abs(lg).max()))
    cfg = tuned.loss_cfg(weight_mul=3.0)
    ot = tuned.train_step(x, y, 0.0, cfg)
    og = generic.train_step(x, y, 0.0, cfg)
    assert abs(ot.loss - og.loss) <= 1e-4 * max(1.0, abs(og.loss))
    gt, gg = tuned.get_grads().astype(np.float64), generic.get_grads().astype(np.float64)
    spec = O.ModelSpec(arch, C, **full)
    errs = Hp.assert_grads_per_tensor_nofixture(spec, gt, gg, DENSE_FULL_TOL, what='tuned vs generic')
    assert np.median(list(errs.values())) <= 1e-2
    assert Hp.rel_err(tuned.get_state(), generic.get_state()) <= 1e-4          # BatchNorm moving statistics
    names = set(r[0] for r in tuned.plan())
    assert {'ig3x_conv_fwd', 'ig3x_wgrad', 'first_fwd', 'bn_apply_pool'} <= names
    tuned.close()
    generic.close()
