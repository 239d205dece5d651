"""-m gpu: pixel metrics.  Device-side TP/FP/FN/TN (dnnca_pixel_confusion / dnnca_pixel_confusion_of: integer
work, BIT-EXACT against numpy `prob > t`, the Keras Precision/Recall/AUC comparison) and the metric formulas built on
them (utils/metrics.py:37-61 FBetaScore; configs/additionals/metrics.yaml:2-23), plus the engine's validation pass with a
validation batch larger than the training batch (data_options.yaml: train 8, eval 64)."""

import numpy as np
import pytest

from oracle import unet_oracle as O

pytestmark = pytest.mark.gpu

UNET = dict(n_filters_first=3, n_downsample=3, rate=2, kernel_size=3, conv_stride=1, bn=False, padding='same')


def auc_thresholds(n):
    """tf.keras.metrics.AUC(num_thresholds=n) [TF-2.6]: (i + 1) / (n - 1) for i < n - 2, bracketed by -eps and 1 + eps."""
    eps = 1e-7
    return np.array([0.0 - eps] + [(i + 1) / (n - 1) for i in range(n - 2)] + [1.0 + eps], np.float32)


def numpy_counts(prob, y, thr):
    p, yy = np.asarray(prob, np.float32).ravel(), np.asarray(y, np.float32).ravel() > 0.5
    out = []
    for t in np.asarray(thr, np.float32).ravel():
        pp = p > t
        out.append((int((pp & yy).sum()), int((pp & ~yy).sum()), int((~pp & yy).sum()), int((~pp & ~yy).sum())))
    return out


def adversarial_probs(rng, n, thr):
    """uniform values, a heavy cluster near 0 and 1 (what a trained segmentation net emits), and MANY values exactly on a
    threshold and one ulp either side of it"""
    p = rng.random(n).astype(np.float32)
    p[rng.random(n) < 0.5] *= np.float32(1e-3)
    p[rng.random(n) < 0.1] = 1.0
    p[rng.random(n) < 0.05] = 0.0
    t = np.asarray(thr, np.float32)
    k = n // 4
    idx = rng.choice(n, k, replace=False)
    pick = t[rng.integers(0, len(t), k)]
    side = rng.integers(0, 3, k)
    pick = np.where(side == 0, pick, np.where(side == 1, np.nextafter(pick, np.float32(2)), np.nextafter(pick, np.float32(-2))))
    p[idx] = pick.astype(np.float32)
    return p


@pytest.mark.parametrize('B,H,W', [(1, 8, 8), (3, 40, 24), (8, 512, 512)])
def test_pixel_confusion_bit_exact(gpu, B, H, W):
    rng = np.random.default_rng(B * 1000 + H)
    m = gpu.DeviceModel('unet', 1, H, W, B, **UNET)
    try:
        n = B * H * W
        y = (rng.random(n) < 0.03).astype(np.float32)
        grids = [np.array([0.5], np.float32), np.array([0.8], np.float32), np.array([0.5, 0.8], np.float32), auc_thresholds(150),
                 auc_thresholds(1024), np.array([0.8, 0.2, 0.8, 0.5, 1.5, -0.5, 0.2], np.float32)]      # unsorted, duplicates, outside [0, 1]
        for thr in grids:
            p = adversarial_probs(rng, n, thr)
            got = m.pixel_confusion_of(p, y, thr)
            want = numpy_counts(p, y, thr)
            assert [tuple(int(v) for v in row) for row in got] == want
            assert all(sum(row) == n for row in got)
        # fewer pixels than the model holds, NaN probabilities (nan > t is False), soft labels (cast to bool at 0.5)
        k = max(1, n // 3)
        p = adversarial_probs(rng, k, [0.5])
        p[::7] = np.nan
        ys = rng.random(k).astype(np.float32)
        assert [tuple(int(v) for v in r) for r in m.pixel_confusion_of(p, ys, [0.5, 0.8])] == numpy_counts(p, ys, [0.5, 0.8])
        with pytest.raises(Exception):
            m.pixel_confusion_of(p, ys, auc_thresholds(1025))
        with pytest.raises(Exception):
            m.pixel_confusion_of(p, ys, [np.nan])
    finally:
        m.close()


def test_confusion_of_the_last_eval_step_and_metric_formulas(gpu):
    """dnnca_pixel_confusion counts the probabilities of the last forward/eval step; metrics.py builds Precision / Recall /
    FBetaScore (utils/metrics.py:37-61) / AUC on the counts.  Checked against the formulas written out on numpy counts."""
    from dnncancerannotator_amd import metrics as M
    B, H, W = 4, 64, 64
    x, y = O.synthetic_batch(B, H, W, 1)
    m = gpu.DeviceModel('unet', 1, H, W, B, **UNET)
    try:
        m.init_glorot(seed=5)
        out, prob = m.eval_step(x, y, m.loss_cfg(weight_mul=3.0), return_prob=True)
        thr = np.quantile(prob, [0.3, 0.5, 0.9]).astype(np.float32)          # thresholds that cut through the actual outputs
        assert [tuple(int(v) for v in r) for r in m.pixel_confusion(y, thr)] == numpy_counts(prob, y, thr)
        t = float(thr[1])
        tp, fp, fn, tn = numpy_counts(prob, y, [t])[0]
        mets = dict(p=M.Precision(thresholds=t), r=M.Recall(thresholds=t), f1=M.FBetaScore(beta=1.0, thresholds=t),
                    f2=M.FBetaScore(beta=2.0, thresholds=t), roc=M.AUC(curve='ROC', num_thresholds=150),
                    pr=M.AUC(curve='PR', num_thresholds=150))
        for k in mets.values():
            k.update_state(m, y)
            k.update_state(m, y)            # accumulation over two batches doubles every count
        prec, rec = tp / max(tp + fp, 1), tp / max(tp + fn, 1)
        assert mets['p'].result() == pytest.approx(prec, abs=1e-12) and mets['r'].result() == pytest.approx(rec, abs=1e-12)
        for beta, key in ((1.0, 'f1'), (2.0, 'f2')):       # utils/metrics.py:57-60
            assert mets[key].result() == pytest.approx((1 + beta ** 2) * prec * rec / (beta ** 2 * prec + rec + 1e-7), abs=1e-12)
        # ROC-AUC by trapezoids over the 150-threshold grid, written out independently
        c = np.array(numpy_counts(prob, y, auc_thresholds(150)), np.float64)
        tpr, fpr = c[:, 0] / (c[:, 0] + c[:, 2]), c[:, 1] / (c[:, 1] + c[:, 3])
        assert mets['roc'].result() == pytest.approx(float(np.sum((fpr[:-1] - fpr[1:]) * (tpr[:-1] + tpr[1:]) / 2)), abs=1e-12)
        assert 0.0 <= mets['pr'].result() <= 1.0
        assert np.array_equal(mets['roc'].counts, 2 * c)
    finally:
        m.close()


def test_validation_batch_larger_than_training_batch(gpu, tmp_path):
    """the shipped data_options.yaml trains with batch 8 and validates with batch 64.  The validation loss must use
    the positive rate of the WHOLE validation batch (utils/losses.py:24-27,87-95), not of chunks sized by the training batch."""
    from dnncancerannotator_amd import data, engine
    cfg = {'model': 'UNetAnnotator', 'model_options': UNET,
           'deploy_options': {'optimizer': 'adam', 'loss': {'class_name': 'WeightedCrossentropy', 'config': {'weight_mul': 3.0}},
                              'enable_multigpu': False, 'metrics': [{'FBetaScore': {'thresholds': 0.5, 'beta': 1.0, 'name': 'pixel/F1-score'}}]}}
    H = W = 32
    xv, yv = O.synthetic_batch(6, H, W, 1, seed_x=7, seed_y=8)
    yv[3:] = 0.0                                   # half of the validation batch has no positives at all
    val = data.ArrayDataset(xv, yv, 6)
    train = data.SyntheticDataset(2, H, W, 1, n_batches=2, seed=3)
    e = engine.TFKerasModel(cfg)
    res = e.train(train, val_data=val, save_path=str(tmp_path / 'run'), max_steps=2, save_freq=2)
    assert e.device_model.max_batch == 6           # sized for the validation batch up front
    spec = O.ModelSpec('unet', 1, **UNET)
    params = O.unflatten(spec, e.device_model.get_params().astype(np.float64))
    _, logits = O.predict(spec, params, xv.astype(np.float64))
    per, _ = O.weighted_crossentropy(yv, logits, weight_mul=3.0)
    assert res.history['val_loss'][-1] == pytest.approx(float(per.mean()), rel=1e-5)
    chunked = np.mean([O.weighted_crossentropy(yv[i:i + 2], logits[i:i + 2], weight_mul=3.0)[0].mean() for i in (0, 2, 4)])
    assert abs(chunked - per.mean()) > 1e-3 * per.mean()       # the per-chunk weighting really is a different number
    # a model built for a small batch grows on demand (evaluate after train in one process), state carried over
    e2 = engine.TFKerasModel(cfg)
    e2._build(train)
    e2.load(str(tmp_path / 'run' / 'checkpoints' / 'ckpt-2'))
    assert e2.device_model.max_batch == 2
    r2 = e2._evaluate(val)
    assert e2.device_model.max_batch == 6 and r2['loss'] == pytest.approx(float(per.mean()), rel=1e-5)
    assert e2.device_model.get_opt_state()[2] == 2
