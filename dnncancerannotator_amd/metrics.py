"""Pixel metrics on top of device-side confusion counts (utils/metrics.py:19-77 + the Keras metrics named in
configs/additionals/metrics.yaml:2-23).  TP/FP/FN/TN are counted on the GPU (dnnca_pixel_confusion: prediction > threshold,
the Keras Precision/Recall convention); region-based metrics (metrics.py:80-510: connected components on the CPU) are
outside the accelerated path and are skipped with a warning."""

import logging

import numpy as np


def _div_no_nan(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.where(b > 0, a / np.where(b > 0, b, 1.0), 0.0)


class _ConfusionMetric:
    def __init__(self, thresholds, name):
        self.thresholds = np.atleast_1d(np.asarray(thresholds, np.float32))
        self.name = name
        self.reset_state()

    def reset_state(self):
        self.counts = np.zeros((len(self.thresholds), 4), np.float64)     # tp, fp, fn, tn

    def update_state(self, device_model, y):
        self.counts += np.asarray(device_model.pixel_confusion(y, self.thresholds), np.float64)

    def merge(self, reduce_fn):
        self.counts = reduce_fn(self.counts)


class Precision(_ConfusionMetric):
    def __init__(self, thresholds=0.5, name='precision', **kw):
        super().__init__(thresholds, name)

    def result(self):
        tp, fp = self.counts[:, 0], self.counts[:, 1]
        r = _div_no_nan(tp, tp + fp)
        return float(r[0]) if len(r) == 1 else r


class Recall(_ConfusionMetric):
    def __init__(self, thresholds=0.5, name='recall', **kw):
        super().__init__(thresholds, name)

    def result(self):
        tp, fn = self.counts[:, 0], self.counts[:, 2]
        r = _div_no_nan(tp, tp + fn)
        return float(r[0]) if len(r) == 1 else r


class FBetaScore(_ConfusionMetric):
    """utils/metrics.py:37-61: (1 + b^2) P R / (b^2 P + R + eps); beta = 1 is the pixel Dice / F1."""

    def __init__(self, beta, thresholds, epsilon=1e-07, name='fbeta', **kw):
        assert beta > 0
        super().__init__(thresholds, name)
        self.beta, self.epsilon = beta, epsilon

    def result(self):
        tp, fp, fn = self.counts[:, 0], self.counts[:, 1], self.counts[:, 2]
        p, r = _div_no_nan(tp, tp + fp), _div_no_nan(tp, tp + fn)
        s = (1 + self.beta ** 2) * p * r / (self.beta ** 2 * p + r + self.epsilon)
        return float(s[0]) if len(s) == 1 else s


class AUC(_ConfusionMetric):
    """tf.keras.metrics.AUC(curve, num_thresholds) [TF-2.6]: ROC by trapezoids, PR by the Davis-Goadrich interpolation."""

    def __init__(self, curve='ROC', num_thresholds=200, name='auc', **kw):
        eps = 1e-7
        thr = [0.0 - eps] + [(i + 1) / (num_thresholds - 1) for i in range(num_thresholds - 2)] + [1.0 + eps]
        super().__init__(thr, name)
        self.curve = curve.upper()

    def result(self):
        tp, fp, fn, tn = (self.counts[:, i] for i in range(4))
        if self.curve == 'ROC':
            recall, fpr = _div_no_nan(tp, tp + fn), _div_no_nan(fp, fp + tn)
            return float(np.sum((fpr[:-1] - fpr[1:]) * (recall[:-1] + recall[1:]) / 2.0))
        dtp = tp[:-1] - tp[1:]
        p = tp + fp
        dp = p[:-1] - p[1:]
        slope = _div_no_nan(dtp, np.maximum(dp, 0))
        intercept = tp[1:] - slope * p[1:]
        ok = (p[:-1] > 0) & (p[1:] > 0)
        ratio = np.where(ok, _div_no_nan(p[:-1], np.maximum(p[1:], 0)), 1.0)
        inc = _div_no_nan(slope * (dtp + intercept * np.log(ratio)), np.maximum(tp[1:] + fn[1:], 0))
        return float(np.sum(inc))


_REGISTRY = {'Precision': Precision, 'Recall': Recall, 'AUC': AUC, 'FBetaScore': FBetaScore}


def solve_metric(metric_spec):
    """utils/metrics.py:19-34: {ClassName: {kwargs}} -> metric instance (None for metrics outside the hot path)."""
    if isinstance(metric_spec, str):
        metric_spec = {metric_spec: {}}
    if not isinstance(metric_spec, dict):
        raise ValueError
    assert len(metric_spec) == 1
    name, options = list(metric_spec.items())[0]
    options = dict(options or {})
    if name.startswith('RegionBased'):
        logging.warning('metric %s is a CPU region metric outside the accelerated path: skipped', name)
        return None
    if name not in _REGISTRY:
        raise ValueError(f'Unknown metric: {name}')
    return _REGISTRY[name](**options)
