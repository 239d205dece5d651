"""Loss registry -- annotator/utils/losses.py.  The arithmetic runs inside libdnnca (fused with the head); this module
only carries the configuration the way `tf.keras.losses.get({'class_name': ..., 'config': ...})` does (engine.py:270-271)."""


class WeightedCrossentropy:
    """utils/losses.py:40-84 TFWeightedCrossentropy: weight = weight_mul * (weight or 1/positive_rate) + weight_add."""

    name = 'weighted_crossentropy'

    def __init__(self, weight=None, weight_add=0.0, weight_mul=1.0, label_smoothing=False,
                 label_smoothing_filter_size=6, label_smoothing_sigma=3):
        if label_smoothing and not (1 <= int(label_smoothing_filter_size) <= 15 and float(label_smoothing_sigma) > 0):
            raise ValueError('label_smoothing: filter size must be in [1, 15] and sigma > 0')
        self.weight = weight
        self.weight_add = weight_add
        self.weight_mul = weight_mul
        self.label_smoothing = label_smoothing
        self.label_smoothing_filter_size = label_smoothing_filter_size
        self.label_smoothing_sigma = label_smoothing_sigma

    def get_config(self):
        return dict(weight=self.weight, weight_add=self.weight_add, weight_mul=self.weight_mul,
                    label_smoothing=self.label_smoothing, label_smoothing_filter_size=self.label_smoothing_filter_size,
                    label_smoothing_sigma=self.label_smoothing_sigma)

    def device_cfg(self):
        # label_smoothing (losses.py:62-67): the Gaussian blur of the labels runs on the device before the loss
        return dict(weight=self.weight, weight_add=self.weight_add, weight_mul=self.weight_mul,
                    label_smoothing=bool(self.label_smoothing), label_smoothing_filter_size=int(self.label_smoothing_filter_size),
                    label_smoothing_sigma=float(self.label_smoothing_sigma))


_REGISTRY = {'WeightedCrossentropy': WeightedCrossentropy, 'weighted_crossentropy': WeightedCrossentropy}


def get(identifier):
    """Mirror of tf.keras.losses.get for the objects registered at utils/losses.py:105-106."""
    if isinstance(identifier, WeightedCrossentropy):
        return identifier
    if isinstance(identifier, str):
        if identifier not in _REGISTRY:
            raise ValueError(f'Unknown loss: {identifier}')
        return _REGISTRY[identifier]()
    if isinstance(identifier, dict):
        name = identifier['class_name']
        if name not in _REGISTRY:
            raise ValueError(f'Unknown loss: {name}')
        return _REGISTRY[name](**identifier.get('config', {}))
    raise ValueError(f'Could not interpret loss identifier: {identifier}')
