"""Model registry -- annotator/models/tf_models/__init__.py:1-2.  engine.py:267-268 resolves the YAML `model:` name with
getattr(tf_models, name)(**model_options); the classes here take the same keyword arguments (unet.py:195-207) and are
materialised on the GPU by `.build(input_shape, max_batch)` (engine.py:93 model.build(element_spec.shape))."""

from . import device


def _solve_activation(identifier):
    """components.py:323-335: 'relu' or {'class_name': 'LeakyReLU', 'config': {'alpha': a}} -> leaky slope (0 = relu)."""
    if isinstance(identifier, str):
        if identifier != 'relu':
            raise ValueError(f'Failed to resolve activation: {identifier} (supported: relu, LeakyReLU)')
        return 0.0
    if isinstance(identifier, dict):
        if identifier.get('class_name') != 'LeakyReLU':
            raise ValueError(f'Failed to resolve activation: {identifier}')
        return float(identifier.get('config', {}).get('alpha', 0.3))
    raise ValueError(f'Failed to resolve activation: {identifier}')


def _solve_regularizer(spec):
    """configs/additionals/kernel_regularizer.yaml:1-4 -> l2 factor."""
    if spec is None:
        return 0.0
    if isinstance(spec, dict) and spec.get('class_name') in ('L2', 'l2'):
        return float(spec.get('config', {}).get('l2', 0.01))
    raise ValueError(f'unsupported kernel_regularizer: {spec}')


class UNetAnnotator:
    """models/tf_models/unet.py:194-282."""
    arch = 'unet'

    def __init__(self, n_filters_first, n_downsample, rate, kernel_size, conv_stride, bn=False, padding='valid',
                 activation='relu', kernel_regularizer=None, **kargs):
        self.configs = dict(n_filters_first=n_filters_first, n_downsample=n_downsample, rate=rate, kernel_size=kernel_size,
                            conv_stride=conv_stride, bn=bn, padding=padding, activation=activation,
                            kernel_regularizer=kernel_regularizer, **kargs)
        self.reference_index = kargs.get('reference_index', 0)
        self.dtype = kargs.get('dtype', 'f32')
        self.device_model = None

    def get_config(self):
        return self.configs

    @classmethod
    def from_config(cls, config):
        return cls(**config)

    def build(self, input_shape, max_batch=None, seed=None, force_generic=False):
        """input_shape = [B or None, H, W, C] (engine.py:93).  Allocates weights/activations in HBM, glorot-initialises."""
        b, h, w, c = input_shape
        c_ = self.configs
        self.device_model = device.DeviceModel(
            self.arch, c, h, w, max_batch or b or 1, c_['n_filters_first'], c_['n_downsample'], rate=c_['rate'],
            kernel_size=c_['kernel_size'], conv_stride=c_['conv_stride'], bn=c_['bn'], padding=c_['padding'],
            leaky_alpha=_solve_activation(c_['activation']), l2=_solve_regularizer(c_['kernel_regularizer']),
            reference_index=self.reference_index, dtype=self.dtype, force_generic=force_generic)
        self.device_model.init_glorot(seed)
        return self.device_model


class MulmoUNetAnnotator(UNetAnnotator):
    """models/tf_models/unet.py:285-300: one encoder per input channel, decoder fed by encoder[reference_index]."""
    arch = 'mulmo'


def _unsupported(name, why):
    class _Unsupported:
        def __init__(self, *a, **k):
            raise NotImplementedError(f'{name}: {why}')
    _Unsupported.__name__ = name
    return _Unsupported


# names exported by the reference that are outside the accelerated hot path
UNet = _unsupported('UNet', 'bare backbone without the annotator head is not a trainable model in the reference configs')
MulmoUNet = _unsupported('MulmoUNet', 'bare backbone without the annotator head is not a trainable model in the reference configs')
MultiResUnet = _unsupported('MultiResUnet', 'MultiResUNet is outside the hot path this engine accelerates')
