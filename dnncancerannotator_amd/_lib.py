"""ctypes binding of libdnnca.so (include/dnnca.h).  The only bridge between the Python host and the HIP engine.

There is no CPU fallback: if the library is missing or no GPU is present, the calls fail loudly."""

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('DNNCA_LIB') or os.path.join(HERE, 'libdnnca.so')      # DNNCA_LIB: A/B of two builds (development aid)

OK = 0
ARCH_UNET, ARCH_MULMO = 0, 1
PAD_VALID, PAD_SAME = 0, 1
F32, BF16 = 0, 1
UNIQUE_ID_BYTES = 128
FLAG_GENERIC = 1


class DnncaError(RuntimeError):
    def __init__(self, code, message):
        super().__init__('libdnnca error %d: %s' % (code, message))
        self.code = code


class ModelDesc(C.Structure):
    _fields_ = [
        ('arch', C.c_int32), ('in_channels', C.c_int32), ('height', C.c_int32), ('width', C.c_int32),
        ('max_batch', C.c_int32), ('n_filters_first', C.c_int32), ('n_downsample', C.c_int32), ('rate', C.c_int32),
        ('kernel_size', C.c_int32), ('conv_stride', C.c_int32), ('bn', C.c_int32), ('padding', C.c_int32),
        ('reference_index', C.c_int32), ('n_conv', C.c_int32), ('leaky_alpha', C.c_float), ('l2', C.c_float),
        ('dtype', C.c_int32), ('flags', C.c_int32),
    ]


class LossCfg(C.Structure):
    _fields_ = [('has_weight', C.c_int32), ('weight', C.c_float), ('weight_add', C.c_float), ('weight_mul', C.c_float),
                ('label_smoothing', C.c_int32), ('label_smoothing_filter_size', C.c_int32), ('label_smoothing_sigma', C.c_float)]


class StepOut(C.Structure):
    _fields_ = [('loss', C.c_float), ('positive_rate', C.c_float), ('weight', C.c_float),
                ('label_min', C.c_float), ('label_max', C.c_float)]


class AugParam(C.Structure):                       # dnnca_aug_param
    _fields_ = [('dy', C.c_int32), ('dx', C.c_int32), ('flip', C.c_int32), ('contrast', C.c_float)]


class Confusion(C.Structure):
    _fields_ = [('tp', C.c_double), ('fp', C.c_double), ('fn', C.c_double), ('tn', C.c_double)]


_FP = C.POINTER(C.c_float)
_VP = C.c_void_p

# name -> (restype, argtypes); mirrors include/dnnca.h one to one
SIGNATURES = {
    'dnnca_version': (C.c_char_p, []),
    'dnnca_last_error': (C.c_char_p, []),
    'dnnca_device_count': (C.c_int, [C.POINTER(C.c_int)]),
    'dnnca_init': (C.c_int, [C.c_int]),
    'dnnca_model_create': (C.c_int, [C.POINTER(ModelDesc), C.POINTER(_VP)]),
    'dnnca_model_destroy': (C.c_int, [_VP]),
    'dnnca_param_count': (C.c_int, [_VP, C.POINTER(C.c_int)]),
    'dnnca_param_info': (C.c_int, [_VP, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_int64), C.POINTER(C.c_int),
                                   C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    'dnnca_num_trainable': (C.c_int, [_VP, C.POINTER(C.c_int64)]),
    'dnnca_num_state': (C.c_int, [_VP, C.POINTER(C.c_int64)]),
    'dnnca_set_params': (C.c_int, [_VP, _FP, C.c_int64]),
    'dnnca_get_params': (C.c_int, [_VP, _FP, C.c_int64]),
    'dnnca_set_state': (C.c_int, [_VP, _FP, C.c_int64]),
    'dnnca_get_state': (C.c_int, [_VP, _FP, C.c_int64]),
    'dnnca_get_grads': (C.c_int, [_VP, _FP, C.c_int64]),
    'dnnca_set_opt_state': (C.c_int, [_VP, _FP, _FP, C.c_int64, C.c_int64]),
    'dnnca_get_opt_state': (C.c_int, [_VP, _FP, _FP, C.c_int64, C.POINTER(C.c_int64)]),
    'dnnca_set_adam': (C.c_int, [_VP, C.c_float, C.c_float, C.c_float]),
    'dnnca_forward': (C.c_int, [_VP, _FP, C.c_int, C.c_int, _FP, _FP]),
    'dnnca_train_step': (C.c_int, [_VP, _FP, _FP, C.c_int, C.c_float, C.POINTER(LossCfg), C.POINTER(StepOut)]),
    'dnnca_eval_step': (C.c_int, [_VP, _FP, _FP, C.c_int, C.POINTER(LossCfg), C.POINTER(StepOut), _FP]),
    'dnnca_dev_alloc': (C.c_int, [C.POINTER(_VP), C.c_size_t]),
    'dnnca_dev_free': (C.c_int, [_VP]),
    'dnnca_memcpy_h2d': (C.c_int, [_VP, _VP, C.c_size_t]),
    'dnnca_warp_f32': (C.c_int, [_VP, _VP, _VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                 _VP, _VP]),
    'dnnca_augment_u8': (C.c_int, [_VP, _VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint, C.POINTER(AugParam), C.c_int, C.c_int,
                                   _VP, _VP]),
    'dnnca_memcpy_d2h': (C.c_int, [_VP, _VP, C.c_size_t]),
    'dnnca_train_step_dev': (C.c_int, [_VP, _VP, _VP, C.c_int, C.c_float, C.POINTER(LossCfg), C.POINTER(StepOut)]),
    'dnnca_forward_dev': (C.c_int, [_VP, _VP, C.c_int, C.c_int]),
    'dnnca_last_step_out': (C.c_int, [_VP, C.POINTER(StepOut)]),
    'dnnca_sync': (C.c_int, [_VP]),
    'dnnca_crc32c': (C.c_int, [_VP, C.c_size_t, C.POINTER(C.c_uint32)]),
    'dnnca_stage_init': (C.c_int, [_VP, C.c_int, C.c_size_t]),
    'dnnca_stage_upload': (C.c_int, [_VP, C.c_int, _VP, C.c_size_t, _VP, C.c_size_t, C.POINTER(_VP), C.POINTER(_VP)]),
    'dnnca_stage_uploaded': (C.c_int, [_VP, C.c_int]),
    'dnnca_stage_wait': (C.c_int, [_VP, C.c_int]),
    'dnnca_train_step_staged': (C.c_int, [_VP, C.c_int, _VP, _VP, C.c_int, C.c_float, C.POINTER(LossCfg)]),
    'dnnca_staged_out': (C.c_int, [_VP, C.c_int, C.POINTER(StepOut)]),
    'dnnca_eval_begin': (C.c_int, [_VP, _VP, C.c_int]),
    'dnnca_eval_step_staged': (C.c_int, [_VP, C.c_int, _VP, _VP, C.c_int, C.POINTER(LossCfg)]),
    'dnnca_eval_end': (C.c_int, [_VP, C.POINTER(Confusion)]),
    'dnnca_pixel_confusion': (C.c_int, [_VP, _FP, C.c_int, _FP, C.c_int, C.POINTER(Confusion)]),
    'dnnca_pixel_confusion_of': (C.c_int, [_VP, _FP, _FP, C.c_int64, _FP, C.c_int, C.POINTER(Confusion)]),
    'dnnca_comm_unique_id': (C.c_int, [_VP]),
    'dnnca_comm_init': (C.c_int, [_VP, C.c_int, C.c_int, _VP, C.c_size_t]),
    'dnnca_comm_world': (C.c_int, [_VP, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    'dnnca_comm_collectives': (C.c_int, [_VP, C.POINTER(C.c_int)]),
    'dnnca_comm_broadcast_weights': (C.c_int, [_VP, C.c_int]),
    'dnnca_comm_average_state': (C.c_int, [_VP]),
    'dnnca_comm_allreduce_host': (C.c_int, [_VP, C.POINTER(C.c_double), C.c_int64, C.c_int]),
    'dnnca_timer_start': (C.c_int, [_VP]),
    'dnnca_timer_stop': (C.c_int, [_VP, C.POINTER(C.c_float)]),
    'dnnca_profile_enable': (C.c_int, [_VP, C.c_int]),
    'dnnca_profile_focus': (C.c_int, [_VP, C.c_char_p]),
    'dnnca_profile_sample': (C.c_int, [_VP, C.c_int]),
    'dnnca_profile_reset': (C.c_int, [_VP]),
    'dnnca_profile_count': (C.c_int, [_VP, C.POINTER(C.c_int)]),
    'dnnca_profile_get': (C.c_int, [_VP, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                    C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    'dnnca_plan_dump': (C.c_int, [_VP, C.c_char_p, C.c_size_t]),
}

_lib = None


def _hip_runtime_mapped():
    """Is a HIP runtime already mapped into this process (somebody else may have initialised it)?"""
    try:
        with open('/proc/self/maps') as f:
            return any('libamdhip64' in line for line in f)
    except OSError:
        return False


def load():
    """dlopen libdnnca.so and declare every prototype.  Raises if the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError('%s is missing: run `python -m dnncancerannotator_amd.build` (hipcc, gfx950). '
                          'There is no CPU fallback.' % LIB_PATH)
    # Multi-process GPU work on this platform needs dmabuf IPC: with the legacy mode RCCL's communicator set-up fails with
    # `hipIpcGetMemHandle: invalid argument` (the host driver only supports dmabuf).  The ROCm runtime reads the variable when it
    # initialises, i.e. at the first HIP call behind this dlopen -- so this is the ONE place every process of the package passes
    # through in time: workers of `python -m dnncancerannotator_amd.launch`, ranks started by torch.distributed.run (bench.py),
    # single-GPU runs.  A value the caller exported wins.
    if 'HSA_ENABLE_IPC_MODE_LEGACY' not in os.environ:
        os.environ['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
        if _hip_runtime_mapped():
            import warnings
            warnings.warn('libamdhip64 was already loaded when dnncancerannotator_amd set HSA_ENABLE_IPC_MODE_LEGACY=0: if the ROCm '
                          'runtime has initialised, the setting is ignored and multi-process RCCL set-up may fail '
                          '(hipIpcGetMemHandle: invalid argument).  Export it before starting the process.', RuntimeWarning)
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code):
    if code != OK:
        raise DnncaError(code, load().dnnca_last_error().decode(errors='replace'))


def fptr(a):
    return a.ctypes.data_as(_FP)


def as_f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)
