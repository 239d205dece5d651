"""DeviceModel: one annotator network living on one MI355X, driven through the C ABI (include/dnnca.h).

This is the object that replaces the compiled tf.keras.Model of the reference (engine.py:254-288): it owns the
weights, the Adam slots and the activations in HBM and exposes forward / train_step / eval_step."""

import ctypes as C
from collections import OrderedDict

import numpy as np

from . import _lib
from ._lib import check, fptr, as_f32


def init_device(ordinal=0):
    check(_lib.load().dnnca_init(int(ordinal)))


def device_count():
    n = C.c_int(0)
    lib = _lib.load()
    if lib.dnnca_device_count(C.byref(n)) != 0:
        return 0
    return n.value


class DeviceBuffer:
    """A raw HBM allocation holding a float32 array (device-resident batches for the benchmark loop)."""

    def __init__(self, array):
        array = as_f32(array)
        self.shape = array.shape
        self.nbytes = array.nbytes
        self.ptr = C.c_void_p()
        check(_lib.load().dnnca_dev_alloc(C.byref(self.ptr), self.nbytes))
        check(_lib.load().dnnca_memcpy_h2d(self.ptr, array.ctypes.data_as(C.c_void_p), self.nbytes))

    def to_host(self):
        out = np.empty(self.shape, np.float32)
        check(_lib.load().dnnca_memcpy_d2h(out.ctypes.data_as(C.c_void_p), self.ptr, self.nbytes))
        return out

    def free(self):
        if self.ptr:
            _lib.load().dnnca_dev_free(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class RawDeviceBuffer:
    """An HBM allocation of `nbytes` bytes (uint8 source batches of the device-side augmentation; grown on demand)."""

    def __init__(self):
        self.ptr, self.nbytes = C.c_void_p(), 0

    def reserve(self, nbytes):
        if nbytes > self.nbytes:
            self.free()
            check(_lib.load().dnnca_dev_alloc(C.byref(self.ptr), nbytes))
            self.nbytes = nbytes

    def upload(self, array):
        array = np.ascontiguousarray(array)
        self.reserve(array.nbytes)
        check(_lib.load().dnnca_memcpy_h2d(self.ptr, array.ctypes.data_as(C.c_void_p), array.nbytes))

    def free(self):
        if self.ptr:
            _lib.load().dnnca_dev_free(self.ptr)
            self.ptr, self.nbytes = C.c_void_p(), 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceView:
    """A float32 tensor inside a RawDeviceBuffer (what train_step_dev takes: .ptr and .shape)."""

    def __init__(self, buf, shape):
        self.buf, self.ptr, self.shape = buf, buf.ptr, tuple(shape)
        self.nbytes = int(np.prod(shape)) * 4

    def to_host(self):
        out = np.empty(self.shape, np.float32)
        check(_lib.load().dnnca_memcpy_d2h(out.ctypes.data_as(C.c_void_p), self.ptr, self.nbytes))
        return out


class StagingRing:
    """The model's staging slots in HBM + its copy stream (include/dnnca.h, dnnca_stage_*): what ds.prefetch + Keras' asynchronous
    input feeding are for the reference (annotator/data.py:110,143; engine.py:126-135).  `upload` may run on a second thread."""

    MAX_SLOTS = 8          # engine: slots 0-3 feed the train steps, 4-7 the evaluation / validation steps

    def __init__(self, dm, slots=8, slot_bytes=0):
        self.dm, self.slots = dm, int(slots)
        check(dm.lib.dnnca_stage_init(dm.handle, self.slots, int(slot_bytes)))
        x_bytes = dm.max_batch * int(np.prod(dm.in_shape)) * 4
        y_bytes = dm.max_batch * dm.in_shape[0] * dm.in_shape[1] * 4
        self.slot_bytes = max(int(slot_bytes), ((x_bytes + 255) & ~255) + y_bytes)

    def fits(self, a, b=None):
        return ((a.nbytes + 255) & ~255) + (0 if b is None else b.nbytes) <= self.slot_bytes

    def upload(self, slot, a, b=None, wait=True):
        """host arrays -> the slot, on the copy stream (behind the step that read the slot's previous content); returns the device
        addresses (c_void_p) of a and b.  wait=True blocks the CALLING thread until the copy has completed, so the arrays may be
        reused or freed afterwards; with wait=False the caller keeps them alive and unchanged until the slot's step has run."""
        a = np.ascontiguousarray(a)
        pa, pb = C.c_void_p(), C.c_void_p()
        if b is None:
            check(self.dm.lib.dnnca_stage_upload(self.dm.handle, int(slot), a.ctypes.data_as(C.c_void_p), a.nbytes, None, 0,
                                                 C.byref(pa), C.byref(pb)))
            pb = None
        else:
            b = np.ascontiguousarray(b)
            check(self.dm.lib.dnnca_stage_upload(self.dm.handle, int(slot), a.ctypes.data_as(C.c_void_p), a.nbytes,
                                                 b.ctypes.data_as(C.c_void_p), b.nbytes, C.byref(pa), C.byref(pb)))
        if wait:
            check(self.dm.lib.dnnca_stage_uploaded(self.dm.handle, int(slot)))
        return pa, pb

    def wait(self, slot):
        """the model's stream waits for the slot's upload (before kernels other than the train step read it)"""
        check(self.dm.lib.dnnca_stage_wait(self.dm.handle, int(slot)))

    def train_step(self, slot, x_ptr, y_ptr, batch, lr, cfg):
        """asynchronous train step on device-resident (x, y) that depend on the slot's upload; outputs: out(slot)"""
        check(self.dm.lib.dnnca_train_step_staged(self.dm.handle, int(slot), x_ptr, y_ptr, int(batch), float(lr), C.byref(cfg)))

    # keras Model.evaluate (engine.py:198-203) over the ring: the confusion histogram of ALL thresholds stays on the device
    def eval_begin(self, thresholds=()):
        thr = as_f32(np.asarray(thresholds, np.float32)).ravel()
        self._n_thr = int(thr.size)
        check(self.dm.lib.dnnca_eval_begin(self.dm.handle, thr.ctypes.data_as(C.c_void_p) if thr.size else None, self._n_thr))

    def eval_step(self, slot, x_ptr, y_ptr, batch, cfg):
        check(self.dm.lib.dnnca_eval_step_staged(self.dm.handle, int(slot), x_ptr, y_ptr, int(batch), C.byref(cfg)))

    def eval_end(self):
        """[(tp, fp, fn, tn)] per threshold of eval_begin, summed over every eval_step since (exact integers)"""
        out = (_lib.Confusion * max(self._n_thr, 1))()
        check(self.dm.lib.dnnca_eval_end(self.dm.handle, out))
        return [(c.tp, c.fp, c.fn, c.tn) for c in out[:self._n_thr]]

    def out(self, slot):
        """waits for the step that last ran on the slot; raises what train_step would have raised (label / weight assertions)"""
        out = _lib.StepOut()
        check(self.dm.lib.dnnca_staged_out(self.dm.handle, int(slot), C.byref(out)))
        return out


class DeviceModel:
    def __init__(self, arch, in_channels, height, width, max_batch, n_filters_first, n_downsample, rate=2, kernel_size=3,
                 conv_stride=1, bn=False, padding='valid', leaky_alpha=0.0, l2=0.0, reference_index=0, n_conv=2,
                 dtype='f32', force_generic=False):
        self.lib = _lib.load()
        d = _lib.ModelDesc()
        d.arch = {'unet': _lib.ARCH_UNET, 'mulmo': _lib.ARCH_MULMO}[arch]
        d.in_channels, d.height, d.width, d.max_batch = int(in_channels), int(height), int(width), int(max_batch)
        d.n_filters_first, d.n_downsample, d.rate = int(n_filters_first), int(n_downsample), int(rate)
        d.kernel_size, d.conv_stride, d.bn = int(kernel_size), int(conv_stride), int(bool(bn))
        d.padding = {'same': _lib.PAD_SAME, 'valid': _lib.PAD_VALID}[padding]
        d.reference_index, d.n_conv = int(reference_index), int(n_conv)
        d.leaky_alpha, d.l2 = float(leaky_alpha), float(l2)
        d.dtype = {'f32': _lib.F32, 'bf16': _lib.BF16}[dtype]
        d.flags = _lib.FLAG_GENERIC if force_generic else 0
        self.desc = d
        self.handle = C.c_void_p()
        check(self.lib.dnnca_model_create(C.byref(d), C.byref(self.handle)))
        self.in_shape = (int(height), int(width), int(in_channels))
        self.max_batch = int(max_batch)
        n = C.c_int64()
        check(self.lib.dnnca_num_trainable(self.handle, C.byref(n)))
        self.n_trainable = n.value
        check(self.lib.dnnca_num_state(self.handle, C.byref(n)))
        self.n_state = n.value

    # ---- life-cycle -------------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, 'handle', None):
            self.lib.dnnca_model_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- variables --------------------------------------------------------------------------------------------
    def param_infos(self):
        """[(name, shape, trainable, offset)] in Keras creation order."""
        cnt = C.c_int()
        check(self.lib.dnnca_param_count(self.handle, C.byref(cnt)))
        out = []
        name = C.create_string_buffer(256)
        shape = (C.c_int64 * 4)()
        ndim, tr, off = C.c_int(), C.c_int(), C.c_int64()
        for i in range(cnt.value):
            check(self.lib.dnnca_param_info(self.handle, i, name, 256, shape, C.byref(ndim), C.byref(tr), C.byref(off)))
            out.append((name.value.decode(), tuple(shape[k] for k in range(ndim.value)), bool(tr.value), off.value))
        return out

    def _get(self, fn, n):
        out = np.empty(n, np.float32)
        check(fn(self.handle, fptr(out), n))
        return out

    def get_params(self):
        return self._get(self.lib.dnnca_get_params, self.n_trainable)

    def get_state(self):
        return self._get(self.lib.dnnca_get_state, self.n_state)

    def get_grads(self):
        return self._get(self.lib.dnnca_get_grads, self.n_trainable)

    def set_params(self, flat):
        flat = as_f32(flat).ravel()
        check(self.lib.dnnca_set_params(self.handle, fptr(flat), flat.size))

    def set_state(self, flat):
        flat = as_f32(flat).ravel()
        check(self.lib.dnnca_set_state(self.handle, fptr(flat), flat.size))

    def get_opt_state(self):
        m = np.empty(self.n_trainable, np.float32)
        v = np.empty(self.n_trainable, np.float32)
        it = C.c_int64()
        check(self.lib.dnnca_get_opt_state(self.handle, fptr(m), fptr(v), self.n_trainable, C.byref(it)))
        return m, v, it.value

    def set_opt_state(self, m, v, iterations):
        m, v = as_f32(m).ravel(), as_f32(v).ravel()
        check(self.lib.dnnca_set_opt_state(self.handle, fptr(m), fptr(v), m.size, int(iterations)))

    def set_adam(self, beta1=0.9, beta2=0.999, epsilon=1e-7):
        check(self.lib.dnnca_set_adam(self.handle, beta1, beta2, epsilon))

    def named_params(self):
        """OrderedDict name -> ndarray (trainable and state variables)."""
        p, s = self.get_params(), self.get_state()
        out = OrderedDict()
        for name, shape, tr, off in self.param_infos():
            src = p if tr else s
            out[name] = src[off:off + int(np.prod(shape))].reshape(shape).copy()
        return out

    def init_glorot(self, seed=None):
        """Keras default initialisers: glorot_uniform kernels, zero biases (BN defaults are set by the library)."""
        rng = np.random.default_rng(seed)
        flat = self.get_params()
        for name, shape, tr, off in self.param_infos():
            if tr and name.endswith('.kernel'):
                kh, kw, a, b = shape
                limit = np.sqrt(6.0 / (kh * kw * (a + b)))
                flat[off:off + kh * kw * a * b] = rng.uniform(-limit, limit, kh * kw * a * b).astype(np.float32)
        self.set_params(flat)

    # ---- hot path ---------------------------------------------------------------------------------------------
    @staticmethod
    def loss_cfg(weight=None, weight_add=0.0, weight_mul=1.0, label_smoothing=False, label_smoothing_filter_size=6,
                 label_smoothing_sigma=3, **ignored):
        c = _lib.LossCfg()
        c.has_weight = 0 if weight is None else 1
        c.weight = 0.0 if weight is None else float(weight)
        c.weight_add, c.weight_mul = float(weight_add), float(weight_mul)
        c.label_smoothing = int(bool(label_smoothing))
        c.label_smoothing_filter_size, c.label_smoothing_sigma = int(label_smoothing_filter_size), float(label_smoothing_sigma)
        return c

    def _check_x(self, x):
        x = as_f32(x)
        if x.ndim != 4 or x.shape[1:] != self.in_shape:
            raise ValueError('x shape %s does not match the built input %s' % (x.shape, ('B',) + self.in_shape))
        if not 1 <= x.shape[0] <= self.max_batch:
            raise ValueError('batch %d outside [1, %d]' % (x.shape[0], self.max_batch))
        return x

    def forward(self, x, training=False, return_logits=False):
        x = self._check_x(x)
        B = x.shape[0]
        prob = np.empty((B, self.in_shape[0], self.in_shape[1], 1), np.float32)
        logits = np.empty_like(prob) if return_logits else None
        check(self.lib.dnnca_forward(self.handle, fptr(x), B, int(training), fptr(prob),
                                     fptr(logits) if return_logits else None))
        return (prob, logits) if return_logits else prob

    def train_step(self, x, y, lr, cfg):
        x = self._check_x(x)
        y = as_f32(y)
        if y.shape != x.shape[:3]:
            raise ValueError('y shape %s does not match x %s' % (y.shape, x.shape))
        out = _lib.StepOut()
        check(self.lib.dnnca_train_step(self.handle, fptr(x), fptr(y), x.shape[0], float(lr), C.byref(cfg), C.byref(out)))
        return out

    def eval_step(self, x, y, cfg, return_prob=False):
        x = self._check_x(x)
        y = as_f32(y)
        if y.shape != x.shape[:3]:
            raise ValueError('y shape %s does not match x %s' % (y.shape, x.shape))
        out = _lib.StepOut()
        prob = np.empty(x.shape[:3] + (1,), np.float32) if return_prob else None
        check(self.lib.dnnca_eval_step(self.handle, fptr(x), fptr(y), x.shape[0], C.byref(cfg), C.byref(out),
                                       fptr(prob) if return_prob else None))
        return (out, prob) if return_prob else out

    def check_dev(self, xbuf, ybuf, batch):
        """Device-resident (x, y) that carry their shapes (DeviceBuffer, DeviceView) must match the built model: the staged and the
        *_dev entry points take bare pointers, a mismatched batch would be read as misaligned memory (the host entry points check in
        _check_x).  Raises ValueError like them."""
        xs, ys = getattr(xbuf, 'shape', None), getattr(ybuf, 'shape', None)
        if xs is None:
            return
        if len(xs) != 4 or tuple(xs[1:]) != self.in_shape:
            raise ValueError('x shape %s does not match the built input %s' % (tuple(xs), ('B',) + self.in_shape))
        if ys is not None and tuple(ys) != tuple(xs[:3]):
            raise ValueError('y shape %s does not match x %s' % (tuple(ys), tuple(xs)))
        if not 1 <= int(batch) <= min(int(xs[0]), self.max_batch):
            raise ValueError('batch %d outside [1, %d]' % (batch, min(int(xs[0]), self.max_batch)))

    def train_step_dev(self, xbuf, ybuf, batch, lr, cfg, want_out=False):
        self.check_dev(xbuf, ybuf, batch)
        out = _lib.StepOut() if want_out else None
        check(self.lib.dnnca_train_step_dev(self.handle, xbuf.ptr, ybuf.ptr, int(batch), float(lr), C.byref(cfg),
                                            C.byref(out) if want_out else None))
        return out

    def last_step_out(self):
        out = _lib.StepOut()
        check(self.lib.dnnca_last_step_out(self.handle, C.byref(out)))
        return out

    def sync(self):
        check(self.lib.dnnca_sync(self.handle))

    def staging(self, slots=StagingRing.MAX_SLOTS, slot_bytes=0):
        """the model's StagingRing (created on first use; one per model)"""
        if getattr(self, '_ring', None) is None:
            self._ring = StagingRing(self, slots, slot_bytes)
        return self._ring

    def pixel_confusion(self, y, thresholds):
        y = as_f32(y)
        thr = as_f32(thresholds).ravel()
        out = (_lib.Confusion * thr.size)()
        check(self.lib.dnnca_pixel_confusion(self.handle, fptr(y), y.shape[0], fptr(thr), thr.size, out))
        return [(c.tp, c.fp, c.fn, c.tn) for c in out]

    def pixel_confusion_of(self, prob, y, thresholds):
        """Metric.update_state(y_true, y_pred) on caller-supplied probabilities: [(tp, fp, fn, tn)] per threshold (exact)."""
        prob, y = as_f32(prob).ravel(), as_f32(y).ravel()
        if prob.size != y.size:
            raise ValueError('prob and y differ in size: %d vs %d' % (prob.size, y.size))
        thr = as_f32(thresholds).ravel()
        out = (_lib.Confusion * thr.size)()
        check(self.lib.dnnca_pixel_confusion_of(self.handle, fptr(prob), fptr(y), prob.size, fptr(thr), thr.size, out))
        return [(c.tp, c.fp, c.fn, c.tn) for c in out]

    # ---- device-side augmentation (annotator/data.py:62-111 train_ds) ------------------------------------------
    def augment_u8(self, raw, params, out_size, label_index, contrast_channels=None, src_ptr=None):
        """raw uint8 [B, Hs, Ws, Cs] (host) + per-image draws [(dy, dx, flip, contrast)] -> device-resident (x [B, Ho, Wo, Cs-1],
        y [B, Ho, Wo]) views, valid until the next call.  contrast_channels: source channels to adjust (default: all features).
        src_ptr: the batch is in HBM already (a StagingRing slot; `raw` is then only read for its shape)."""
        if src_ptr is None:
            raw = np.ascontiguousarray(raw, np.uint8)
        if raw.ndim != 4:
            raise ValueError('raw batch must be [B, H, W, C] uint8, got %s' % (raw.shape,))
        B, hs, ws, cs = raw.shape
        ho, wo = int(out_size[0]), int(out_size[1])
        if len(params) != B:
            raise ValueError('%d parameter rows for %d images' % (len(params), B))
        if contrast_channels is None:
            contrast_channels = [c for c in range(cs) if c != label_index]
        mask = 0
        for c in contrast_channels:
            mask |= 1 << int(c)
        if not hasattr(self, '_aug'):
            self._aug = (RawDeviceBuffer(), RawDeviceBuffer(), RawDeviceBuffer())
        src, xb, yb = self._aug
        own_upload = src_ptr is None
        if own_upload:
            src.upload(raw)
            src_ptr = src.ptr
        xb.reserve(B * ho * wo * (cs - 1) * 4)
        yb.reserve(B * ho * wo * 4)
        prm = (_lib.AugParam * B)(*[_lib.AugParam(int(p[0]), int(p[1]), int(p[2]), float(p[3])) for p in params])
        check(self.lib.dnnca_augment_u8(self.handle, src_ptr, B, hs, ws, cs, int(label_index), mask, prm, ho, wo, xb.ptr, yb.ptr))
        if own_upload:
            # dnnca_augment_u8 is asynchronous on the model's stream (a staged batch keeps the loop one step ahead); the batch this
            # call uploaded itself lives in a buffer the next call overwrites with a blocking copy, so here the stream is drained
            self.sync()
        return DeviceView(xb, (B, ho, wo, cs - 1)), DeviceView(yb, (B, ho, wo))

    def warp(self, xv, yv, ctrl, wv):
        """random_warp's dense part on device-resident (x, y) views: ctrl [B, n, 2], wv [B, n + 3, 2] from augment.solve_warp."""
        B, h, w, c = xv.shape
        ctrl, wv = np.ascontiguousarray(ctrl, np.float64), np.ascontiguousarray(wv, np.float64)
        dptr = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))        # noqa: E731
        if not hasattr(self, '_warp'):
            self._warp = (RawDeviceBuffer(), RawDeviceBuffer())
        xo, yo = self._warp
        xo.reserve(xv.nbytes)
        yo.reserve(yv.nbytes)
        check(self.lib.dnnca_warp_f32(self.handle, xv.ptr, yv.ptr, B, h, w, c, ctrl.shape[1], dptr(ctrl), dptr(wv), xo.ptr, yo.ptr))
        return DeviceView(xo, xv.shape), DeviceView(yo, yv.shape)

    # ---- data parallel ----------------------------------------------------------------------------------------
    @staticmethod
    def comm_unique_id():
        buf = C.create_string_buffer(_lib.UNIQUE_ID_BYTES)
        check(_lib.load().dnnca_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, rank, world, unique_id):
        check(self.lib.dnnca_comm_init(self.handle, int(rank), int(world), unique_id, len(unique_id) if unique_id else 0))

    def comm_collectives(self):
        """gradient all-reduce calls of the last train step (> 1: bucketed)"""
        n = C.c_int()
        check(self.lib.dnnca_comm_collectives(self.handle, C.byref(n)))
        return n.value

    def comm_broadcast_weights(self, root=0):
        check(self.lib.dnnca_comm_broadcast_weights(self.handle, int(root)))

    def comm_average_state(self):
        check(self.lib.dnnca_comm_average_state(self.handle))

    def comm_allreduce(self, values, op='sum'):
        v = np.array(values, dtype=np.float64).ravel()
        check(self.lib.dnnca_comm_allreduce_host(self.handle, v.ctypes.data_as(C.POINTER(C.c_double)), v.size, 1 if op == 'max' else 0))
        return v

    # ---- measurement ------------------------------------------------------------------------------------------
    def timer_start(self):
        check(self.lib.dnnca_timer_start(self.handle))

    def timer_stop(self):
        ms = C.c_float()
        check(self.lib.dnnca_timer_stop(self.handle, C.byref(ms)))
        return ms.value

    def profile_enable(self, mode=1, focus=None, period=1):
        check(self.lib.dnnca_profile_sample(self.handle, int(period)))
        if focus is not None:
            check(self.lib.dnnca_profile_focus(self.handle, focus.encode()))
        check(self.lib.dnnca_profile_enable(self.handle, int(mode)))

    def profile_reset(self):
        check(self.lib.dnnca_profile_reset(self.handle))

    def profile(self):
        """[(kernel, launches, total_ms, algorithmic bytes per launch, flops per launch)]"""
        cnt = C.c_int()
        check(self.lib.dnnca_profile_count(self.handle, C.byref(cnt)))
        name = C.create_string_buffer(128)
        n, ms, by, fl = C.c_int64(), C.c_double(), C.c_double(), C.c_double()
        out = []
        for i in range(cnt.value):
            check(self.lib.dnnca_profile_get(self.handle, i, name, 128, C.byref(n), C.byref(ms), C.byref(by), C.byref(fl)))
            out.append((name.value.decode(), n.value, ms.value, by.value, fl.value))
        return out

    def plan(self, variants=False):
        """The launch schedule of one train step at max_batch: [(kernel, algorithmic bytes, flops)].  variants: keep the template
        variant the library appends to a launch name (`ig_conv_fwd#3n2w8`): the kernel-coverage test tells them apart."""
        buf = C.create_string_buffer(1 << 20)
        check(self.lib.dnnca_plan_dump(self.handle, buf, len(buf)))
        out = []
        for line in buf.value.decode().splitlines():
            k, b, f = line.split('\t')
            out.append((k if variants else k.split('#')[0], float(b), float(f)))
        return out
