"""Test double of dnncancerannotator_amd.device.DeviceModel for the CPU tests of the engine's data-parallel branches:
the numpy oracle stands in for the HIP kernels, torch.distributed (gloo) for RCCL.  Same method surface and the same
per-replica semantics as libdnnca (csrc/model.hip): rank-local positive rate and BatchNorm statistics, ONE all-reduce of
[gradients..., loss] per step, 1/world inside Adam, host all-reduce in doubles.  Test infrastructure only."""

from collections import OrderedDict, namedtuple

import numpy as np

from oracle import unet_oracle as O

StepOut = namedtuple('StepOut', ['loss', 'positive_rate', 'weight', 'label_min', 'label_max'])


class FakeDeviceModel:
    dist = None                      # torch.distributed once a process group exists

    def __init__(self, arch, in_channels, height, width, max_batch, n_filters_first, n_downsample, rate=2, kernel_size=3,
                 conv_stride=1, bn=False, padding='same', leaky_alpha=0.0, l2=0.0, reference_index=0, n_conv=2, dtype='f32',
                 force_generic=False):
        act = {'class_name': 'LeakyReLU', 'config': {'alpha': leaky_alpha}} if leaky_alpha else 'relu'
        reg = {'class_name': 'L2', 'config': {'l2': l2}} if l2 else None
        self.spec = O.ModelSpec(arch, in_channels, n_filters_first, n_downsample, rate=rate, kernel_size=kernel_size, bn=bn,
                                padding=padding, activation=act, kernel_regularizer=reg, reference_index=reference_index)
        self.in_shape, self.max_batch = (height, width, in_channels), int(max_batch)
        self.params = O.init_params(self.spec, seed=0, dtype=np.float64)
        self.n_trainable = len(O.flatten(self.spec, self.params))
        self.n_state = len(O.flatten(self.spec, self.params, trainable=False))
        self.m, self.v, self.iterations = {}, {}, 0
        self.rank, self.world = 0, 1
        self.prob = None
        self.grads = None
        self.calls = []              # (kind, batch) log the tests read

    # ---- variables
    def param_infos(self):
        out, off = [], {True: 0, False: 0}
        for n, s, t in O.param_specs(self.spec):
            out.append((n, tuple(s), t, off[t]))
            off[t] += int(np.prod(s))
        return out

    def get_params(self):
        return O.flatten(self.spec, self.params).astype(np.float32)

    def get_state(self):
        return O.flatten(self.spec, self.params, trainable=False).astype(np.float32)

    def set_params(self, flat):
        O.unflatten(self.spec, np.asarray(flat, np.float64), into=self.params)

    def set_state(self, flat):
        O.unflatten(self.spec, np.asarray(flat, np.float64), trainable=False, into=self.params)

    def get_grads(self):
        return self.grads.astype(np.float32)

    def get_opt_state(self):
        z = OrderedDict((n, np.zeros(s)) for n, s, t in O.param_specs(self.spec) if t)
        m = O.flatten(self.spec, dict(z, **self.m)).astype(np.float32)
        v = O.flatten(self.spec, dict(z, **self.v)).astype(np.float32)
        return m, v, self.iterations

    def set_opt_state(self, m, v, iterations):
        self.m = dict(O.unflatten(self.spec, np.asarray(m, np.float64)))
        self.v = dict(O.unflatten(self.spec, np.asarray(v, np.float64)))
        self.iterations = int(iterations)

    def set_adam(self, beta1=0.9, beta2=0.999, epsilon=1e-7):
        self.adam = (beta1, beta2, epsilon)

    def init_glorot(self, seed=None):
        self.params = O.init_params(self.spec, seed=seed or 0, dtype=np.float64)

    def close(self):
        pass

    # ---- hot path
    @staticmethod
    def loss_cfg(**kw):
        return {k: v for k, v in kw.items() if not (k == 'weight' and v is None)}

    def _check(self, x):
        if not 1 <= len(x) <= self.max_batch:
            raise ValueError('batch %d outside [1, %d]' % (len(x), self.max_batch))

    def train_step(self, x, y, lr, cfg):
        self._check(x)
        self.calls.append(('train', len(x)))
        loss, grads, _, state = O.loss_and_grads(self.spec, self.params, np.asarray(x, np.float64), y, cfg, training=True)
        flat = np.concatenate([O.flatten(self.spec, grads), [loss]])
        if self.world > 1:                                   # model.hip: one ncclAllReduce([grads..., loss]), 1/N in Adam
            import torch
            t = torch.from_numpy(flat)
            self.dist.all_reduce(t)
            flat = t.numpy() / self.world
        self.grads = flat[:-1]
        self.iterations += 1
        g = O.unflatten(self.spec, flat[:-1])
        new = O.adam_step({n: self.params[n] for n in g}, g, self.m, self.v, self.iterations, lr)
        self.params.update(new)
        self.params.update(state)
        w = O.loss_weight(y, **{k: cfg[k] for k in ('weight', 'weight_add', 'weight_mul') if k in cfg})
        return StepOut(float(flat[-1]), float(O.positive_rate(y)), float(w), float(y.min()), float(y.max()))

    def eval_step(self, x, y, cfg, return_prob=False):
        self._check(x)
        self.calls.append(('eval', len(x)))
        prob, logits = O.predict(self.spec, self.params, np.asarray(x, np.float64))
        per, _ = O.weighted_crossentropy(y, logits, **cfg)
        w = O.loss_weight(y, **{k: cfg[k] for k in ('weight', 'weight_add', 'weight_mul') if k in cfg})
        self.prob = prob.astype(np.float32)
        out = StepOut(float(per.mean() + O.l2_penalty(self.spec, self.params)), float(np.mean(y)), float(w), float(y.min()), float(y.max()))
        return (out, self.prob) if return_prob else out

    def pixel_confusion(self, y, thresholds):
        p, yy = self.prob.reshape(-1), (np.asarray(y, np.float32).reshape(-1) > 0.5)
        out = []
        for t in np.asarray(thresholds, np.float32).ravel():
            pp = p > t
            out.append((float((pp & yy).sum()), float((pp & ~yy).sum()), float((~pp & yy).sum()), float((~pp & ~yy).sum())))
        return out

    # ---- data parallel
    @staticmethod
    def comm_unique_id():
        return bytes(range(128))

    def comm_init(self, rank, world, unique_id):
        assert world == 1 or (unique_id is not None and len(unique_id) == 128)
        self.rank, self.world = rank, world
        if world > 1:
            self.dist.barrier()          # ncclCommInitRank is collective: it returns once every rank has joined

    def comm_broadcast_weights(self, root=0):
        import torch
        for buf_get, buf_set in ((self.get_params, self.set_params), (self.get_state, self.set_state)):
            t = torch.from_numpy(buf_get().astype(np.float64))
            if t.numel():
                self.dist.broadcast(t, root)
                buf_set(t.numpy())

    def comm_average_state(self):
        import torch
        if self.world > 1 and self.n_state:
            t = torch.from_numpy(O.flatten(self.spec, self.params, trainable=False).astype(np.float64))
            self.dist.all_reduce(t)
            self.set_state(t.numpy() / self.world)

    def comm_allreduce(self, values, op='sum'):
        import torch
        t = torch.from_numpy(np.array(values, np.float64).ravel())
        if self.world > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX if op == 'max' else self.dist.ReduceOp.SUM)
        return t.numpy()

    def profile_enable(self, *a, **k):
        pass

    def profile(self):
        return []
