"""Engine facade: the MI355X stand-in for annotator/engine.py:36-288 (class TFKerasModel).

Same constructor, same methods (train / eval / predict / save / load / get_ckpts / get_config), same save-directory
layout (checkpoints/ckpt-{step}.index, options.yaml, results.csv); underneath, a DeviceModel (libdnnca, hand-written HIP)
instead of a compiled tf.keras.Model.  One Keras "epoch" of the reference is one optimizer step (engine.py:126-135), so
the loop here is a plain step loop.

Data parallel (deploy_options.enable_multigpu, engine.py:260-263): one process per GPU.  Each rank takes its contiguous
shard of every global batch, libdnnca sums the gradients with one RCCL all-reduce per step, rank 0 writes files.
Ranks come from RANK / LOCAL_RANK / WORLD_SIZE (set by `python -m dnncancerannotator_amd.launch` or torchrun)."""

import copy
import json
import logging
import os
import re
import time
from collections import OrderedDict

import numpy as np

from . import augment, device, distributed, losses as custom_losses, metrics as custom_metrics, models
from .feeder import BatchFeeder


class History:
    """The part of keras.callbacks.History that utils/dump.py:68-73 reads."""

    def __init__(self, model):
        self.epoch = []
        self.history = {}
        self.params = {}
        self.model = model

    def log(self, step, logs):
        self.epoch.append(step)
        for k, v in logs.items():
            self.history.setdefault(k, []).append(v)


def _element_shape(dataset):
    """Input shape [B, H, W, C] of a dataset: `element_spec[0].shape` (engine.py:93) or, failing that, its first batch."""
    spec = getattr(dataset, 'element_spec', None)
    if spec is not None:
        return tuple(spec[0].shape)
    for element in dataset:          # (x, y) for train / eval, (x,) for predict
        return tuple(np.shape(element[0]))
    raise ValueError('empty dataset')


class TFKerasModel:
    """Encapsulates the DNN model and the library behind it (name kept from the reference for drop-in use)."""

    def __init__(self, model_config):
        self.model_config = copy.deepcopy(model_config)
        self.ctx = distributed.context()
        self.model = self.from_config(model_config)
        self.current_step = 0
        self.ckpt_pattern = 'ckpt-{epoch}'
        self.device_model = None

    # ---- construction (engine.py:254-288) --------------------------------------------------------------------
    def from_config(self, model_config):
        assert 'model' in model_config
        assert 'model_options' in model_config
        assert 'deploy_options' in model_config
        deploy = copy.deepcopy(model_config['deploy_options'])
        self.enable_multigpu = deploy.pop('enable_multigpu', True)
        if not self.enable_multigpu and self.ctx.world > 1:
            raise ValueError('enable_multigpu is false but WORLD_SIZE is %d' % self.ctx.world)
        self.learning_rate_scheduler = deploy.pop('LearningRateScheduler', None)
        model = getattr(models, model_config['model'])(**model_config['model_options'])
        self.loss = custom_losses.get(deploy['loss']) if 'loss' in deploy else custom_losses.WeightedCrossentropy()
        self.metrics = [m for m in map(custom_metrics.solve_metric, deploy.get('metrics', [])) if m is not None]
        if deploy.get('optimizer') != 'adam':
            raise NotImplementedError('only the reference\'s optimizer: adam is supported (engine.py:276-284)')
        self.learning_rate = 0.001          # engine.py:278
        self.adam = dict(beta1=0.9, beta2=0.999, epsilon=1e-7)
        return model

    def _build(self, dataset, max_batch=None, also=()):
        """model.build(element_spec.shape) (engine.py:93).  `also`: further datasets that will be fed to this model (the
        validation set of train(): configs/additionals/data_options.yaml has train batch 8, eval batch 64) -- the device
        model is sized for the largest per-rank batch so that a validation batch is ONE eval step and its positive-rate
        weight is computed over the whole per-replica batch, as utils/losses.py:24-27,87-95 does."""
        shape = _element_shape(dataset)
        global_batch = max_batch or shape[0]
        if global_batch is None:
            raise ValueError('dataset batch size unknown')
        if global_batch % self.ctx.world:
            raise ValueError('global batch %d is not divisible by %d ranks' % (global_batch, self.ctx.world))
        per_rank = global_batch // self.ctx.world
        for other in also:
            if other is not None:
                b = _element_shape(other)[0]
                if b:
                    per_rank = max(per_rank, -(-b // self.ctx.world))
        if self.device_model is not None:
            self._ensure_capacity(per_rank)
            return
        device.init_device(self.ctx.local_rank)
        if self.enable_multigpu and self.ctx.world == 1 and device.device_count() > 1:
            logging.warning('enable_multigpu: %d GPUs visible but one process; launch with '
                            '`python -m dnncancerannotator_amd.launch --nproc N ...` for data parallel', device.device_count())
        self._input_shape = list(shape[1:])
        self.device_model = self.model.build([per_rank] + self._input_shape, seed=0)
        self.device_model.set_adam(**self.adam)
        self._join_ranks()

    def _join_ranks(self):
        if self.ctx.world > 1:
            self.device_model.comm_init(self.ctx.rank, self.ctx.world, distributed.exchange_unique_id(self.ctx, device.DeviceModel))
            distributed.cleanup(self.ctx)     # ncclCommInitRank returned: every rank has read the id file
            # identical initial weights on every replica (MirroredStrategy mirrors rank 0's variables)
            self.device_model.comm_broadcast_weights(0)

    def _ensure_capacity(self, per_rank_batch):
        """Grow the device model to `per_rank_batch` slices per step (weights, BN statistics and Adam slots carried over).
        Returns False when HBM cannot hold it (the caller then evaluates in chunks)."""
        dm = self.device_model
        if per_rank_batch <= dm.max_batch:
            return True
        if self.ctx.world > 1:
            return False      # the communicator belongs to the existing handle; _build sizes DP models up front
        params, state, (m, v, it) = dm.get_params(), dm.get_state(), dm.get_opt_state()
        try:
            bigger = self.model.build([per_rank_batch] + self._input_shape, seed=0)
        except Exception as e:           # out of HBM: keep the model we have
            logging.warning('cannot size the model for a batch of %d (%s): evaluating in chunks', per_rank_batch, e)
            self.model.device_model = dm
            return False
        dm.close()
        bigger.set_params(params)
        if bigger.n_state:
            bigger.set_state(state)
        bigger.set_opt_state(m, v, it)
        bigger.set_adam(**self.adam)
        self.device_model = bigger
        return True

    def _shard(self, x, y=None):
        """Contiguous shard of a global batch for this rank (Keras splits the batch across replicas [TF-2.6]); a batch
        that does not divide evenly (the last batch of an evaluation set) gives its remainder to the first ranks."""
        if self.ctx.world == 1:
            return x, y
        lo, hi = distributed.shard_bounds(len(x), self.ctx.rank, self.ctx.world, even=False)
        return x[lo:hi], (None if y is None else y[lo:hi])

    def _shard_fn(self, dataset):
        """the split for the elements of `dataset`: none for one rank and for datasets that hand every rank its part already
        (tfrecord.TFRecordDataset(shard=...): `pre_sharded`)"""
        if self.ctx.world == 1 or getattr(dataset, 'pre_sharded', False):
            return lambda x, y=None: (x, y)
        return self._shard

    # ---- checkpoints (engine.py:55-78, 103-106, 224-231) -----------------------------------------------------
    def get_ckpts(self, base_path):
        regex_pattern = fr'^{self.ckpt_pattern}\.index$'.format(epoch=r'(\d+)')
        files = [f for f in os.listdir(base_path) if re.match(regex_pattern, f)]
        steps = [int(re.sub(regex_pattern, r'\1', f)) for f in files]
        paths = [os.path.join(base_path, f[:-len('.index')]) for f in files]
        return OrderedDict(sorted(zip(steps, paths), key=lambda x: x[0]))

    def _auto_resume(self, base_path):
        if not os.path.exists(base_path):
            return
        ckpts = self.get_ckpts(base_path)
        if not ckpts:
            return
        latest_step = max(ckpts.keys())
        self.load(ckpts[latest_step])
        self.current_step = latest_step
        logging.warning(f'Resumed from {latest_step}')

    def save(self, path, fileformat=None):
        """Weights + Adam slots + BN moving statistics + step counters: `<path>.index` (JSON) and `<path>.data-00000-of-00001`."""
        dm = self.device_model
        if self.ctx.world > 1:
            dm.comm_average_state()      # BN moving statistics: mean over replicas (ON_READ / MEAN aggregation [TF-2.6])
        if self.ctx.rank != 0:
            return self
        m, v, iterations = dm.get_opt_state()
        os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
        with open(path + '.data-00000-of-00001', 'wb') as f:
            np.savez(f, params=dm.get_params(), state=dm.get_state(), adam_m=m, adam_v=v)
        index = dict(format='dnnca-ckpt-1', step=int(self.current_step), iterations=int(iterations),
                     learning_rate=float(self.learning_rate),
                     variables=[dict(name=n, shape=list(s), trainable=t, offset=o) for n, s, t, o in dm.param_infos()])
        with open(path + '.index', 'w') as f:      # written last: its presence marks a complete checkpoint
            json.dump(index, f)
        return self

    def load(self, path):
        dm = self.device_model
        with open(path + '.index') as f:
            index = json.load(f)
        have = [(v['name'], tuple(v['shape']), v['trainable'], v['offset']) for v in index['variables']]
        if have != [(n, tuple(s), t, o) for n, s, t, o in dm.param_infos()]:
            raise ValueError(f'checkpoint {path} does not match the model (assert_existing_objects_matched)')
        with np.load(path + '.data-00000-of-00001') as z:
            dm.set_params(z['params'])
            if dm.n_state:
                dm.set_state(z['state'])
            dm.set_opt_state(z['adam_m'], z['adam_v'], index['iterations'])
        return self

    # ---- training (engine.py:80-137) -------------------------------------------------------------------------
    def train(self, dataset, val_data=None, save_path=None, save_freq=100, max_steps=None, early_stop_steps=None,
              visualization=None, auto_resume=True, profile=False):
        self._build(dataset, also=(val_data,))
        dm = self.device_model
        if auto_resume and save_path is not None:
            self._auto_resume(os.path.join(save_path, 'checkpoints'))
        schedule = eval(self.learning_rate_scheduler) if self.learning_rate_scheduler is not None else None  # engine.py:98-100
        log_file = None
        if save_path is not None and self.ctx.rank == 0:
            os.makedirs(os.path.join(save_path, 'checkpoints'), exist_ok=True)
            os.makedirs(os.path.join(save_path, 'tfevents'), exist_ok=True)
            log_file = open(os.path.join(save_path, 'tfevents', 'train_log.csv'), 'a')
        if visualization:
            logging.warning('visualization callbacks are outside the accelerated path: ignored')
        cfg = dm.loss_cfg(**self.loss.device_cfg())
        results = History(self.model)
        results.params = dict(epochs=max_steps, steps=1, verbose=0)
        best_val, wait = np.inf, 0
        shard = self._shard_fn(dataset)
        source = iter(dataset)
        step = self.current_step
        if profile:
            dm.profile_enable(1)
        # Input side (annotator/data.py:110,143 `.prefetch(AUTOTUNE)` + Keras' asynchronous feeding): a BatchFeeder thread draws
        # the elements, shards them and uploads them into the model's staging slots on a copy stream while the GPU is still on
        # the previous step; the loop enqueues the step and reads the scalars of the step BEFORE it, so one step is always queued
        # behind the running one.  Checkpoint / validation steps, the last step and DNNCA_NO_FEEDER=1 read them at once.
        feeder = None
        if hasattr(dm, 'staging') and not os.environ.get('DNNCA_NO_FEEDER') and (max_steps is None or step < max_steps):
            try:
                first = next(source)
            except StopIteration:
                first = None
            if first is not None:
                feeder = source = BatchFeeder(dm, source, shard, first=first)
            else:
                source = iter(())
        pending = None          # (slot, step, lr) of an enqueued step whose scalars have not been read
        t0 = time.time()

        def log_step(at, out, lr, extra=None):
            logs = dict(loss=float(out.loss), lr=lr)
            if extra:
                logs.update(extra)
            results.log(at - 1, logs)
            if log_file is not None:
                log_file.write('%d,%s\n' % (at, ','.join('%s=%.8g' % kv for kv in logs.items())))
            if self.ctx.rank == 0 and (at % 100 == 0 or at == max_steps):
                logging.info('step %d loss %.6f lr %.3g (%.1f steps/s)', at, out.loss, lr,
                             (at - results.epoch[0]) / max(time.time() - t0, 1e-9))

        def read_pending():
            nonlocal pending
            if pending is not None:
                slot_, at, lr = pending
                pending = None
                out_ = feeder.ring.out(slot_)       # waits for that step only; raises its label / weight assertions
                feeder.release(slot_)
                log_step(at, out_, lr)

        try:
            while max_steps is None or step < max_steps:
                try:
                    item = next(source)
                except StopIteration:
                    logging.warning('dataset exhausted at step %d', step)
                    break
                if schedule is not None:
                    self.learning_rate = float(schedule(step, self.learning_rate))
                if feeder is None:
                    item = ('host', item)
                slot, out = None, None
                if item[0] == 'staged':              # float (x, y), already on its way into a staging slot
                    _, slot, px, py, n = item
                    feeder.ring.train_step(slot, px, py, n, self.learning_rate, cfg)
                elif item[0] == 'raw':               # uint8 source batch in a staging slot: augmentation kernels, then the step
                    _, slot, src, batch, n = item
                    feeder.ring.wait(slot)
                    params, _ = shard(batch.params)
                    xb, yb = dm.augment_u8(shard(batch.raw)[0], params, batch.output_size, batch.label_index, src_ptr=src)
                    if batch.warp is not None:
                        xb, yb = dm.warp(xb, yb, shard(batch.warp[0])[0], shard(batch.warp[1])[0])
                    dm.check_dev(xb, yb, n)
                    feeder.ring.train_step(slot, xb.ptr, yb.ptr, n, self.learning_rate, cfg)
                else:
                    batch = item[1]
                    if isinstance(batch, augment.RawBatch):
                        # uint8 slices + their random draws: crop / flip / contrast / 255 / feature-label split on the device
                        raw, _ = shard(batch.raw)
                        params, _ = shard(batch.params)
                        xb, yb = dm.augment_u8(raw, params, batch.output_size, batch.label_index)
                        if batch.warp is not None:
                            xb, yb = dm.warp(xb, yb, shard(batch.warp[0])[0], shard(batch.warp[1])[0])
                        out = dm.train_step_dev(xb, yb, len(raw), self.learning_rate, cfg, want_out=True)
                    else:
                        x, y = shard(np.asarray(batch[0]), np.asarray(batch[1]))
                        out = dm.train_step(x, y, self.learning_rate, cfg)
                step += 1
                self.current_step = step
                read_pending()                       # the step before this one (its slot goes back to the feeder)
                boundary = step % save_freq == 0 and (save_path is not None or val_data is not None)
                if slot is not None:
                    if not boundary and (max_steps is None or step < max_steps):
                        pending = (slot, step, self.learning_rate)
                        continue
                    out = feeder.ring.out(slot)
                    feeder.release(slot)
                extra = {}
                if save_path is not None and step % save_freq == 0:
                    self.save(os.path.join(save_path, 'checkpoints', self.ckpt_pattern.format(epoch=step)))
                stop = False
                if val_data is not None and step % save_freq == 0:
                    val = self._evaluate(val_data, staged=feeder is not None)      # on the ring's evaluation slots
                    extra.update({'val_' + k: v for k, v in val.items()})
                    if early_stop_steps is not None:
                        if val['loss'] < best_val:
                            best_val, wait = val['loss'], 0
                        else:
                            wait += 1
                            stop = wait >= early_stop_steps
                log_step(step, out, self.learning_rate, extra)
                if stop:
                    logging.warning('early stopping at step %d', step)
                    break
            read_pending()
        finally:
            if feeder is not None:
                feeder.close()
        if log_file is not None:
            log_file.close()
        if profile and save_path is not None and self.ctx.rank == 0:
            with open(os.path.join(save_path, 'tfevents', 'kernel_profile.txt'), 'w') as f:
                for row in sorted(dm.profile(), key=lambda r: -r[2]):
                    f.write('%-28s launches %8d  total %10.3f ms\n' % row[:3])
        return results

    # ---- evaluation (engine.py:139-210) ----------------------------------------------------------------------
    def _evaluate(self, dataset, staged=False):
        """keras Model.evaluate(return_dict=True): mean loss over batches + pixel metrics, training=False.  One batch of
        the dataset is one test step per replica: the positive-rate class weight (utils/losses.py:24-27) is taken over the
        whole per-replica batch, never over a chunk of it."""
        cfg_kw = self.loss.device_cfg()
        for m in self.metrics:
            m.reset_state()
        total, count = 0.0, 0
        shard = self._shard_fn(dataset)
        if staged and self._staged_eval_possible():
            total, count, dataset = self._evaluate_staged(dataset, cfg_kw, shard)      # what is left: batches the ring could not take
        for el in dataset:
            x, y = augment.raw_to_float(el) if isinstance(el, augment.RawBatch) else el
            x, y = shard(np.asarray(x), np.asarray(y))
            if len(x):
                self._ensure_capacity(len(x))
            dm = self.device_model
            if 0 < len(x) <= dm.max_batch:
                out = dm.eval_step(x, y, dm.loss_cfg(**cfg_kw))
                total += float(out.loss) * len(x)
                count += len(x)
                for m in self.metrics:
                    m.update_state(dm, y)
            elif len(x):
                # HBM cannot hold the batch in one step: chunks, each with the weight of the WHOLE batch passed explicitly
                kw = dict(cfg_kw)
                if kw.get('weight') is None:
                    if kw.get('label_smoothing'):
                        logging.warning('label smoothing + chunked validation: the class weight is taken per chunk')
                    else:
                        rate = float(np.asarray(y, np.float64).mean())
                        kw['weight'] = 1.0 / rate if rate > 0 else 1.0          # utils/losses.py:25-27
                cfg = dm.loss_cfg(**kw)
                for i in range(0, len(x), dm.max_batch):
                    xb, yb = x[i:i + dm.max_batch], y[i:i + dm.max_batch]
                    out = dm.eval_step(xb, yb, cfg)
                    total += float(out.loss) * len(xb)
                    count += len(xb)
                    for m in self.metrics:
                        m.update_state(dm, yb)
        if self.ctx.world > 1:
            dm = self.device_model
            total, count = (float(v) for v in dm.comm_allreduce([total, count]))
            for m in self.metrics:                      # counts travel as doubles: exact far beyond 2^24 pixels
                m.merge(lambda c: np.asarray(dm.comm_allreduce(c.ravel()), np.float64).reshape(c.shape))
        results = OrderedDict(loss=total / max(count, 1))
        for m in self.metrics:
            r = m.result()
            results[m.name] = float(r) if np.ndim(r) == 0 else [float(v) for v in r]
        return results

    def _staged_eval_possible(self):
        n_thr = sum(len(m.thresholds) for m in self.metrics)
        return (hasattr(self.device_model, 'staging') and not os.environ.get('DNNCA_NO_FEEDER') and n_thr <= 1024 and
                all(hasattr(m, 'thresholds') and hasattr(m, 'counts') for m in self.metrics))

    def _evaluate_staged(self, dataset, cfg_kw, shard):
        """The test steps of _evaluate over the staging ring (feeder.py): batches travel to HBM on the copy stream while the
        previous one is evaluated, every metric's thresholds share ONE confusion histogram that stays on the device and is read
        once at the end (exact integer counts), and a batch's loss is read one step late.  Returns (loss sum, sample count,
        the batches the ring could not take -- larger than max_batch -- for the chunked path)."""
        dm = self.device_model
        source = iter(dataset)
        try:
            first = next(source)
        except StopIteration:
            return 0.0, 0, []
        self._ensure_capacity(len(shard(np.asarray(first.raw if isinstance(first, augment.RawBatch) else first[0]))[0]))
        dm = self.device_model
        feeder = BatchFeeder(dm, source, shard, slots=BatchFeeder.EVAL_SLOTS, first=first)
        ring = feeder.ring
        thr = np.concatenate([m.thresholds for m in self.metrics]) if self.metrics else np.zeros(0, np.float32)
        cfg = dm.loss_cfg(**cfg_kw)
        total, count, left, pending = 0.0, 0, [], None
        ring.eval_begin(thr)
        try:
            for item in feeder:
                if item[0] == 'host':
                    left.append(item[1])
                    continue
                if item[0] == 'raw':         # uint8 slices in the slot: centre crop, / 255 and the feature-label split on the device
                    _, slot, src, batch, n = item
                    ring.wait(slot)
                    xv, yv = dm.augment_u8(shard(batch.raw)[0], augment.plain_params(n), batch.output_size, batch.label_index,
                                           contrast_channels=(), src_ptr=src)
                    dm.check_dev(xv, yv, n)
                    px, py = xv.ptr, yv.ptr
                else:
                    _, slot, px, py, n = item
                ring.eval_step(slot, px, py, n, cfg)
                if pending is not None:
                    total += float(ring.out(pending[0]).loss) * pending[1]
                    count += pending[1]
                    feeder.release(pending[0])
                pending = (slot, n)
            if pending is not None:
                total += float(ring.out(pending[0]).loss) * pending[1]
                count += pending[1]
        finally:
            feeder.close()
            counts = np.asarray(ring.eval_end(), np.float64).reshape(-1, 4)
        lo = 0
        for m in self.metrics:
            m.counts += counts[lo:lo + len(m.thresholds)]
            lo += len(m.thresholds)
        return total, count, left

    def eval(self, dataset, save_path, viz_ds=None, tag='val', avoid_overwrite=False, export_path=None, export_images=False,
             visualize_sensitivity=False, export_csv=False, min_interval=1, step_range=None, overlay=False,
             export_casewise_metrics=False):
        self._build(dataset)
        ckpt_path = os.path.join(save_path, 'checkpoints')
        if not export_path:
            export_path = os.path.join(save_path, 'tfevents')
        if os.path.exists(os.path.join(export_path, tag)):
            if avoid_overwrite:
                while os.path.exists(os.path.join(export_path, tag)):
                    tag += '_'
            else:
                raise ValueError(f'tag: {tag} already exists.')
        if step_range is None:
            step_range = 0, float('inf')
        else:
            assert len(step_range) == 2
            assert 0 <= step_range[0] <= step_range[1]
        if viz_ds is not None or export_images or visualize_sensitivity or overlay:
            logging.warning('visualisation / image export are outside the accelerated path: ignored')
        rows = OrderedDict()
        previous_step = None
        for ckpt_step, ckpt_path_ in self.get_ckpts(ckpt_path).items():
            if not step_range[0] <= ckpt_step <= step_range[1]:
                continue
            if previous_step is not None and (ckpt_step - previous_step) < min_interval:
                logging.warning(f'Ignored {ckpt_path_} due to min_interval:{min_interval}.')
                continue
            previous_step = ckpt_step
            self.load(ckpt_path_)
            rows[ckpt_step] = self._evaluate(dataset, staged=True)
        if export_csv and self.ctx.rank == 0:
            os.makedirs(os.path.join(export_path, tag), exist_ok=True)
            with open(os.path.join(export_path, tag, 'results.csv'), 'w') as f:
                cols = list(next(iter(rows.values())).keys()) if rows else ['loss']
                f.write('step,' + ','.join(cols) + '\n')
                for step, r in rows.items():
                    f.write(str(step) + ',' + ','.join(str(r[c]) for c in cols) + '\n')
        return rows

    def predict(self, dataset):
        """Probabilities [N, H, W, 1] for every element of `dataset` (elements are x or (x, ...))."""
        self._build(dataset)
        dm = self.device_model
        outs = []
        for el in dataset:
            x = np.asarray(el[0] if isinstance(el, (tuple, list)) else el)
            for i in range(0, len(x), dm.max_batch):
                outs.append(dm.forward(x[i:i + dm.max_batch], training=False))
        return np.concatenate(outs) if outs else np.zeros((0,))

    def get_config(self):
        return self.model_config
