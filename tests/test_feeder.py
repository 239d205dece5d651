"""The input side of the train loop: BatchFeeder (host logic, fake ring) and, -m gpu, the StagingRing / copy-stream path
against the synchronous host-buffer path it replaces in engine.train (annotator/data.py:110,143 prefetch; engine.py:126-135)."""

import threading
import time

import numpy as np
import pytest

from dnncancerannotator_amd import augment
from dnncancerannotator_amd.feeder import BatchFeeder

UNET = dict(n_filters_first=3, n_downsample=3, rate=2, kernel_size=3, conv_stride=1, bn=False, padding='same')


# ---------------------------------------------------------------------------------------------- host logic (no GPU)
class FakeRing:
    slots = 3

    def __init__(self, slot_bytes):
        self.slot_bytes, self.content, self.lock = slot_bytes, {}, threading.Lock()

    def fits(self, a, b=None):
        return a.nbytes + (0 if b is None else b.nbytes) <= self.slot_bytes

    def upload(self, slot, a, b=None, wait=True):
        with self.lock:
            self.content[slot] = (np.array(a), None if b is None else np.array(b))
        return ('a', slot), ('b', slot)


class FakeModel:
    max_batch = 4
    in_shape = (8, 8, 1)

    def __init__(self, slot_bytes=1 << 20):
        self.ring = FakeRing(slot_bytes)

    def staging(self, slots=None, slot_bytes=0):
        return self.ring


def _batches(n, b=2, s=8):
    rng = np.random.default_rng(0)
    return [(rng.random((b, s, s, 1), np.float32), (rng.random((b, s, s)) > 0.5).astype(np.float32)) for _ in range(n)]


def test_feeder_keeps_order_and_never_overwrites_an_unreleased_slot():
    dm, data = FakeModel(), _batches(10)
    f = BatchFeeder(dm, iter(data[1:]), lambda a, b=None: (a, b), first=data[0])
    held, seen = [], 0
    for kind, slot, px, py, n in f:
        assert kind == 'staged' and n == 2 and px == ('a', slot)
        # the slot's content is this batch, and slots we have not released still hold theirs
        assert np.array_equal(dm.ring.content[slot][0], data[seen][0])
        held.append((slot, seen))
        for s_, i_ in held:
            assert np.array_equal(dm.ring.content[s_][1], data[i_][1])
        seen += 1
        if len(held) == 2:                   # like the train loop: a slot goes back one step late
            f.release(held.pop(0)[0])
    assert seen == 10
    with pytest.raises(StopIteration):
        next(f)
    f.close()


def test_feeder_shards_falls_back_and_reports_errors():
    dm = FakeModel(slot_bytes=2 * (8 * 8 + 8 * 8) * 4)          # room for a batch of 2, not of 4
    small, big = _batches(1, b=2)[0], _batches(1, b=4)[0]
    shard = lambda a, b=None: (a[:len(a) // 2], None if b is None else b[:len(b) // 2])      # noqa: E731 (rank 0 of 2)

    def gen():
        yield big                           # its shard (2 images) fits
        yield (np.concatenate([big[0]] * 4), np.concatenate([big[1]] * 4))      # shard of 8 images: more than a slot and max_batch
        yield small
        raise RuntimeError('broken exam')

    f = BatchFeeder(dm, gen(), shard)
    a = next(f)
    assert a[0] == 'staged' and a[4] == 2 and np.array_equal(dm.ring.content[a[1]][0], big[0][:2])
    b = next(f)
    assert b[0] == 'host' and len(b[1][0]) == 16             # the loop uploads it itself (and shards it itself)
    c = next(f)
    assert c[0] == 'staged' and c[4] == 1
    with pytest.raises(RuntimeError, match='broken exam'):
        next(f)
    f.close()


def test_feeder_without_enough_slots_only_prefetches():
    dm, data = FakeModel(), _batches(4)
    f = BatchFeeder(dm, iter(data), lambda a, b=None: (a, b), slots=(0,))        # one slot: the loop would wait for itself
    got = list(f)
    assert [g[0] for g in got] == ['host'] * 4 and all(g[1] is d for g, d in zip(got, data)) and not dm.ring.content
    f.close()


def test_feeder_raw_batches_and_close_while_blocked():
    dm = FakeModel()
    raw = np.arange(2 * 12 * 12 * 2, dtype=np.uint8).reshape(2, 12, 12, 2)
    rb = augment.RawBatch(raw, [(0, 0, 0, 1.0)] * 2, (8, 8), 1, None)
    f = BatchFeeder(dm, iter([rb] * 50), lambda a, b=None: (a, b), first=rb)
    kind, slot, src, batch, n = next(f)
    assert kind == 'raw' and n == 2 and batch is rb and np.array_equal(dm.ring.content[slot][0], raw)
    time.sleep(0.05)                        # the producer now sits on an empty free-slot queue / a full ready queue
    f.close()
    assert not f.thread.is_alive()


def test_feeder_never_stages_a_batch_the_built_model_cannot_take():
    """A float batch of another image size, a uint8 batch cropped to another size / with another channel count / larger than
    max_batch go to the loop unstaged ('host'): its host path raises DeviceModel._check_x's ValueError or chunks the batch.  Staged,
    they would reach the kernels as bare pointers -- a 64 x 512 x 512 x 6 uint8 validation batch fills exactly the slot of a
    256 x 256 model and would be evaluated on misread memory."""
    dm = FakeModel()
    ident = lambda a, b=None: (a, b)                             # noqa: E731
    good, other_size = _batches(1, s=8)[0], _batches(1, s=16)[0]
    bad_y = (good[0], np.zeros((2, 8, 4), np.float32))
    raw = np.zeros((2, 12, 12, 2), np.uint8)
    prm = [(0, 0, 0, 1.0)] * 2
    elements = [good, other_size, bad_y,
                augment.RawBatch(raw, prm, (8, 8), 1, None),                                     # fits the built (8, 8, 1) input
                augment.RawBatch(raw, prm, (6, 6), 1, None),                                     # cropped to another size
                augment.RawBatch(np.zeros((2, 12, 12, 3), np.uint8), prm, (8, 8), 1, None),      # two feature channels
                augment.RawBatch(np.zeros((5, 12, 12, 2), np.uint8), [(0, 0, 0, 1.0)] * 5, (8, 8), 1, None)]      # 5 > max_batch
    f = BatchFeeder(dm, iter(elements), ident)
    kinds = []
    for item in f:
        kinds.append(item[0])
        if item[0] != 'host':
            f.release(item[1])
    f.close()
    assert kinds == ['staged', 'host', 'host', 'raw', 'host', 'host', 'host']


# ---------------------------------------------------------------------------------------------- -m gpu
def _model(gpu, seed=2, size=32, batch=2):
    m = gpu.DeviceModel('unet', 1, size, size, batch, **UNET)
    m.init_glorot(seed=seed)
    return m


def _host_batches(n, size=32, batch=2):
    from dnncancerannotator_amd.synthetic import synthetic_batch
    return [synthetic_batch(batch, size, size, 1, seed_x=50 + i, seed_y=70 + i) for i in range(n)]


@pytest.mark.gpu
def test_staging_ring_steps_equal_host_buffer_steps(gpu):
    data = _host_batches(7)
    ref, m = _model(gpu), _model(gpu)
    cfg = ref.loss_cfg(weight_mul=3.0)
    want = [ref.train_step(x, y, 1e-3, cfg) for x, y in data]
    ring = m.staging()
    got, prev = [], None
    for i, (x, y) in enumerate(data):
        slot = i % ring.slots
        px, py = ring.upload(slot, x, y)
        ring.train_step(slot, px, py, len(x), 1e-3, cfg)
        if prev is not None:
            got.append(ring.out(prev))       # one step late: the step just enqueued is still running
        prev = slot
    got.append(ring.out(prev))
    for a, b in zip(want, got):
        assert abs(a.loss - b.loss) <= 2e-5 * abs(a.loss)
        assert a.positive_rate == b.positive_rate and a.weight == b.weight and a.label_max == b.label_max
    pa, pb = ref.get_params(), m.get_params()
    for name, shape, tr, off in ref.param_infos():
        if tr:
            n = int(np.prod(shape))
            assert np.abs(pa[off:off + n] - pb[off:off + n]).max() <= 1e-4 * max(np.abs(pa[off:off + n]).max(), 1e-3), name


@pytest.mark.gpu
def test_staging_ring_errors_and_late_assertion(gpu):
    m = _model(gpu)
    ring = m.staging(slots=2)
    assert m.staging() is ring                                   # one ring per model
    cfg = m.loss_cfg(weight_mul=3.0)
    x = np.zeros((2, 32, 32, 1), np.float32)
    with pytest.raises(Exception):
        ring.upload(2, x, x[..., 0])                             # no such slot
    with pytest.raises(Exception):
        ring.upload(0, np.zeros((64, 32, 32, 1), np.float32), x[..., 0])      # larger than a slot
    with pytest.raises(Exception):
        ring.out(1)                                              # nothing ran on it
    px, py = ring.upload(0, x, np.full((2, 32, 32), 1.5, np.float32))
    ring.train_step(0, px, py, 2, 1e-3, cfg)                     # enqueued: the assertion surfaces with the scalars
    with pytest.raises(Exception) as e:                          # assert_on_max (utils/losses.py:91)
        ring.out(0)
    assert 'label outside' in str(e.value)
    px, py = ring.upload(1, x, np.zeros((2, 32, 32), np.float32))
    ring.train_step(1, px, py, 2, 1e-3, cfg)
    out = ring.out(1)
    assert out.positive_rate == 0.0 and out.weight == 3.0
    # staged evaluation: call order is checked, the histogram cannot be borrowed in between
    with pytest.raises(Exception):
        ring.eval_step(1, px, py, 2, cfg)                        # no eval_begin
    ring.eval_begin([0.5, 0.25])
    with pytest.raises(Exception):
        m.pixel_confusion(np.zeros((2, 32, 32), np.float32), [0.5])
    ring.eval_step(1, px, py, 2, cfg)
    counts = ring.eval_end()
    assert len(counts) == 2 and sum(counts[0]) == 2 * 32 * 32 and counts[0][0] == 0 and counts[1][1] >= counts[0][1]
    assert ring.out(1).loss > 0
    with pytest.raises(Exception):
        ring.eval_end()                                          # ended already


@pytest.mark.gpu
def test_engine_train_with_feeder_equals_the_synchronous_loop(gpu, tmp_path, monkeypatch):
    from dnncancerannotator_amd import data, engine
    config = {'model': 'UNetAnnotator', 'model_options': UNET,
              'deploy_options': {'optimizer': 'adam', 'loss': {'class_name': 'WeightedCrossentropy', 'config': {'weight_mul': 3.0}},
                                 'LearningRateScheduler': 'lambda epoch, current_lr: 0.001 * 0.96 ** (epoch // 3)',
                                 'enable_multigpu': False}}

    def run(tag):
        ds = data.SyntheticDataset(4, 64, 64, 1, n_batches=3, seed=3)
        val = data.SyntheticDataset(4, 64, 64, 1, n_batches=1, seed=9, repeat=False)
        m = engine.TFKerasModel(config)
        res = m.train(ds, save_path=str(tmp_path / tag), max_steps=11, save_freq=4, val_data=val)
        return m, res

    m1, r1 = run('feeder')
    assert getattr(m1.device_model, '_ring', None) is not None            # the staged path really ran
    monkeypatch.setenv('DNNCA_NO_FEEDER', '1')
    m2, r2 = run('sync')
    assert getattr(m2.device_model, '_ring', None) is None
    assert r1.epoch == r2.epoch == list(range(11))
    assert np.allclose(r1.history['loss'], r2.history['loss'], rtol=2e-4)
    assert r1.history['lr'] == r2.history['lr']
    assert len(r1.history['val_loss']) == len(r2.history['val_loss']) == 2  # steps 4 and 8
    assert np.allclose(r1.history['val_loss'], r2.history['val_loss'], rtol=2e-4)
    assert list(m1.get_ckpts(str(tmp_path / 'feeder' / 'checkpoints'))) == [4, 8]
    assert m1.current_step == m2.current_step == 11


@pytest.mark.gpu
def test_engine_train_refuses_a_validation_set_of_another_size(gpu, tmp_path):
    """`annotator train --validate` with data_options whose train and eval sizes differ (data_options.yaml: 256 vs 512): the
    validation inside train() must raise the shape error, staged or not -- not evaluate misread memory and feed early stopping."""
    from dnncancerannotator_amd import data, engine
    config = {'model': 'UNetAnnotator', 'model_options': UNET,
              'deploy_options': {'optimizer': 'adam', 'loss': {'class_name': 'WeightedCrossentropy', 'config': {'weight_mul': 3.0}},
                                 'enable_multigpu': False}}
    ds = data.SyntheticDataset(4, 32, 32, 1, n_batches=3, seed=3)
    val = data.SyntheticDataset(1, 64, 64, 1, n_batches=1, seed=9, repeat=False)      # same bytes as a 4 x 32 x 32 batch: it "fits" a slot
    m = engine.TFKerasModel(config)
    with pytest.raises(ValueError, match='does not match the built input'):
        m.train(ds, save_path=str(tmp_path / 'run'), max_steps=5, save_freq=2, val_data=val)
    # device-resident views carry their shapes: the *_dev entry point refuses them as well
    from dnncancerannotator_amd import device as dev
    xb, yb = dev.DeviceBuffer(np.zeros((1, 64, 64, 1), np.float32)), dev.DeviceBuffer(np.zeros((1, 64, 64), np.float32))
    with pytest.raises(ValueError, match='does not match the built input'):
        m.device_model.train_step_dev(xb, yb, 1, 1e-3, m.device_model.loss_cfg(weight_mul=3.0))


@pytest.mark.gpu
def test_engine_feeder_with_device_side_augmentation(gpu, monkeypatch):
    """uint8 source batches travel through the staging slots too; same draws -> same losses as the loop that uploads them itself"""
    from dnncancerannotator_amd import engine
    config = {'model': 'UNetAnnotator', 'model_options': UNET,
              'deploy_options': {'optimizer': 'adam', 'loss': {'class_name': 'WeightedCrossentropy', 'config': {'weight_mul': 3.0}},
                                 'enable_multigpu': False}}
    rng = np.random.default_rng(5)
    raws = []
    for _ in range(5):
        raw = rng.integers(0, 256, size=(2, 44, 44, 2), dtype=np.uint8)
        raw[..., 1] = (raw[..., 1] > 200).astype(np.uint8) * 255           # label channel: 0 / 255
        params = [(int(rng.integers(-6, 7)), int(rng.integers(-6, 7)), int(rng.integers(0, 2)), float(rng.uniform(0.8, 1.2)))
                  for _ in range(2)]
        raws.append(augment.RawBatch(raw, params, (32, 32), 1, None))

    class DS:
        element_spec = None

        def __iter__(self):
            return iter(raws)

    def run():
        m = engine.TFKerasModel(config)
        m._build([(np.zeros((2, 32, 32, 1), np.float32), np.zeros((2, 32, 32), np.float32))])
        return m.train(DS(), max_steps=5).history['loss']

    a = run()
    monkeypatch.setenv('DNNCA_NO_FEEDER', '1')
    b = run()
    assert len(a) == len(b) == 5 and np.allclose(a, b, rtol=2e-4)


@pytest.mark.gpu
def test_engine_eval_over_the_staging_ring_equals_the_per_batch_path(gpu, tmp_path, monkeypatch):
    """`annotator evaluate`: test steps on staged batches, ONE device-resident confusion histogram for all metrics (302
    thresholds of configs/additionals/metrics.yaml:2-23), losses read one step late -- same rows as the per-batch path"""
    from dnncancerannotator_amd import data, engine
    config = {'model': 'UNetAnnotator', 'model_options': UNET,
              'deploy_options': {'optimizer': 'adam', 'loss': {'class_name': 'WeightedCrossentropy', 'config': {'weight_mul': 3.0}},
                                 'enable_multigpu': False,
                                 'metrics': [{'Precision': {'thresholds': 0.8, 'name': 'pixel/precision'}},
                                             {'Recall': {'thresholds': 0.8, 'name': 'pixel/recall'}},
                                             {'AUC': {'curve': 'PR', 'name': 'pixel/AUPRC', 'num_thresholds': 150}},
                                             {'AUC': {'curve': 'ROC', 'name': 'pixel/AUROC', 'num_thresholds': 150}},
                                             {'FBetaScore': {'thresholds': 0.8, 'beta': 1.0, 'name': 'pixel/F1-score'}},
                                             {'FBetaScore': {'thresholds': 0.5, 'beta': 2.0, 'name': 'pixel/F2-score'}}]}}
    save = str(tmp_path / 'run')
    m = engine.TFKerasModel(config)
    m.train(data.SyntheticDataset(4, 64, 64, 1, n_batches=2, seed=3), save_path=save, max_steps=12, save_freq=6)
    # 5 batches of 4 and a last one of 2 (a remainder batch is a smaller test step)
    xs, ys = zip(*[_host_batches(1, size=64, batch=4)[0] for _ in range(5)])
    x, y = np.concatenate(xs)[:18], np.concatenate(ys)[:18]
    x[4:] = np.random.default_rng(1).random(x[4:].shape, np.float32)
    ev = data.ArrayDataset(x, y, 4)
    rows_a = engine.TFKerasModel(config).eval(ev, save_path=save, tag='staged')
    monkeypatch.setenv('DNNCA_NO_FEEDER', '1')
    rows_b = engine.TFKerasModel(config).eval(ev, save_path=save, tag='per_batch')
    assert list(rows_a) == list(rows_b) == [6, 12]
    for step in rows_a:
        a, b = rows_a[step], rows_b[step]
        assert set(a) == set(b) and len(a) == 7
        assert abs(a['loss'] - b['loss']) <= 1e-6 * abs(b['loss'])
        for k in a:
            if k != 'loss':
                assert a[k] == b[k], (step, k)        # integer counts: the metrics are identical


@pytest.mark.gpu
def test_evaluate_tfrecords_with_device_side_conversion(gpu, tmp_path, monkeypatch):
    """`annotator evaluate` on exam files: uint8 slices through the staging ring, / 255 + feature-label split on the device --
    the same rows as float batches converted on the host, staged or not"""
    from dnncancerannotator_amd import data, engine, tfrecord as T
    from dnncancerannotator_amd.runs.train import make_dataset
    rng = np.random.default_rng(4)
    paths = []
    for e, n in enumerate((5, 3, 6)):
        s = rng.integers(0, 256, size=(n, 72, 80, 3), dtype=np.uint8)
        s[..., 2] = (s[..., 2] > 215) * 255
        p = str(tmp_path / ('exam%d.tfrecords' % e))
        T.write_records(p, [T.make_example(s, e, e, '/e/%d' % e, 'cancer', ['TRA', 'ADC', 'label'])])
        paths.append(p)
    opts = dict(batch_size=4, output_size=(64, 64), slice_types=['ADC', 'label'])
    config = {'model': 'UNetAnnotator', 'model_options': UNET,
              'deploy_options': {'optimizer': 'adam', 'loss': {'class_name': 'WeightedCrossentropy', 'config': {'weight_mul': 3.0}},
                                 'enable_multigpu': False,
                                 'metrics': [{'Precision': {'thresholds': 0.5, 'name': 'pixel/precision'}},
                                             {'AUC': {'curve': 'ROC', 'name': 'pixel/AUROC', 'num_thresholds': 50}}]}}
    save = str(tmp_path / 'run')
    engine.TFKerasModel(config).train(data.SyntheticDataset(4, 64, 64, 1, n_batches=2, seed=3), save_path=save, max_steps=6, save_freq=6)
    ds = make_dataset(paths, opts, training=False)
    assert ds.device_convert and isinstance(next(iter(ds)), augment.RawBatch)
    rows = {'device': engine.TFKerasModel(config).eval(ds, save_path=save, tag='device')}
    host = T.TFRecordDataset(paths, opts['slice_types'], 4, output_size=(64, 64))
    rows['host'] = engine.TFKerasModel(config).eval(host, save_path=save, tag='host')
    monkeypatch.setenv('DNNCA_NO_FEEDER', '1')
    rows['device_sync'] = engine.TFKerasModel(config).eval(ds, save_path=save, tag='device_sync')
    for k in ('host', 'device_sync'):
        a, b = rows['device'][6], rows[k][6]
        assert abs(a['loss'] - b['loss']) <= 1e-6 * abs(b['loss']), k
        assert a['pixel/precision'] == b['pixel/precision'] and a['pixel/AUROC'] == b['pixel/AUROC'], k
