"""Worker of tests/test_dp_gloo.py::test_engine_world2: drives engine.TFKerasModel.train / save / _evaluate / eval on
WORLD_SIZE CPU ranks (gloo) with tests/fake_device.FakeDeviceModel (the numpy oracle) in place of the HIP DeviceModel, so
that the engine's own world > 1 branches run: rank discovery, id rendezvous + cleanup, broadcast of rank 0's weights, batch
shards (with a remainder), per-replica validation weights, double-precision metric merging (150-threshold AUC = 600
counters), BatchNorm-state averaging before a checkpoint, rank-0-only files."""

import json
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]

from dnncancerannotator_amd import data, device, distributed, engine, models   # noqa: E402
from fake_device import FakeDeviceModel                                          # noqa: E402
from oracle import unet_oracle as O                                              # noqa: E402

OPTS = dict(n_filters_first=3, n_downsample=2, rate=2, kernel_size=3, conv_stride=1, bn=True, padding='same')
CONFIG = {
    'model': 'UNetAnnotator', 'model_options': OPTS,
    'deploy_options': {'optimizer': 'adam', 'LearningRateScheduler': 'lambda epoch, current_lr: 0.001 * 0.96 ** (epoch // 1000)',
                       'loss': {'class_name': 'WeightedCrossentropy', 'config': {'weight_mul': 3.0}}, 'enable_multigpu': True,
                       'metrics': [{'Precision': {'thresholds': 0.5, 'name': 'pixel/precision'}},
                                   {'AUC': {'curve': 'PR', 'name': 'pixel/AUPRC', 'num_thresholds': 150}},
                                   {'AUC': {'curve': 'ROC', 'name': 'pixel/AUROC', 'num_thresholds': 150}}]},
}


def main():
    out_dir = sys.argv[1]
    ctx = distributed.context()
    dist.init_process_group('gloo', rank=ctx.rank, world_size=ctx.world)
    FakeDeviceModel.dist = dist

    # the engine is used unmodified; only the device layer is replaced
    device.init_device = lambda ordinal=0: None
    device.device_count = lambda: 1
    device.DeviceModel = FakeDeviceModel
    built = []

    def build(self, input_shape, max_batch=None, seed=None, force_generic=False):
        b, h, w, c = input_shape
        self.device_model = FakeDeviceModel(self.arch, c, h, w, max_batch or b, **{k: v for k, v in self.configs.items()
                                                                                    if k in OPTS})
        self.device_model.init_glorot(seed=(seed or 0) + 100 * ctx.rank)     # ranks start DIFFERENT: the broadcast must fix it
        built.append(self.device_model.max_batch)
        return self.device_model
    models.UNetAnnotator.build = build

    # sizes (tests/test_dp_gloo.py sets them for the world-4 case: evaluation batches of 10 -> shards 3 + 3 + 2 + 2)
    train_batch, val_n, val_batch = (int(os.environ.get(k, d)) for k, d in (('DP_TRAIN_BATCH', 4), ('DP_VAL_N', 11), ('DP_VAL_BATCH', 6)))
    fail_rank, fail_step = int(os.environ.get('DP_FAIL_RANK', -1)), int(os.environ.get('DP_FAIL_STEP', 2))
    if ctx.rank == fail_rank:
        # a rank-local failure in the middle of a step (the reference's label assertion, utils/losses.py:91-99, is rank-local too):
        # this rank raises BEFORE its all-reduce, its siblings are left waiting inside theirs -- the launcher must end them
        real_step = FakeDeviceModel.train_step

        def failing_step(self, x, y, lr, cfg):
            if self.iterations + 1 == fail_step:
                raise RuntimeError('injected failure on rank %d in step %d' % (ctx.rank, fail_step))
            return real_step(self, x, y, lr, cfg)
        FakeDeviceModel.train_step = failing_step

    H = W = 16
    xt, yt = O.synthetic_batch(8, H, W, 1, seed_x=3, seed_y=4)
    yt[1::2] = 0.0
    yt[1::2, 2:5, 3:6] = 1.0                                   # the two shards of a batch see different positive rates
    train = data.ArrayDataset(xt, yt, train_batch, repeat=True)
    xv, yv = O.synthetic_batch(val_n, H, W, 1, seed_x=5, seed_y=6)     # validation (default): batch 6 then a last batch of 5 (remainder)
    yv[7] = 0.0
    val = data.ArrayDataset(xv, yv, val_batch)
    save = os.path.join(out_dir, 'run')

    m = engine.TFKerasModel(CONFIG)
    res = m.train(train, val_data=val, save_path=save, max_steps=4, save_freq=2)
    dm = m.device_model
    result = dict(rank=ctx.rank, built=list(built), max_batch=dm.max_batch, calls=dm.calls,
                  params=dm.get_params().astype(np.float64).tolist(), state=dm.get_state().astype(np.float64).tolist(),
                  loss=res.history['loss'], val_loss=res.history.get('val_loss'),
                  val_auprc=res.history.get('val_pixel/AUPRC'), val_precision=res.history.get('val_pixel/precision'),
                  files=sorted(os.listdir(os.path.join(save, 'checkpoints'))) if os.path.isdir(os.path.join(save, 'checkpoints')) else None)
    ev = m._evaluate(val)
    result['eval'] = {k: v for k, v in ev.items()}

    # the same job on datasets that hand every rank its part already (tfrecord.TFRecordDataset(shard=...): pre_sharded, the
    # element spec still names the global batch): the engine must not split again
    class PreSharded:
        pre_sharded = True

        def __init__(self, ds):
            self.ds, self.element_spec = ds, ds.element_spec

        def __iter__(self):
            for x, y in self.ds:
                lo, hi = distributed.shard_bounds(len(x), ctx.rank, ctx.world, even=False)
                yield x[lo:hi], y[lo:hi]

    m2 = engine.TFKerasModel(CONFIG)
    res2 = m2.train(PreSharded(train), val_data=PreSharded(val), save_path=save + '_presharded', max_steps=4, save_freq=2)
    result['loss_presharded'], result['val_loss_presharded'] = res2.history['loss'], res2.history.get('val_loss')
    result['auc_counts'] = m.metrics[1].counts.tolist()
    dist.barrier()
    if ctx.rank == 0:
        # single-process emulation of the whole job with the oracle: replicas = contiguous shards, rank-local loss weight and
        # BatchNorm statistics, summed gradients / world, identical Adam; BatchNorm state averaged at the checkpoints
        spec = O.ModelSpec('unet', 1, **OPTS)
        # rank 0's initial weights as they travel in the broadcast (float32)
        p = [{n: v.astype(np.float32).astype(np.float64) for n, v in O.init_params(spec, seed=0, dtype=np.float64).items()}
             for _ in range(ctx.world)]
        mm, vv = {}, {}
        cfg = dict(weight_mul=3.0)
        losses, vals = [], []
        it = iter(train)
        for step in range(4):
            x, y = next(it)
            tot, lsum, states = None, 0.0, []
            for r in range(ctx.world):
                lo, hi = distributed.shard_bounds(len(x), r, ctx.world)
                l, g, _, st = O.loss_and_grads(spec, p[r], x[lo:hi].astype(np.float64), y[lo:hi], cfg, training=True)
                f = O.flatten(spec, g)
                tot = f if tot is None else tot + f
                lsum += l
                states.append(st)
            g = O.unflatten(spec, tot / ctx.world)
            new = O.adam_step({n: p[0][n] for n in g}, g, mm, vv, step + 1, 1e-3)
            for r in range(ctx.world):
                p[r].update(new)
                p[r].update(states[r])
            losses.append(lsum / ctx.world)
            if (step + 1) % 2 == 0:
                avg = {n: sum(p[r][n] for r in range(ctx.world)) / ctx.world for n in states[0]}     # comm_average_state
                for r in range(ctx.world):
                    p[r].update(avg)
                tot_l, cnt = 0.0, 0
                for x, y in val:
                    for r in range(ctx.world):
                        lo, hi = distributed.shard_bounds(len(x), r, ctx.world, even=False)
                        _, lg = O.predict(spec, p[r], x[lo:hi].astype(np.float64))
                        per, _ = O.weighted_crossentropy(y[lo:hi], lg, **cfg)
                        tot_l += float(per.mean()) * (hi - lo)
                        cnt += hi - lo
                vals.append(tot_l / cnt)
        result['ref_loss'], result['ref_val_loss'] = losses, vals
        result['ref_params'] = O.flatten(spec, p[0]).tolist()
        result['ref_state'] = O.flatten(spec, p[0], trainable=False).tolist()
        # metric counts over the WHOLE validation set, whatever the sharding: exact integers
        prob = np.concatenate([O.predict(spec, p[0], x.astype(np.float64))[0].astype(np.float32).ravel() for x, _ in val])
        yy = np.concatenate([y.ravel() for _, y in val]) > 0.5
        thr = m.metrics[1].thresholds
        result['ref_auc_counts'] = [[float(((prob > t) & yy).sum()), float(((prob > t) & ~yy).sum()),
                                     float((~(prob > t) & yy).sum()), float((~(prob > t) & ~yy).sum())] for t in thr]
        # resume in a fresh engine on rank 0's files happens in the parent test
    with open(os.path.join(out_dir, 'rank%d.json' % ctx.rank), 'w') as f:
        json.dump(result, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
