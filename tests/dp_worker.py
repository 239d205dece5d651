"""Worker of tests/test_dp_gloo.py: one of WORLD_SIZE CPU processes joined by torch.distributed (gloo).

Exercises the host side of the data-parallel path exactly as the GPU ranks run it -- rendezvous of the 128-byte
communicator id through distributed.exchange_unique_id, contiguous batch shards, per-replica loss weighting, one
all-reduce(sum) of the flat gradient vector, 1/world scaling, identical Adam on every rank -- with the numpy oracle
standing in for the HIP kernels and gloo standing in for RCCL."""

import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from dnncancerannotator_amd import distributed   # noqa: E402
from oracle import unet_oracle as O               # noqa: E402


class FakeIdSource:
    @staticmethod
    def comm_unique_id():
        return bytes(np.random.default_rng(os.getpid()).integers(0, 256, 128, dtype=np.uint8))


def main():
    out_dir = sys.argv[1]
    ctx = distributed.context()
    dist.init_process_group('gloo', rank=ctx.rank, world_size=ctx.world)
    result = {}

    # (a) unique-id rendezvous: every rank ends up with rank 0's bytes
    uid = distributed.exchange_unique_id(ctx, FakeIdSource, timeout=60)
    gathered = [None] * ctx.world
    dist.all_gather_object(gathered, uid.hex())
    result['uid_equal'] = len(set(gathered)) == 1 and len(uid) == 128

    # (b) one data-parallel train step
    spec = O.ModelSpec('unet', 1, 3, 2, bn=True, padding='same')
    params = O.init_params(spec, seed=2, dtype=np.float64)
    x, y = O.synthetic_batch(4, 16, 16, 1)
    y[2:] = 0.0
    y[2, 4:6, 4:6] = 1.0                 # the two shards have different positive rates
    x = x.astype(np.float64)
    lo, hi = distributed.shard_bounds(len(x), ctx.rank, ctx.world)
    cfg = dict(weight_mul=3.0)
    loss, grads, _, state = O.loss_and_grads(spec, params, x[lo:hi], y[lo:hi], cfg, training=True)
    flat = torch.from_numpy(np.concatenate([O.flatten(spec, grads), [loss]]))
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)                    # the single collective of the step
    flat = flat.numpy() / ctx.world
    g = O.unflatten(spec, flat[:-1])
    new = O.adam_step({n: params[n] for n in g}, g, {}, {}, 1, 1e-3)
    result['loss'] = float(flat[-1])
    result['params_after'] = O.flatten(spec, dict(params, **new)).tolist()
    result['state_local'] = O.flatten(spec, dict(params, **state), trainable=False).tolist()

    if ctx.rank == 0:
        # single-process emulation of the replicas (per-replica positive rate and BN statistics, summed gradients)
        tot, lsum = None, 0.0
        for r in range(ctx.world):
            a, b = distributed.shard_bounds(len(x), r, ctx.world)
            l, gr, _, _ = O.loss_and_grads(spec, params, x[a:b], y[a:b], cfg, training=True, n_replicas=ctx.world)
            f = O.flatten(spec, gr)
            tot = f if tot is None else tot + f
            lsum += l
        ref = O.adam_step({n: params[n] for n in g}, O.unflatten(spec, tot), {}, {}, 1, 1e-3)
        result['ref_loss'] = lsum
        result['ref_params_after'] = O.flatten(spec, dict(params, **ref)).tolist()
        # what a single replica over the whole batch would give (global positive rate / BN statistics): differs
        lg, gg, _, _ = O.loss_and_grads(spec, params, x, y, cfg, training=True)
        result['global_batch_loss'] = lg
    distributed.cleanup(ctx)
    with open(os.path.join(out_dir, 'rank%d.json' % ctx.rank), 'w') as f:
        json.dump(result, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
