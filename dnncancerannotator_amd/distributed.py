"""Process-level plumbing of the data-parallel path: rank discovery, RCCL unique-id exchange, batch sharding.

The reference gets data parallelism from tf.distribute.MirroredStrategy inside one process (engine.py:260-263).  Here it
is one process per GPU: every rank owns a libdnnca model handle, the handles are joined by dnnca_comm_init, and the only
data-path collective is ONE ncclAllReduce (RCCL over xGMI) of the flat gradient vector per step, inside libdnnca.
The 128-byte RCCL unique id is the only thing that has to travel between processes before that; it goes through a file
(no torch in the workers: torch bundles its own ROCm runtime and must not share a process with libdnnca)."""

import os
import time
from collections import namedtuple

Context = namedtuple('Context', ['rank', 'local_rank', 'world', 'rdzv_key'])


def context(environ=None):
    """RANK / LOCAL_RANK / WORLD_SIZE as exported by `python -m dnncancerannotator_amd.launch` or torch.distributed.run."""
    env = os.environ if environ is None else environ
    world = int(env.get('WORLD_SIZE', 1))
    rank = int(env.get('RANK', 0))
    local_rank = int(env.get('LOCAL_RANK', rank))
    key = env.get('DNNCA_RDZV_KEY') or '%s_%s_%s' % (os.getppid(), env.get('MASTER_PORT', '0'), env.get('TORCHELASTIC_RUN_ID', 'none'))
    if not 0 <= rank < world:
        raise ValueError('RANK %d outside WORLD_SIZE %d' % (rank, world))
    return Context(rank, local_rank, world, key)


def rendezvous_path(ctx, tag=''):
    return os.path.join(os.environ.get('TMPDIR', '/tmp'), 'dnnca_rdzv_%s%s.id' % (ctx.rdzv_key, '_' + tag if tag else ''))


_published = []          # rendezvous files this process wrote (rank 0): removed by cleanup()


def exchange_unique_id(ctx, id_source, timeout=300.0, force=False, tag=''):
    """Rank 0 creates the RCCL unique id (`id_source.comm_unique_id()`) and publishes it atomically in a file named
    after the launcher; the other ranks poll for it.  Returns the id bytes (None when world == 1, unless `force`: the
    one-rank rehearsal of the same path -- rank 0 then reads its own file back).  `tag` keeps several communicators of
    one job apart."""
    if ctx.world == 1 and not force:
        return None
    path = rendezvous_path(ctx, tag)
    if ctx.rank == 0:
        uid = id_source.comm_unique_id()
        tmp = '%s.tmp%d' % (path, os.getpid())
        with open(tmp, 'wb') as f:
            f.write(uid)
        os.replace(tmp, path)
        _published.append(path)
        if ctx.world > 1:
            return uid
    t0 = time.time()
    while True:
        try:
            if os.path.getmtime(path) > t0 - 600:          # ignore a stale file of an earlier job with the same key
                with open(path, 'rb') as f:
                    uid = f.read()
                if len(uid) == 128:
                    return uid
        except OSError:
            pass
        if time.time() - t0 > timeout:
            raise RuntimeError('timed out waiting for the RCCL unique id at %s' % path)
        time.sleep(0.05)


def cleanup(ctx):
    paths = list(_published) + ([rendezvous_path(ctx)] if ctx.world > 1 and ctx.rank == 0 else [])
    del _published[:]
    for path in paths:
        try:
            os.remove(path)
        except OSError:
            pass


def shard_bounds(n, rank, world, even=True):
    """[begin, end) of this rank's contiguous shard of a global batch of n.  even=True: n must divide evenly (a training
    batch under MirroredStrategy [TF-2.6]); even=False: the first n % world ranks take one extra sample (the last, partial
    batch of an evaluation set -- no sample is dropped)."""
    if n % world and even:
        raise ValueError('global batch %d is not divisible by %d ranks' % (n, world))
    per, rem = divmod(n, world)
    lo = rank * per + min(rank, rem)
    return lo, lo + per + (1 if rank < rem else 0)
