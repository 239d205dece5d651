"""CPU, world_size 2 over gloo: the N > 1 path of the engine (rank discovery, id rendezvous, shards, one all-reduce)."""

import json
import os
import socket
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_two_rank_data_parallel_step(tmp_path):
    world = 2
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), DNNCA_RDZV_KEY='pytest_%d' % port, TMPDIR=str(tmp_path), OMP_NUM_THREADS='2')
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, 'dp_worker.py'), str(tmp_path)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), '\n'.join(outs)
    r = [json.load(open(tmp_path / ('rank%d.json' % k))) for k in range(world)]
    assert r[0]['uid_equal'] and r[1]['uid_equal']
    # every rank holds identical weights after the step, equal to the replica emulation
    assert np.array_equal(r[0]['params_after'], r[1]['params_after'])
    assert np.abs(np.array(r[0]['params_after']) - np.array(r[0]['ref_params_after'])).max() < 1e-12
    assert abs(r[0]['loss'] - r[0]['ref_loss']) < 1e-12 and abs(r[0]['loss'] - r[1]['loss']) < 1e-15
    # per-replica semantics (MirroredStrategy): the loss weight and BN statistics come from the local shard, so the
    # result is NOT the single-replica global-batch loss, and BN moving statistics differ between ranks until averaged
    assert abs(r[0]['loss'] - r[0]['global_batch_loss']) > 1e-6
    assert np.abs(np.array(r[0]['state_local']) - np.array(r[1]['state_local'])).max() > 1e-9
    assert not os.path.exists(tmp_path / ('dnnca_rdzv_pytest_%d.id' % port))      # rank 0 removed the rendezvous file


def _spawn(worker, tmp_path, world=2, extra_env=None):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), DNNCA_RDZV_KEY='pytest_%d' % port, TMPDIR=str(tmp_path), OMP_NUM_THREADS='2')
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, worker), str(tmp_path)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=900)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), '\n'.join(outs)
    return [json.load(open(tmp_path / ('rank%d.json' % k))) for k in range(world)], port


def test_engine_world2(tmp_path):
    """engine.TFKerasModel.train / save / _evaluate on two gloo ranks (the engine's own world > 1 branches; the device layer
    is tests/fake_device.py = the oracle) against a single-process emulation of the replicas."""
    r, port = _spawn('dp_engine_worker.py', tmp_path)
    r0, r1 = r
    # sized once for the larger of (train 4, validation 6) / 2 ranks: a validation batch is ONE eval step per replica
    assert r0['built'] == [3] and r1['built'] == [3] and r0['max_batch'] == 3
    evals = [c for c in r0['calls'] if c[0] == 'eval']
    assert [c[1] for c in evals[:2]] == [3, 3] and [c[1] for c in r1['calls'] if c[0] == 'eval'][:2] == [3, 2]   # 6 -> 3+3, 5 -> 3+2: nothing dropped
    # identical weights on both ranks (rank 0's initial weights were broadcast; ranks started different), equal to the emulation
    assert np.array_equal(r0['params'], r1['params'])
    assert np.abs(np.array(r0['params']) - np.array(r0['ref_params'])).max() < 1e-6       # float32 get_params
    assert np.allclose(r0['loss'], r0['ref_loss'], rtol=1e-9) and np.allclose(r0['loss'], r1['loss'], rtol=0, atol=1e-15)
    # validation every save_freq steps: per-replica weights over the whole per-replica batch, summed over ranks
    assert np.allclose(r0['val_loss'], r0['ref_val_loss'], rtol=1e-9) and r0['val_loss'] == r1['val_loss']
    # BatchNorm moving statistics were averaged at the last checkpoint (step 4 = the end): equal on both ranks
    assert np.abs(np.array(r0['state']) - np.array(r1['state'])).max() < 1e-7
    assert np.abs(np.array(r0['state']) - np.array(r0['ref_state'])).max() < 1e-6
    # metric counts: 150 thresholds x 4 counters merged across ranks, exact integers, every validation pixel counted once
    assert r0['auc_counts'] == r0['ref_auc_counts'] == r1['auc_counts']
    assert np.array(r0['auc_counts']).sum(1).tolist() == [11 * 16 * 16] * 150
    assert r0['eval']['loss'] == r1['eval']['loss'] and 0.0 <= r0['eval']['pixel/AUROC'] <= 1.0
    # datasets that shard themselves (pre_sharded): the same losses, nothing is split twice
    assert r0['loss_presharded'] == r0['loss'] and r1['loss_presharded'] == r1['loss']
    assert r0['val_loss_presharded'] == r0['val_loss']
    # rank 0 alone wrote the checkpoints; the rendezvous file is gone
    assert r0['files'] == ['ckpt-2.data-00000-of-00001', 'ckpt-2.index', 'ckpt-4.data-00000-of-00001', 'ckpt-4.index']
    assert not os.path.exists(tmp_path / ('dnnca_rdzv_pytest_%d.id' % port))


def test_engine_world4_with_uneven_evaluation_shards(tmp_path):
    """The same job on FOUR gloo ranks: training batches of 8 (2 per replica), evaluation batches of 10 that do not divide --
    shards 3 + 3 + 2 + 2 (engine.py:260-263: MirroredStrategy splits whatever batch it is handed; no sample may be dropped or counted
    twice).  Same single-process emulation as the world-2 test."""
    r, port = _spawn('dp_engine_worker.py', tmp_path, world=4, extra_env=dict(DP_TRAIN_BATCH='8', DP_VAL_N='20', DP_VAL_BATCH='10',
                                                                             OMP_NUM_THREADS='1'))
    assert all(x['built'] == [3] and x['max_batch'] == 3 for x in r)
    per_rank = [[c[1] for c in x['calls'] if c[0] == 'eval'][:2] for x in r]
    assert per_rank == [[3, 3], [3, 3], [2, 2], [2, 2]], per_rank
    assert [c[1] for c in r[0]['calls'] if c[0] == 'train'][:4] == [2, 2, 2, 2]
    for x in r[1:]:
        assert np.array_equal(r[0]['params'], x['params'])
        assert np.allclose(r[0]['loss'], x['loss'], rtol=0, atol=1e-15) and r[0]['val_loss'] == x['val_loss']
        assert np.abs(np.array(r[0]['state']) - np.array(x['state'])).max() < 1e-7
        assert r[0]['auc_counts'] == x['auc_counts'] and r[0]['eval']['loss'] == x['eval']['loss']
        assert x['loss_presharded'] == x['loss']
    assert np.abs(np.array(r[0]['params']) - np.array(r[0]['ref_params'])).max() < 1e-6
    assert np.allclose(r[0]['loss'], r[0]['ref_loss'], rtol=1e-9)
    assert np.allclose(r[0]['val_loss'], r[0]['ref_val_loss'], rtol=1e-9)
    assert np.abs(np.array(r[0]['state']) - np.array(r[0]['ref_state'])).max() < 1e-6
    assert r[0]['auc_counts'] == r[0]['ref_auc_counts']
    assert np.array(r[0]['auc_counts']).sum(1).tolist() == [20 * 16 * 16] * 150          # every validation pixel once
    assert r[0]['files'] == ['ckpt-2.data-00000-of-00001', 'ckpt-2.index', 'ckpt-4.data-00000-of-00001', 'ckpt-4.index']
    assert not os.path.exists(tmp_path / ('dnnca_rdzv_pytest_%d.id' % port))


def test_a_rank_that_raises_mid_step_ends_the_job(tmp_path):
    """Four engine workers under launch.supervise; rank 2 raises in its second train step, in front of its all-reduce, so ranks
    0, 1 and 3 sit inside theirs.  The launcher must return that rank's non-zero exit code in bounded time with every process
    gone, and the rendezvous file of the job must not survive (it is removed right after the id exchange)."""
    import time
    sys.path.insert(0, os.path.dirname(HERE))
    from dnncancerannotator_amd import launch
    world, port = 4, _free_port()
    procs, logs = [], []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), DNNCA_RDZV_KEY='pytest_%d' % port, TMPDIR=str(tmp_path), OMP_NUM_THREADS='1',
                   DP_TRAIN_BATCH='8', DP_VAL_N='20', DP_VAL_BATCH='10', DP_FAIL_RANK='2', DP_FAIL_STEP='2')
        logs.append(open(tmp_path / ('log%d.txt' % rank), 'wb'))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, 'dp_engine_worker.py'), str(tmp_path)], env=env,
                                      stdout=logs[-1], stderr=subprocess.STDOUT))
    t0 = time.time()
    code = launch.supervise(procs, grace=5.0)
    for f in logs:
        f.close()
    assert code != 0, code
    assert time.time() - t0 < 300 and all(p.poll() is not None for p in procs)
    assert b'injected failure on rank 2 in step 2' in open(tmp_path / 'log2.txt', 'rb').read()
    assert not any(os.path.exists(tmp_path / ('rank%d.json' % k)) for k in range(world))          # nobody finished the job
    assert not [f for f in os.listdir(tmp_path) if f.startswith('dnnca_rdzv_')], os.listdir(tmp_path)


def test_shard_bounds_cover_every_sample():
    from dnncancerannotator_amd import distributed
    for n in range(0, 20):
        for world in (1, 2, 3, 8):
            b = [distributed.shard_bounds(n, r, world, even=False) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            assert max(hi - lo for lo, hi in b) - min(hi - lo for lo, hi in b) <= 1


def test_launcher_stops_siblings_when_a_rank_dies(tmp_path):
    """launch.supervise: a rank that exits non-zero must not leave its siblings waiting in a collective for ever."""
    import time
    sys.path.insert(0, os.path.dirname(HERE))
    from dnncancerannotator_amd import launch
    sleeper = [sys.executable, '-c', 'import time; time.sleep(600)']
    t0 = time.time()
    procs = [subprocess.Popen(sleeper), subprocess.Popen([sys.executable, '-c', 'import sys, time; time.sleep(0.5); sys.exit(7)']),
             subprocess.Popen(sleeper)]
    assert launch.supervise(procs, grace=5.0) == 7
    assert time.time() - t0 < 30 and all(p.poll() is not None for p in procs)
    assert launch.supervise([subprocess.Popen([sys.executable, '-c', 'pass']) for _ in range(3)]) == 0
