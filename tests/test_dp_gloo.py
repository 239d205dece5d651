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
