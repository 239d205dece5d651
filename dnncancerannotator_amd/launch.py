"""`python -m dnncancerannotator_amd.launch --nproc N <subcommand> ...`: one worker process per GPU of this node.

Replaces the in-process tf.distribute.MirroredStrategy of the reference (engine.py:260-263): exports RANK, LOCAL_RANK,
WORLD_SIZE and a rendezvous key, starts `python -m dnncancerannotator_amd <subcommand> ...` N times and supervises them.
MirroredStrategy lives in one process and cannot half-die; N processes can: a rank that exits non-zero (a label assertion,
DNNCA_EASSERT, is rank-local) would leave its siblings waiting inside ncclAllReduce for ever.  So the launcher polls all
children, and on the first non-zero exit terminates the others (SIGTERM, then SIGKILL after a grace period) and returns
that exit code.  Children are only ever started fresh; nothing is re-executed."""

import argparse
import os
import subprocess
import sys
import time
import uuid


def supervise(procs, poll=0.1, grace=10.0):
    """Wait for every child; on the first non-zero exit stop the rest and return that code (0 when all succeed)."""
    alive = list(procs)
    while alive:
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0:
                for q in alive:
                    q.terminate()
                deadline = time.time() + grace
                for q in alive:
                    try:
                        q.wait(timeout=max(0.0, deadline - time.time()))
                    except subprocess.TimeoutExpired:
                        q.kill()
                        q.wait()
                return code
        if alive:
            time.sleep(poll)
    return 0


def main(argv=None, module='dnncancerannotator_amd'):
    ap = argparse.ArgumentParser(prog='python -m dnncancerannotator_amd.launch')
    ap.add_argument('--nproc', type=int, required=True, help='number of GPUs / worker processes on this node')
    ap.add_argument('rest', nargs=argparse.REMAINDER, help='annotator sub-command and its arguments')
    args = ap.parse_args(argv)
    key = uuid.uuid4().hex
    procs = []
    for rank in range(args.nproc):
        # HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC, which RCCL needs on this platform) goes into the child's environment, where it
        # takes effect before process start whatever the worker imports first; _lib.load() sets it too (for ranks started by
        # torch.distributed.run) but can only do so ahead of the first HIP call.  A value the caller exported wins.
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.nproc), DNNCA_RDZV_KEY=key)
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen([sys.executable, '-m', module] + args.rest, env=env))
    return supervise(procs)


if __name__ == '__main__':
    sys.exit(main())
