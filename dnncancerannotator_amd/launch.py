"""`python -m dnncancerannotator_amd.launch --nproc N <subcommand> ...`: one worker process per GPU of this node.

Replaces the in-process tf.distribute.MirroredStrategy of the reference (engine.py:260-263): exports RANK, LOCAL_RANK,
WORLD_SIZE and a rendezvous key, starts `python -m dnncancerannotator_amd <subcommand> ...` N times and waits."""

import argparse
import os
import subprocess
import sys
import uuid


def main(argv=None):
    ap = argparse.ArgumentParser(prog='python -m dnncancerannotator_amd.launch')
    ap.add_argument('--nproc', type=int, required=True, help='number of GPUs / worker processes on this node')
    ap.add_argument('rest', nargs=argparse.REMAINDER, help='annotator sub-command and its arguments')
    args = ap.parse_args(argv)
    key = uuid.uuid4().hex
    procs = []
    for rank in range(args.nproc):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.nproc), DNNCA_RDZV_KEY=key,
                   HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, '-m', 'dnncancerannotator_amd'] + args.rest, env=env))
    code = 0
    for p in procs:
        code = p.wait() or code
    return code


if __name__ == '__main__':
    sys.exit(main())
