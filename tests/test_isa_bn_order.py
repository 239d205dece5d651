"""CPU: the self-folding BatchNorm reduction (csrc/bn_dev.h) is ordered by construction in the COMPILED gfx950 code.

bn_last_block's contract: every thread waits for its own bucket adds (no-return `global_atomic_add_f64`) with
`s_waitcnt vmcnt(0)` before the workgroup barrier in front of the ticket (`global_atomic_add ... sc0`, a returning atomic).
Round 3 relied on a workgroup-scope release fence for that, which emits no wait on gfx950 -- a latent race on every BatchNorm
of configs/unet_big.yaml and configs/mulmo_unet.yaml (components.py:57,59,130-131).  This test unbundles the device code of
every object whose source calls bn_last_block / bn_self_fold, disassembles it and checks, per kernel:

    last global_atomic_add_f64  <  s_waitcnt with vmcnt(0)  <  s_barrier  <  first returning global_atomic_add (the ticket)

in address order (the reduction tail of these kernels is straight-line code; a kernel whose adds sit in a loop still has the
loop's body in front of the wait).  No GPU, no compute calls."""

import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'dnncancerannotator_amd', 'csrc')
LLVM = '/opt/rocm/lib/llvm/bin'
TARGET = 'hipv4-amdgcn-amd-amdhsa--gfx950'


def _sources_using_bn_fold():
    out = []
    for f in sorted(os.listdir(CSRC)):
        if f.endswith('.hip') and re.search(r'\bbn_(last_block|self_fold)\s*\(', open(os.path.join(CSRC, f)).read()):
            out.append(f)
    return out


def _disassemble(obj, tmp):
    fat, co = os.path.join(tmp, 'fat.bin'), os.path.join(tmp, 'dev.co')
    subprocess.check_call([os.path.join(LLVM, 'llvm-objcopy'), '--dump-section', '.hip_fatbin=' + fat, obj, os.path.join(tmp, 'scratch.o')])
    subprocess.check_call([os.path.join(LLVM, 'clang-offload-bundler'), '--unbundle', '--type=o', '--input=' + fat,
                           '--targets=' + TARGET, '--output=' + co])
    return subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '-d', '--mcpu=gfx950', co], capture_output=True, text=True, check=True).stdout


def _kernels(asm):
    """{mangled name: [instruction text, ...]} of a disassembly"""
    out, cur = {}, None
    for line in asm.splitlines():
        m = re.match(r'^[0-9a-f]+ <(.+)>:', line)
        if m:
            cur = out.setdefault(m.group(1), [])
        elif cur is not None and line.startswith('\t'):
            cur.append(line.strip().split('//')[0].strip())
    return out


def check_kernel(ins):
    """None if the kernel has no self-folding reduction; else (ok, why)."""
    adds = [i for i, t in enumerate(ins) if t.startswith('global_atomic_add_f64')]
    tickets = [i for i, t in enumerate(ins) if re.match(r'global_atomic_add\s', t) and ' sc0' in t]
    if not adds or not tickets:
        return None
    last_add = adds[-1]
    after = [i for i in tickets if i > last_add]
    if not after:
        return False, 'no ticket behind the last bucket add'
    ticket = after[0]
    barriers = [i for i in range(last_add, ticket) if ins[i] == 's_barrier']
    if not barriers:
        return False, 'no s_barrier between the last bucket add and the ticket'
    waits = [i for i in range(last_add, barriers[0]) if ins[i].startswith('s_waitcnt') and 'vmcnt(0)' in ins[i]]
    if not waits:
        return False, 'no s_waitcnt vmcnt(0) between the last bucket add (+%d) and the barrier (+%d)' % (last_add, barriers[0])
    # nothing that could start a new vector-memory operation and leave before the barrier matters; but no branch may skip the wait
    return True, ''


@pytest.mark.timeout(900)
def test_bucket_adds_are_waited_for_before_the_ticket(tmp_path):
    sys.path.insert(0, ROOT)
    from dnncancerannotator_amd import build
    build.build_library()
    sources = _sources_using_bn_fold()
    assert {'kernels_misc.hip', 'kernels_igemm.hip', 'kernels_first.hip'} <= set(sources), sources
    checked, bad = [], []
    for src in sources:
        d = tmp_path / src
        d.mkdir()
        ks = _kernels(_disassemble(os.path.join(CSRC, src.replace('.hip', '.o')), str(d)))
        n_here = 0
        for name, ins in ks.items():
            r = check_kernel(ins)
            if r is None:
                continue
            n_here += 1
            checked.append(name)
            if not r[0]:
                bad.append((src, name, r[1]))
        assert n_here, '%s calls bn_self_fold but no compiled kernel shows a bucket add + ticket' % src
    assert not bad, bad
    # the families that use the helper (kernels_misc.hip: statistics, apply + pool, backward reduce; kernels_igemm.hip: the
    # conv / transposed-conv epilogues; kernels_first.hip: the one-channel first conv)
    for family in ('k_bn_stats_fast', 'k_bn_apply_pool_fast', 'k_bn_bwd_reduce_fast', 'k_igb_conv3', 'k_ig_conv3', 'k_igb_tconv_fwd', 'k_first_fwd'):
        assert any(family in n for n in checked), (family, len(checked))


def test_checker_rejects_the_round3_shape():
    """the sequence round 3 shipped (only lgkmcnt waited for) must fail, the fixed one must pass"""
    old = ['global_atomic_add_f64 v[6:7], v[4:5], off', 's_waitcnt lgkmcnt(0)', 's_barrier', 'global_atomic_add v2, v2, v3, s[6:7] offset:4 sc0']
    new = ['global_atomic_add_f64 v[6:7], v[4:5], off', 's_waitcnt vmcnt(0) lgkmcnt(0)', 's_barrier', 'global_atomic_add v2, v2, v3, s[6:7] offset:4 sc0']
    late = ['global_atomic_add_f64 v[6:7], v[4:5], off', 's_barrier', 's_waitcnt vmcnt(0)', 'global_atomic_add v2, v2, v3, s[6:7] offset:4 sc0']
    assert check_kernel(old)[0] is False
    assert check_kernel(late)[0] is False
    assert check_kernel(new)[0] is True
    assert check_kernel(['v_mov_b32 v0, v1']) is None
