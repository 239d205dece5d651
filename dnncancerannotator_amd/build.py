"""Builds libdnnca.so (hand-written HIP for gfx950) in-tree with hipcc.  No torch, no cmake."""

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libdnnca.so')
SOURCES = ['model.hip', 'kernels_generic.hip', 'kernels_mfma.hip', 'kernels_fused.hip', 'kernels_fused_bwd.hip', 'kernels_misc.hip', 'kernels_igemm.hip', 'kernels_ig3x.hip', 'kernels_first.hip', 'kernels_aug.hip', 'debug_tools.hip']
# -amdgpu-kernarg-preload-count: the first 16 dwords of a kernel's arguments arrive in SGPRs with the wave instead of through a cold
# scalar load at its top (every launch of the unet.yaml step starts ~0.15 us earlier: 0.411 -> 0.409 ms per step, A/B on one box)
FLAGS = ['-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-Wall', '-Wno-unused-result', '-mllvm', '-amdgpu-kernarg-preload-count=16']
# per-file extras.  kernels_mfma.hip: no SLP vectorizer -- it packs the scalar FMA chains of k_bwd3v into v_pk_fma_f32 (no faster
# than two v_fma_f32 on gfx950, and the even-aligned register pairs cost hundreds of v_mov and spills)
EXTRA_FLAGS = {'kernels_mfma.hip': ['-fno-slp-vectorize']}
# host-only C++ (no device pass): compiled by the same driver as plain C++
HOST_SOURCES = ['host_util.cpp']
HOST_FLAGS = ['-O3', '-std=c++17', '-fPIC', '-Wall']


def _newer(src_list, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in src_list)


def build_library(force=False, verbose=False):
    """Compile every .hip source to an object and link libdnnca.so.  Returns the library path."""
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    # DNNCA_TUNING=1: the tuning build (in-kernel stamps, phase switches) lives beside the shipped one -- objects *_t.o,
    # libdnnca_tuning.so; select it at run time with DNNCA_LIB=<path>
    tuning = bool(os.environ.get('DNNCA_TUNING'))
    osuf = '_t.o' if tuning else '.o'
    lib_path = os.path.join(HERE, 'libdnnca_tuning.so') if tuning else LIB
    # DNNCA_BUILD_TAG=<tag> DNNCA_EXTRA_FLAGS="...": an A/B variant of the shipped build (objects *_<tag>.o, libdnnca_<tag>.so)
    tag, extra = os.environ.get('DNNCA_BUILD_TAG'), os.environ.get('DNNCA_EXTRA_FLAGS', '').split()
    if tag and not tuning:
        osuf, lib_path = '_%s.o' % tag, os.path.join(HERE, 'libdnnca_%s.so' % tag)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    headers.append(os.path.join(HERE, '..', 'include', 'dnnca.h'))
    objs = []
    procs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace('.hip', osuf))
        objs.append(o)
        if force or _newer([s] + headers, o):
            cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + (['-DDNNCA_TUNING'] if tuning else []) + extra + ['-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src in HOST_SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace('.cpp', osuf))
        objs.append(o)
        if force or _newer([s] + headers, o):
            cmd = [hipcc, '-x', 'c++'] + HOST_FLAGS + ['-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write('hipcc failed for %s:\n%s\n' % (src, out.decode(errors='replace')))
        elif verbose and out:
            print(out.decode(errors='replace'))
    if failed:
        raise RuntimeError('libdnnca build failed')
    if force or procs or _newer(objs, lib_path):
        cmd = [hipcc, '-shared', '-fPIC', '--offload-arch=gfx950', '-o', lib_path] + objs + ['-L/opt/rocm/lib', '-lrccl']
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return lib_path


if __name__ == '__main__':
    print(build_library(force='--force' in sys.argv, verbose=True))
