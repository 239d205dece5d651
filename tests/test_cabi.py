"""CPU: libdnnca.so builds for gfx950, loads, and exports every entry point include/dnnca.h declares (no compute calls).
The library is opened in a child process: torch (used by other CPU tests) bundles its own ROCm runtime."""

import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, 'include', 'dnnca.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(dnnca_[a-z0-9_]+)\s*\(', text)))


def test_build_and_exports():
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import __graft_entry__ as g; g.build()\n"
        "from dnncancerannotator_amd import _lib\n"
        "lib = _lib.load()\n"
        "print('VERSION', lib.dnnca_version().decode())\n"
        "print('SYMS', ' '.join(sorted(_lib.SIGNATURES)))\n"
        "import ctypes as C\n"
        "n = C.c_int(-1); rc = lib.dnnca_device_count(C.byref(n)); print('DEVCOUNT', rc, n.value)\n"
        "h = C.c_void_p(); d = _lib.ModelDesc(); rc = lib.dnnca_model_create(C.byref(d), C.byref(h))\n"
        "print('CREATE_RC', rc, lib.dnnca_last_error().decode()[:60])\n" % ROOT)
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = dict(l.split(' ', 1) for l in out.stdout.strip().splitlines() if ' ' in l)
    assert lines['VERSION'].startswith('dnnca')
    bound = lines['SYMS'].split()
    declared = header_functions()
    assert declared and sorted(bound) == declared, set(bound) ^ set(declared)
    # the product path fails loudly without a usable model description / device: no silent CPU fallback
    assert int(lines['CREATE_RC'].split()[0]) < 0


def test_every_declared_symbol_is_exported():
    lib = os.path.join(ROOT, 'dnncancerannotator_amd', 'libdnnca.so')
    if not os.path.exists(lib):
        sys.path.insert(0, ROOT)
        from dnncancerannotator_amd import build
        build.build_library()
    nm = subprocess.run(['nm', '-D', '--defined-only', lib], capture_output=True, text=True).stdout
    exported = set(re.findall(r' T (dnnca_[a-z0-9_]+)', nm))
    assert set(header_functions()) <= exported, set(header_functions()) - exported
