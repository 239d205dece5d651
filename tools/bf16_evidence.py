"""per-tensor gradient error of the bf16 kernels against the bf16-emulating oracle, NW = 4 and NW = 8 (GPU box)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for nw in ('4', '8'):
    for f0, S, extra in ((64, 24, []), (64, 16, []), (32, 24, []), (64, 20, ['3']), (64, 24, ['2', '1']), (64, 16, ['2', '1']), (64, 32, ['2', '1', '1', '4']), (64, 24, ['2', '1', '32', '1'])):
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'bf16_emul_case.py'), str(f0), str(S)] + extra,
                           env=dict(os.environ, DNNCA_IGB_NW=nw), capture_output=True, text=True, timeout=600)
        if r.returncode:
            print('FAILED', nw, f0, S, r.stderr[-800:])
            continue
        o = json.loads(r.stdout.strip().splitlines()[-1])
        pt = o.pop('per_tensor')
        w = sorted(pt.items(), key=lambda kv: -kv[1])[:4]
        print('NW', nw, 'f0', f0, 'S', S, extra, 'dl_max %.1e dl_med %.1e loss %.5f/%.5f l2 %.1e' % (o['dl_max'], o['dl_median'], o['loss'], o['loss_ref'], o['err_l2']),
              'worst', [(n, '%.1e' % e) for n, e in w], [n for n in o['plan'] if n.startswith('igb_conv')], flush=True)
