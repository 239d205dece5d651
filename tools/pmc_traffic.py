#!/usr/bin/env python3
"""Post-processing of rocprofv3 counter-collection CSVs (one counter per pass, as MI355X_MICROARCH.md's HBM section prescribes).

    python tools/pmc_traffic.py traffic FETCH.csv WRITE.csv > profiles/roofline_traffic.json
        HBM-side bytes per launch and kernel = 2 x FETCH_SIZE + WRITE_SIZE (both reported in KiB; on gfx950 FETCH_SIZE
        tallies 128-byte requests at 64 bytes, hence the factor 2), keyed by the library's launch names.
    python tools/pmc_traffic.py ratio NUM.csv DEN.csv
        per-kernel ratio of two counters summed over launches (e.g. SQ_VALU_MFMA_BUSY_CYCLES over SQ_BUSY_CU_CYCLES).
"""
import csv
import json
import re
import sys
from collections import defaultdict


def short_name(k):
    """HIP kernel name -> the LAUNCH name used by libdnnca's profile table / bench.py."""
    k = re.sub(r'^void ', '', k)
    m = re.match(r'dnnca::k_pgbwd<(\d+), (\d+), (\d+), (true|false)((?:, \w+)*)>', k)
    if m:
        rest = [t.strip() for t in m.group(5).split(',') if t.strip()]      # NT, DB, VW, PF, TCF, TCM
        kind = 'tc_' if 'true' in rest[4:6] else ('pool_' if len(rest) >= 4 and rest[3] == 'true' else '')
        return 'pgbwd_%s%s%sx%s_%s' % ('' if m.group(4) == 'true' else 'w_', kind, m.group(1), m.group(2), m.group(3))
    m = re.match(r'dnnca::k_pgfwd<(\d+), (\d+), (\d+), \d+, (?:true|false)(, true|, false)?>', k)
    if m:
        return 'pgfwd_%s%sx%s_%s' % ('head_' if m.group(4) == ', true' else '', m.group(1), m.group(2), m.group(3))
    m = re.match(r'dnnca::ig3x::k_ig3x_conv3<(\d+), (\d+), (\d+)(?:, (\d+))?>', k)
    if m:      # channel tile 16 NN, mode (0 forward / 1 data gradient), waves, channel split
        return 'ig3x_conv_%s#x3n%sw%s%s' % ('dgrad' if m.group(2) == '1' else 'fwd', m.group(1), m.group(3), 's' if m.group(4) == '2' else '')
    m = re.match(r'dnnca::ig3x::k_ig3x_wgrad<(\d+), (\d+), (\d+), (\d+)>', k)
    if m:
        return 'ig3x_wgrad#x3m%sj%sn%sk%s' % m.groups()
    m = re.match(r'dnnca::k_(pool2_bwd|pool2_fwd|head_train|head_reduce)<(\d+)', k)
    if m:
        return '%s_%s' % m.groups()
    m = re.match(r'dnnca::k_(tconv2_fwd|tconv_bwd)<(\d+), (\d+)', k)
    if m:
        return '%s_%s_%s' % m.groups()
    m = re.match(r'dnnca::fz::k_fz_(down|up)<(\d+), (\d+)((?:, \w+)*)>', k)
    if m:
        extra = [t.strip() for t in m.group(4).split(',') if t.strip()]     # TW, TH, NT, MINW[, NF[, NOTC]]
        if len(extra) >= 6 and extra[5] == 'true':
            return 'fz_up2_%s' % m.group(3)
        ride = 'tc_' if m.group(1) == 'up' and len(extra) >= 5 and extra[4] != '0' else ''
        return 'fz_%s_%s%s_%s' % (m.group(1), ride, m.group(2), m.group(3))
    m = re.match(r'dnnca::fzb::k_fzb<(true|false), (\d+), (\d+)', k)
    if m:
        return 'fzb_up_%s' % m.group(3) if m.group(1) == 'true' else 'fzb_down_%s_%s' % (m.group(2), m.group(3))
    m = re.match(r'dnnca::k_bwd3v<(true|false)', k)
    if m:
        return 'bwd3v_pool_3x1_3' if m.group(1) == 'true' else 'bwd3v_3x1_3'
    for kern, name in (('k_tail3<', 'tail3_3x1_3'), ('k_first3_fwd<', 'first3_fwd'), ('k_first3<', 'first3_bwd'), ('k_up3_fwd<', 'up3_fwd')):
        if k.startswith('dnnca::' + kern):
            return name
    m = re.match(r'dnnca::(?:ig::|igb::|first::)?k_(\w+)', k)
    if m:
        return m.group(1)
    return k


def per_kernel(path):
    tot, cnt, counter = defaultdict(float), defaultdict(int), None
    with open(path) as f:
        for r in csv.DictReader(f):
            counter = r['Counter_Name']
            n = short_name(r['Kernel_Name'])
            tot[n] += float(r['Counter_Value'])
            cnt[n] += 1
    return tot, cnt, counter


def main():
    mode, a, b = sys.argv[1], sys.argv[2], sys.argv[3]
    ta, ca, na = per_kernel(a)
    tb, cb, nb = per_kernel(b)
    if mode == 'traffic':
        out = {}
        for k in sorted(ta, key=lambda k: -ta[k]):
            fetch_kib = ta[k] / ca[k]
            write_kib = tb.get(k, 0.0) / max(cb.get(k, 1), 1)
            out[k] = round((2.0 * fetch_kib + write_kib) * 1024.0, 1)
        json.dump(out, sys.stdout, indent=1)
        print()
    else:
        print('%-40s %14s %14s %8s' % ('kernel', na, nb, 'ratio'))
        for k in sorted(ta, key=lambda k: -tb.get(k, 0)):
            if tb.get(k, 0) > 0:
                print('%-40s %14.0f %14.0f %8.3f' % (k, ta[k], tb[k], ta[k] / tb[k]))


if __name__ == '__main__':
    main()
