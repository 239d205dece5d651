#!/usr/bin/env python3
"""trace_overlap.py <kernel_trace.csv> -- how much do kernels overlap in time?  (rocprofv3 --kernel-trace output: sum of the kernel
durations against the length of the union of their intervals, per queue and in total, over the last third of the trace)"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'], r['Kernel_Name'].split('(')[0][-40:]) for r in rows)
t0 = ev[0][0] + 2 * (ev[-1][1] - ev[0][0]) // 3
ev = [e for e in ev if e[0] >= t0]
tot = sum(e[1] - e[0] for e in ev)
union, cur_s, cur_e = 0, None, None
for s, e, _, _ in ev:
    if cur_e is None or s > cur_e:
        if cur_e is not None: union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
span = ev[-1][1] - ev[0][0]
print('kernels %d, span %.2f ms, sum of durations %.2f ms, union %.2f ms, idle %.2f ms' % (len(ev), span / 1e6, tot / 1e6, union / 1e6, (span - union) / 1e6))
byq = collections.Counter()
for s, e, q, _ in ev: byq[q] += e - s
for q, v in sorted(byq.items()): print('queue %s: %.2f ms busy' % (q, v / 1e6))
