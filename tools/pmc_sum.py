#!/usr/bin/env python3
"""pmc_sum.py <counter_collection.csv> -- per-kernel sum and per-launch mean of the counter(s) in a rocprofv3 --pmc pass."""
import csv
import re
import sys
from collections import defaultdict

tot, cnt = defaultdict(float), defaultdict(int)
with open(sys.argv[1]) as f:
    for row in csv.DictReader(f):
        k = re.sub(r'\(.*', '', row['Kernel_Name'])
        k = re.sub(r'^void ', '', k)
        key = (k, row['Counter_Name'])
        tot[key] += float(row['Counter_Value'])
        cnt[key] += 1
for (k, c), v in sorted(tot.items(), key=lambda kv: -kv[1])[:40]:
    print('%-70s %-28s launches %5d  mean %14.1f' % (k[:70], c, cnt[(k, c)], v / cnt[(k, c)]))
