#!/usr/bin/env python3
"""Load-step periods of a rocprofv3 kernel trace of bench.py: time between the ends of consecutive k_update_state launches, kernel-busy
time inside each period and the largest gaps.   python tools/step_periods.py kernel_trace.csv"""
import csv
import re
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r'^void ', '', r['Kernel_Name']).replace('plfx::', '').split('(')[0]
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), n))
rows.sort()
ends = [i for i, r in enumerate(rows) if r[2].startswith('k_update_state')]
for a, b in zip(ends[:-1], ends[1:]):
    seg = rows[a + 1:b + 1]
    period = (rows[b][1] - rows[a][1]) / 1e3
    busy = sum(e - s for s, e, _ in seg) / 1e3
    gaps = sorted(((seg[i][0] - (rows[a][1] if i == 0 else seg[i - 1][1])) / 1e3, seg[i][2]) for i in range(len(seg)))[-3:]
    print('period %8.1f us  busy %7.1f us  kernels %3d  largest gaps before: %s' % (period, busy, len(seg), ', '.join('%s %.1f' % (n, g) for g, n in reversed(gaps))))
