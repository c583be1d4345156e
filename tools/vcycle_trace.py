#!/usr/bin/env python3
"""Per-launch view of the V-cycle from a rocprofv3 kernel trace of tools/probes/vcycle_ab.py: the last `reps` cycles of the
tight loop, every launch of a cycle in order with its average duration and the average gap to its predecessor.
Usage: vcycle_trace.py <kernel_trace.csv> [cycles]"""
import collections
import csv
import re
import sys


def short(s):
    s = re.sub(r'^void ', '', s).replace('plfx::', '')
    return re.match(r'([A-Za-z_0-9]+(<[^>]*>)?)', s).group(1)


rows = list(csv.DictReader(open(sys.argv[1])))
ncyc = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), int(r.get('Grid_Size_X', r.get('Grid_Size', 0)) or 0))
            for r in rows)
# a cycle starts with the fine-level two-sweep kernel
starts = [i for i, e in enumerate(ev) if e[2].startswith('k_mg_smooth2_zero<1') or e[2].startswith('k_mg_smooth2_zero_march')]
starts = starts[-ncyc - 1:]
per = collections.OrderedDict()
length = starts[1] - starts[0]
tot = 0.
n = 0
for a, b in zip(starts[:-1], starts[1:]):
    if b - a != length:
        continue
    n += 1
    tot += (ev[b][0] - ev[a][0]) / 1e3
    for k in range(a, b):
        key = (k - a, ev[k][2], ev[k][3])
        d = per.setdefault(key, [0., 0.])
        d[0] += (ev[k][1] - ev[k][0]) / 1e3
        d[1] += (ev[k][0] - ev[k - 1][1]) / 1e3
print('%d cycles of %d launches, %.1f us per cycle (start to start)' % (n, length, tot / max(n, 1)))
print('%3s %-28s %9s %9s %9s' % ('#', 'kernel', 'grid', 'dur us', 'gap us'))
sd = sg = 0.
for (pos, name, grid), (d, g) in per.items():
    print('%3d %-28s %9d %9.2f %9.2f' % (pos, name, grid, d / n, g / n))
    sd += d / n
    sg += g / n
print('sum of durations %.1f us, sum of gaps %.1f us' % (sd, sg))
