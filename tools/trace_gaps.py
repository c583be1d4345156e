#!/usr/bin/env python3
"""Timeline view of a rocprofv3 kernel trace of bench.py: GPU-busy vs idle time inside the timed load steps and the
largest idle gaps (which pair of kernels brackets them).  Usage: trace_gaps.py <kernel_trace.csv> [n_timed_sweeps]"""
import csv, re, sys, collections

def short(s):
    s = re.sub(r'^void ', '', s).replace('plfx::', '')
    return re.match(r'([A-Za-z_0-9]+(<[^>]*>)?)', s).group(1)

rows = list(csv.DictReader(open(sys.argv[1])))
nsw = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])) for r in rows)
sw = [i for i, e in enumerate(ev) if e[2].startswith('k_sweep_light')]
i0 = sw[-nsw - 1]
seg = ev[i0 + 2:]     # after the heavy kernel of the sweep preceding the timed ones
span = (seg[-1][1] - seg[0][0]) / 1e3
busy = sum(e - s for s, e, n in seg) / 1e3
print('span %.1f us  busy %.1f us (%.0f %%)  kernels %d' % (span, busy, 100 * busy / span, len(seg)))
gaps = collections.Counter(); gapn = collections.Counter(); big = []
for a, b in zip(seg[:-1], seg[1:]):
    g = (b[0] - a[1]) / 1e3
    gaps[(a[2], b[2])] += g; gapn[(a[2], b[2])] += 1
    if g > 30: big.append((round(g, 1), a[2], b[2]))
print('-- idle time by bracketing pair (us, count)')
for k, v in gaps.most_common(30): print('%9.1f %4d  %s -> %s' % (v, gapn[k], k[0], k[1]))
print('-- gaps > 30 us in order')
for g in big: print(g)
bk = collections.Counter()
for s, e, n in seg: bk[n] += (e - s) / 1e3
print('-- busy by kernel (us)')
for k, v in bk.most_common(20): print('%9.1f  %s' % (v, k))
