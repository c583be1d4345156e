#!/usr/bin/env python3
"""Per-load-step listing of a rocprofv3 kernel trace of bench.py: the kernels between two k_update_state launches, in order,
with duration and the gap before each; coarse-level kernels of the V-cycle are folded into one line per cycle.
    python tools/step_listing.py gpurun_out/r04_step/kernel_trace.csv [step index from the end, default 3]"""
import csv
import re
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name']
    n = re.sub(r'^void ', '', n).replace('plfx::', '')
    n = n.split('(')[0]
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), n, int(r['Grid_Size_X'])))
rows.sort()
ends = [i for i, r in enumerate(rows) if r[2].startswith('k_update_state')]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
a, b = ends[-k - 1] + 1, ends[-k] + 1
# extend to the post of that step
while b < len(rows) and rows[b][2] in ('k_gather', 'k_reduce_rows', 'k_mbox_post'):
    b += 1
a0 = a
while rows[a0][2] in ('k_gather', 'k_reduce_rows', 'k_mbox_post'):
    a0 += 1
seg = rows[a0:b]
t0 = seg[0][0]
print('load step: %d kernels, %.1f us from first start to last end' % (len(seg), (seg[-1][1] - t0) / 1e3))
busy = 0.
fold = None
out = []
prev_end = seg[0][0]
COARSE = ('k_mg_smooth<0', 'k_mg_smooth2_zero<0', 'k_mg_residual<0', 'k_mg_tail', 'k_mg_restrict', 'k_mg_prolong_add')
fine_grid = max(r[3] for r in seg if r[2].startswith('k_mg_smooth<1')) if any(r[2].startswith('k_mg_smooth<1') for r in seg) else 0
for (s, e, n, g) in seg:
    d = (e - s) / 1e3
    gap = (s - prev_end) / 1e3
    busy += d
    coarse = n.startswith(COARSE) and not (n in ('k_mg_restrict', 'k_mg_prolong_add') and g >= fine_grid)
    if coarse:
        if fold is None:
            fold = [0, 0., 0., s]
        fold[0] += 1
        fold[1] += d
        fold[2] += gap
        fold.append(e)
    else:
        if fold is not None:
            out.append('   %-34s %3d launches  busy %7.1f us  gaps %6.1f us  span %7.1f us' % ('[coarse levels of a V-cycle]', fold[0], fold[1], fold[2], (fold[-1] - fold[3]) / 1e3))
            fold = None
        out.append('   %-34s grid %8d  %7.1f us   gap before %6.1f us' % (n, g, d, gap))
    prev_end = e
print('\n'.join(out))
print('busy %.1f us' % busy)
