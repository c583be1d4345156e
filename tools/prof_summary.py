#!/usr/bin/env python3
"""Summarise rocprofv3 output (kernel stats + FETCH_SIZE/WRITE_SIZE PMC passes) into a small text file
for profiles/.  Usage: tools/prof_summary.py <dir with trace/, pmc_fetch/, pmc_write/> <out.txt>"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r'plfx::(k_[a-zA-Z_0-9]+(<[0-9, ]+>)?)', name)
    return m.group(1).replace(' ', '') if m else name[:40]


def main(d, out):
    lines = []
    rows = list(csv.DictReader(open('%s/trace/bench_kernel_stats.csv' % d)))
    lines.append('== rocprofv3 --kernel-trace --stats (per-kernel totals over the whole bench.py run) ==')
    lines.append('%-28s %8s %12s %12s %8s' % ('kernel', 'calls', 'avg_us', 'total_ms', 'pct'))
    for r in rows[:45]:
        lines.append('%-28s %8s %12.2f %12.3f %8.3f' % (short(r['Name']), r['Calls'], float(r['AverageNs']) / 1e3,
                                                        float(r['TotalDurationNs']) / 1e6, float(r['Percentage'])))
    # productive launches only (duration > 20 us filters the post-convergence no-op launches of the PCG kernels)
    dur = defaultdict(list)
    for r in csv.DictReader(open('%s/trace/bench_kernel_trace.csv' % d)):
        dur[short(r['Kernel_Name'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    lines.append('')
    lines.append('== fine-level / productive launches only (coarse-level and post-convergence no-op launches filtered by duration) ==')
    FINE = ('k_mg_smooth_march', 'k_mg_smooth2_zero_march', 'k_mg_residual_march', 'k_spmv_march<1>', 'k_spmv_march<2>', 'k_dot_rz',
            'k_mg_smooth<1,1>', 'k_mg_smooth2_zero<1,1>', 'k_mg_residual<1,1>', 'k_spmv<1,1>', 'k_spmv<2,1>',
            'k_spmv<0,1>', 'k_cg_start<1>', 'k_mg_smooth<1,0>', 'k_mg_smooth2_zero<1,0>', 'k_mg_residual<1,0>',
            'k_spmv<1,0>', 'k_spmv<0,0>', 'k_cg_update', 'k_cg_update_mg', 'k_sweep_light<1>', 'k_sweep_light<0>',
            'k_sweep_heavy<1>', 'k_grid_setup', 'k_grid_diag', 'k_assemble', 'k_mg_tail_mf', 'k_mg_tail_lds',
            'k_update_state<1>', 'k_update_state<0>', 'k_update_state', 'k_scf_elements', 'k_axpy_uf', 'k_bc_finish')
    for k in FINE:
        v = [x for x in dur.get(k, []) if x > 15.]
        if v:
            v.sort()
            lines.append('%-28s n=%6d  avg %9.2f us  median %9.2f us  min %9.2f  max %9.2f' %
                         (k, len(v), sum(v) / len(v), v[len(v) // 2], v[0], v[-1]))
    for tag, cname in (('pmc_fetch', 'FETCH_SIZE'), ('pmc_write', 'WRITE_SIZE')):
        acc = defaultdict(list)
        try:
            for r in csv.DictReader(open('%s/%s/bench_counter_collection.csv' % (d, tag))):
                if r['Counter_Name'] == cname:
                    acc[short(r['Kernel_Name'])].append((float(r['Counter_Value']),
                                                         int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
        except IOError:
            continue
        lines.append('')
        lines.append('== rocprofv3 --pmc %s (raw counter, KiB per dispatch; productive dispatches only) ==' % cname)
        for k in FINE:
            thr = 15000
            v = [x[0] for x in acc.get(k, []) if x[1] > thr]
            if v:
                lines.append('%-28s n=%6d  avg %14.1f KiB = %10.2f MB   (min %.2f MB, max %.2f MB)'
                             % (k, len(v), sum(v) / len(v), sum(v) / len(v) * 1024 / 1e6, min(v) * 1024 / 1e6, max(v) * 1024 / 1e6))
    open(out, 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
