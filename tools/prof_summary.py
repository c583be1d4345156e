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


def sweep_classes(d, dur, kernel='k_sweep_light<1>', nel_bytes=216.):
    """The return-mapping sweep moves 412 B per element + 216 B per REWRITTEN tangent: its launches fall into classes (no tangent
    rewritten / all rewritten / some), and a roofline fraction pairs the bytes of a class with the duration of THAT class.  The
    three rocprofv3 passes run the same deterministic command, so launch i of the kernel is the same sweep in each of them:
    WRITE_SIZE of launch i (PMC pass) classifies it, its duration comes from the kernel-trace pass (the PMC passes serialise
    and slow the kernels)."""
    per = {}
    for tag, cname in (('pmc_fetch', 'FETCH_SIZE'), ('pmc_write', 'WRITE_SIZE')):
        try:
            rows = [r for r in csv.DictReader(open('%s/%s/bench_counter_collection.csv' % (d, tag)))
                    if r['Counter_Name'] == cname and short(r['Kernel_Name']) == kernel]
        except IOError:
            return []
        rows.sort(key=lambda r: int(r['Start_Timestamp']))
        per[tag] = [float(r['Counter_Value']) * 1024 / 1e6 for r in rows]
    tr = [r for r in csv.DictReader(open('%s/trace/bench_kernel_trace.csv' % d)) if short(r['Kernel_Name']) == kernel]
    tr.sort(key=lambda r: int(r['Start_Timestamp']))
    du = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in tr]
    n = len(du)
    out = ['', '== %s launch classes (launch i paired across the passes: WRITE_SIZE / FETCH_SIZE of the PMC passes, duration of the kernel-trace pass) ==' % kernel]
    if not n or len(per['pmc_write']) != n or len(per['pmc_fetch']) != n:
        out.append('launch counts differ between the passes (trace %d, write %d, fetch %d): run the three passes with the same command'
                   % (n, len(per.get('pmc_write', [])), len(per.get('pmc_fetch', []))))
        return out
    w = per['pmc_write']
    wmin, wmax = min(w), max(w)
    span = wmax - wmin
    # rewritten share of a launch from its own WRITE_SIZE: 0 at the smallest write of the run; the largest write of the run is
    # taken as "all rewritten" only when it exceeds the smallest by the 216 B per element a full rewrite adds (checked by the reader:
    # span MB / 216 B = elements)
    cls = {'none': [], 'some': [], 'all': []}
    for i in range(n):
        f = (w[i] - wmin) / span if span > 1. else 0.
        cls['none' if f < 0.02 else 'all' if f > 0.98 else 'some'].append((f, du[i], w[i], per['pmc_fetch'][i]))
    out.append('%d launches; WRITE_SIZE %.2f .. %.2f MB (span %.2f MB = 216 B x %.0f elements)' % (n, wmin, wmax, span, span * 1e6 / nel_bytes))
    for k in ('none', 'some', 'all'):
        v = cls[k]
        if v:
            m = len(v)
            out.append('class %-5s n=%4d  rewritten %.3f  dur %8.2f us  write %8.2f MB  fetch %8.2f MB   (traffic 2 x fetch + write = %.2f MB)'
                       % (k, m, sum(x[0] for x in v) / m, sum(x[1] for x in v) / m, sum(x[2] for x in v) / m, sum(x[3] for x in v) / m,
                          (2 * sum(x[3] for x in v) + sum(x[2] for x in v)) / m))
    return out


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
    lines += sweep_classes(d, dur)
    open(out, 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
