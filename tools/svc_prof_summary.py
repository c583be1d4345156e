#!/usr/bin/env python3
"""Summarise the rocprofv3 passes over tools/svc_profile.py: per-kernel time and SQ counters of the SVC sweep kernels.
Usage: tools/svc_prof_summary.py <dir with trace/ and pmc*/> <elements per corrector launch> <out.txt>"""
import csv
import glob
import re
import sys
from collections import defaultdict


def short(name):
    """k_sweep_svc_row<1, true> (tables in LDS) and <1, false> (tables in device memory) are one kernel family here: <1>"""
    m = re.search(r'plfx::(k_[a-zA-Z_0-9]+)(<[0-9a-z, ]+>)?', name)
    if not m:
        return name[:40]
    t = (m.group(2) or '').replace(' ', '')
    t = re.sub(r',(true|false)>$', '>', t)
    return m.group(1) + t


d, nel, out = sys.argv[1], float(sys.argv[2]), sys.argv[3]
L = []
rows = list(csv.DictReader(open(glob.glob('%s/trace/*kernel_stats.csv' % d)[0])))
L.append('== rocprofv3 --kernel-trace --stats: python tools/svc_profile.py (bounded config-4 sample) ==')
L.append('%-28s %8s %12s %12s %8s' % ('kernel', 'calls', 'avg_us', 'total_ms', 'pct'))
for r in rows[:14]:
    L.append('%-28s %8s %12.2f %12.3f %8.3f' % (short(r['Name']), r['Calls'], float(r['AverageNs']) / 1e3,
                                                float(r['TotalDurationNs']) / 1e6, float(r['Percentage'])))
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob('%s/pmc*/*counter_collection.csv' % d)):
    # every pass carries SQ_INSTS_VALU: the productive dispatches (> 1e6 VALU instructions) are selected pass by pass
    one = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = short(r['Kernel_Name'])
        if k.startswith('k_sweep_svc_wave') or k.startswith('k_sweep_svc_row'):
            one[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, c in one.items():
        keep = [i for i, v in enumerate(c.get('SQ_INSTS_VALU', [])) if v > 1e6]
        for name, v in c.items():
            if name in acc[k]:
                continue            # SQ_INSTS_VALU of a later pass
            acc[k][name] = [v[i] for i in keep] if (keep and len(v) == len(c['SQ_INSTS_VALU'])) else v
L.append('')
L.append('== rocprofv3 --pmc (SQ counters, average per dispatch; productive dispatches = those with > 1e6 VALU instructions) ==')
for k in sorted(acc):
    c = acc[k]
    L.append(k)
    for name in sorted(c):
        v = c[name]
        if v:
            L.append('   %-24s n=%4d  avg %16.1f' % (name, len(v), sum(v) / len(v)))
    keep = list(range(len(c.get('SQ_INSTS_VALU', []))))
    if keep and k.endswith('<1>'):
        iv = [c['SQ_INSTS_VALU'][i] for i in keep]
        L.append('   -> VALU wave-instructions per element update: %.1f (%g elements per launch)' % (sum(iv) / len(iv) / nel, nel))
        if all(n in c for n in ('SQ_INSTS_VALU_FMA_F64', 'SQ_INSTS_VALU_ADD_F64', 'SQ_INSTS_VALU_MUL_F64')):
            # productive dispatches of THAT pass (its own SQ_INSTS_VALU column has the same length)
            kp = list(range(len(c['SQ_INSTS_VALU_FMA_F64'])))
            fma = sum(c['SQ_INSTS_VALU_FMA_F64'][i] for i in kp) / len(kp)
            add = sum(c['SQ_INSTS_VALU_ADD_F64'][i] for i in kp) / len(kp)
            mul = sum(c['SQ_INSTS_VALU_MUL_F64'][i] for i in kp) / len(kp)
            L.append('   -> FP64 wave-instructions per element update: FMA %.1f  ADD %.1f  MUL %.1f  => %.4e FP64 flop per element update '
                     '((2 FMA + ADD + MUL) x 64 lanes)' % (fma / nel, add / nel, mul / nel, (2 * fma + add + mul) * 64. / nel))
        if 'SQ_ACTIVE_INST_VALU' in c and 'SQ_WAVE_CYCLES' in c:
            a = [c['SQ_ACTIVE_INST_VALU'][i] for i in keep]
            w = [c['SQ_WAVE_CYCLES'][i] for i in keep]
            L.append('   -> SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES = %.3f (share of the wave cycles in which a VALU instruction issues)'
                     % (sum(a) / sum(w)))
open(out, 'w').write('\n'.join(L) + '\n')
print('\n'.join(L))
