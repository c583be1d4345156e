#!/bin/bash
# rocprofv3 kernel statistics of the whole config-5 solve (2048^2, 20 load steps): tools/probes/cfg5_profile.sh <tag>
set -u
TAG=${1:-cfg5prof}
O=gpurun_out/$TAG
mkdir -p $O
cd /tmp 2>/dev/null; cd - > /dev/null
export TMPDIR=/tmp
PLFX_MG_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o cfg5 -- python tools/configs_full.py 5full > $O/run.txt 2> $O/trace.err
python - "$O" <<'PY'
import csv, glob, sys, re
o = sys.argv[1]
f = glob.glob(o + '/trace/**/cfg5_kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
out = ['== rocprofv3 --kernel-trace --stats: python tools/configs_full.py 5full (2048^2 laminate, 20 load steps) ==',
       'GPU kernel time in total: %.2f s' % (tot / 1e9), '%-44s %9s %12s %12s %7s' % ('kernel', 'calls', 'avg_us', 'total_ms', 'pct')]
for r in rows[:40]:
    n = re.sub(r'^void plfx::', '', r['Name'])[:44]
    out.append('%-44s %9s %12.2f %12.1f %7.2f' % (n, r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6, float(r['Percentage'])))
open(o + '/cfg5_kernel_summary.txt', 'w').write('\n'.join(out) + '\n')
print('\n'.join(out))
PY
tail -2 $O/run.txt >> $O/cfg5_kernel_summary.txt
rm -rf $O/trace
