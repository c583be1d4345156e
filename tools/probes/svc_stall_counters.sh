#!/bin/bash
# Where the corrector kernel's wave cycles go: SQ wait / fetch / LDS counters of k_sweep_svc_row<1> on the bounded config-4 sample
#   tools/probes/svc_stall_counters.sh <tag>  -> gpurun_out/<tag>/svc_stalls.txt   (counter passes only: no trace domains)
set -u
TAG=${1:-svcstall}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL SQ_LDS_BANK_CONFLICT" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $O/p$i -o svc -- python tools/svc_profile.py 128 > /dev/null 2> $O/p$i.err
done
python - "$O" <<'PY'
import csv, glob, sys, collections
o = sys.argv[1]
acc = collections.defaultdict(float)
n = collections.defaultdict(int)
for f in glob.glob(o + '/p*/**/svc_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_sweep_svc_row<1' in r['Kernel_Name'] and int(r['End_Timestamp']) - int(r['Start_Timestamp']) > 1000000:
            acc[r['Counter_Name']] += float(r['Counter_Value'])
            n[r['Counter_Name']] += 1
out = ['== k_sweep_svc_row<1> (productive launches of tools/svc_profile.py 128): SQ counters, mean per launch ==']
wc = acc['SQ_WAVE_CYCLES'] / max(n['SQ_WAVE_CYCLES'], 1)
for k in sorted(acc):
    v = acc[k] / n[k]
    out.append('%-24s %16.4e   / SQ_WAVE_CYCLES = %.4f' % (k, v, v / wc if wc else 0.))
open(o + '/svc_stalls.txt', 'w').write('\n'.join(out) + '\n')
print('\n'.join(out))
PY
rm -rf $O/p1 $O/p2 $O/p3 $O/p4 $O/p5
