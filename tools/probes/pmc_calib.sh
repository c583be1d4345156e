#!/bin/bash
# run on the GPU box from the repo root: tools/probes/pmc_calib.sh <tag>  -> gpurun_out/<tag>/pmc_calib.txt
set -u
O=gpurun_out/${1:-calib}
mkdir -p $O
export TMPDIR=/tmp
B=tools/probes/pmc_calib
[ -x $B ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o $B $B.hip
$B > $O/pmc_calib.txt 2>&1
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --output-format csv -d $O/cal_$ctr -o cal -- $B > /dev/null 2> $O/cal_$ctr.err
  python - "$O/cal_$ctr" $ctr >> $O/pmc_calib.txt <<'PY'
import csv, glob, sys, collections
d, ctr = sys.argv[1], sys.argv[2]
f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(list)
for row in csv.DictReader(open(f[0])):
    if row['Counter_Name'] == ctr:
        acc[row['Kernel_Name'].split('(')[0]].append(float(row['Counter_Value']))
for k, v in acc.items():
    print('%s %-10s launches %d  mean counter value %.4e (min %.4e max %.4e)' % (ctr, k, len(v), sum(v) / len(v), min(v), max(v)))
PY
done
rm -rf $O/cal_FETCH_SIZE $O/cal_WRITE_SIZE
cat $O/pmc_calib.txt
