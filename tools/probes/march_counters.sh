#!/bin/bash
# SQ / TCP / TCC counters of the gather and marching forms of the fine-level smoother (tools/probes/march_probe <size>)
set -u
N=${1:-1024}; O=gpurun_out/${2:-mc}; mkdir -p $O; export TMPDIR=/tmp
B=tools/probes/march_probe
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM" \
           "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES TA_TA_BUSY" \
           "TCC_HIT TCC_MISS TCC_REQ TCC_EA0_RDREQ"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/p$i -o m -- $B $N > /dev/null 2> $O/p$i.err
done
python - $O > $O/counters_$N.txt <<'PY'
import csv, glob, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    if 'smooth' in k or 'spmv' in k:
        print(k)
        for c in sorted(acc[k]):
            v = acc[k][c]
            print('   %-34s n=%4d avg %16.1f' % (c, len(v), sum(v) / len(v)))
PY
rm -rf $O/p1 $O/p2 $O/p3 $O/p4
cat $O/counters_$N.txt
