#!/bin/bash
# same-box A/B of the grid of k_update_state<1> / k_global_partials (SUMPART): rocprofv3 kernel stats of a short bench run per library
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
O=gpurun_out/r04_sumpart
mkdir -p $O
for v in new2048 new4096 new8192 new2048 new4096; do
  export PLFX_LIB=$PWD/build/libplfx_$v.so
  rm -rf /tmp/sp_$v
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp_$v -o b -- python bench.py --no-tight-loop --no-cpu --no-inclusion --no-svc --no-2048 --steps 10 --warmup 2 > $O/bench_$v.json 2> $O/err_$v.txt
  F=$(find /tmp/sp_$v -name "b_kernel_stats.csv" | head -1)
  echo "== $v  $(python -c "import json;d=json.loads(open('$O/bench_$v.json').read().strip().splitlines()[-1]);print('ms_per_step %.4f'%d['ms_per_step'])")"
  python - "$F" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if any(k in n for k in ('k_update_state', 'k_reduce_rows', 'k_sweep_light', 'k_axpy_uf')):
        print('   %-28s calls %4s  avg %9.2f us  min %9.2f  max %9.2f' % (n.split('(')[0].replace('void plfx::', '').replace('plfx::', ''), r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
PY
done
