#!/bin/bash
# same-box A/B of the end-of-step post: gather / reduce kernels writing the pinned slot themselves (PLFX_FINISH_DIRECT=1, default)
# against the staged copy through one workgroup (=0); rocprofv3 kernel stats of a short bench run each
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
O=gpurun_out/r04_finish
mkdir -p $O
for v in 0 1 0 1; do
  export PLFX_FINISH_DIRECT=$v
  rm -rf /tmp/fd_$v
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fd_$v -o b -- python bench.py --no-tight-loop --no-cpu --no-inclusion --no-svc --no-2048 --steps 10 --warmup 2 > $O/bench_$v.json 2> $O/err_$v.txt
  F=$(find /tmp/fd_$v -name "b_kernel_stats.csv" | head -1)
  echo "== PLFX_FINISH_DIRECT=$v  $(python -c "import json;d=json.loads(open('$O/bench_$v.json').read().strip().splitlines()[-1]);print('ms_per_step %.4f  value %.4e'%(d['ms_per_step'], d['value']))")"
  python - "$F" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if any(k in n for k in ('k_update_state', 'k_reduce_rows', 'k_mbox_post', 'k_gather')):
        print('   %-28s calls %4s  avg %9.2f us  min %9.2f  max %9.2f  total %9.1f us' % (n.split('(')[0].replace('void plfx::', '').replace('plfx::', ''), r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3, float(r['TotalDurationNs']) / 1e3))
PY
done
unset PLFX_FINISH_DIRECT
for v in 0 1 0 1; do
  PLFX_FINISH_DIRECT=$v python bench.py --no-cpu --no-inclusion --no-svc --no-2048 --steps 20 --warmup 5 | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('plain run PLFX_FINISH_DIRECT=$v ms_per_step %.4f'%d['ms_per_step'])"
done
timeout 600 python -m pytest tests/test_gpu_model.py -x -q -k "plastic_tension or resume or bcnode or native_load_step or calc_properties" 2>&1 | tail -2
