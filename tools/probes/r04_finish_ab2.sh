#!/bin/bash
# wall-clock A/B (no profiler): alternating runs of the bench window, PLFX_FINISH_DIRECT = 0 / 1 and the library before the two changes
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2 3 4 5 6; do
  for v in old 0 1; do
    if [ $v = old ]; then export PLFX_LIB=$PWD/build/libplfx_old1024.so; unset PLFX_FINISH_DIRECT; else unset PLFX_LIB; export PLFX_FINISH_DIRECT=$v; fi
    python bench.py --no-cpu --no-inclusion --no-svc --no-2048 --no-tight-loop --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$v %.4f'%d['ms_per_step'])"
  done
done | sort | awk '{a[$1]=a[$1]" "$2; s[$1]+=$2; n[$1]++; if(!($1 in m)||$2<m[$1])m[$1]=$2} END{for(k in a) printf "%-4s mean %.4f min %.4f :%s\n", k, s[k]/n[k], m[k], a[k]}'
