#!/bin/bash
# Profile set of one build, as committed under profiles/ (run on the GPU box from the repo root):
#   tools/profile_round.sh r01i
# writes gpurun_out/<tag>/: kernel trace + stats, FETCH_SIZE / WRITE_SIZE PMC passes (separate runs, counters only with
# --kernel-trace-less passes as the pool requires), the plain bench line, the timeline view and the whole-solve configs.
set -u
TAG=${1:-prof}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd $ROOT
python bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python bench.py --no-cpu --no-inclusion --no-svc > $O/bench_under_rocprof.json 2> $O/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o bench -- python bench.py --no-cpu --no-inclusion --no-svc --steps 3 --warmup 1 > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o bench -- python bench.py --no-cpu --no-inclusion --no-svc --steps 3 --warmup 1 > /dev/null 2> $O/pmc_write.err
python tools/prof_summary.py $O $O/summary.txt
python tools/trace_gaps.py $O/trace/bench_kernel_trace.csv 18 > $O/timeline.txt 2>&1
python tools/configs_full.py 1 2 3 4 > $O/configs.txt 2>&1
cp $O/trace/bench_kernel_stats.csv $O/kernel_stats.csv
# keep the merge-back small: the raw traces stay on the box
rm -rf $O/trace $O/pmc_fetch $O/pmc_write
cat $O/bench.json; tail -5 $O/configs.txt
