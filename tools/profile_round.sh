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
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python bench.py --no-tight-loop --no-cpu --no-inclusion --no-svc --no-2048 > $O/bench_under_rocprof.json 2> $O/trace.err
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o bench -- python bench.py --no-tight-loop --no-cpu --no-inclusion --no-svc --no-2048 > /dev/null 2> $O/pmc_fetch.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o bench -- python bench.py --no-tight-loop --no-cpu --no-inclusion --no-svc --no-2048 > /dev/null 2> $O/pmc_write.err
python tools/prof_summary.py $O $O/summary.txt
# the same three passes on a 2048^2 mesh: one operator pass = 470 MB > 256 MiB Infinity Cache, i.e. FETCH_SIZE is HBM traffic there
mkdir -p $O/m2048
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/m2048/trace -o bench -- python bench.py --no-tight-loop --mesh 2048 --steps 4 --warmup 1 --no-cpu --no-inclusion --no-svc > $O/m2048/bench_under_rocprof.json 2> $O/m2048/trace.err
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/m2048/pmc_fetch -o bench -- python bench.py --no-tight-loop --mesh 2048 --steps 4 --warmup 1 --no-cpu --no-inclusion --no-svc > /dev/null 2> $O/m2048/pmc_fetch.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/m2048/pmc_write -o bench -- python bench.py --no-tight-loop --mesh 2048 --steps 4 --warmup 1 --no-cpu --no-inclusion --no-svc > /dev/null 2> $O/m2048/pmc_write.err
python tools/prof_summary.py $O/m2048 $O/summary_2048.txt
cp $O/m2048/trace/bench_kernel_stats.csv $O/kernel_stats_2048.csv
rm -rf $O/m2048/trace $O/m2048/pmc_fetch $O/m2048/pmc_write
python tools/trace_gaps.py $O/trace/bench_kernel_trace.csv 18 > $O/timeline.txt 2>&1
timeout 600 python tools/configs_full.py 1 2 3 4 > $O/configs.txt 2>&1
cp $O/trace/bench_kernel_stats.csv $O/kernel_stats.csv
# keep the merge-back small: the raw traces stay on the box
rm -rf $O/trace $O/pmc_fetch $O/pmc_write
cat $O/bench.json; tail -5 $O/configs.txt
