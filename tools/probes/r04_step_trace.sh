#!/bin/bash
# raw kernel trace of a short bench run (kept: it is small), for a per-load-step listing with tools/step_listing.py
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
O=gpurun_out/r04_step
mkdir -p $O
rm -rf /tmp/st
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/st -o b -- python bench.py --no-tight-loop --no-cpu --no-inclusion --no-svc --no-2048 --no-reuse-off --steps 10 --warmup 2 > $O/bench.json 2> $O/err.txt
cp $(find /tmp/st -name "b_kernel_trace.csv" | head -1) $O/kernel_trace.csv
ls -la $O
