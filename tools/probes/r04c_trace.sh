# per-launch view of one V-cycle (rocprofv3 kernel trace of the tight loop, eager launches): tools/vcycle_trace.py
mkdir -p gpurun_out/r04c; export TMPDIR=/tmp
O=$(pwd)/gpurun_out/r04c
PLFX_MG_GRAPH=0 timeout 240 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o vc -- python tools/probes/vcycle_ab.py 1024 60 > $O/run.txt 2> $O/err.txt
F=$(find $O/trace -name "*kernel_trace.csv" 2>/dev/null | head -1)
if [ -n "$F" ]; then timeout 120 python tools/vcycle_trace.py "$F" 50 > $O/vcycle_launches.txt 2>&1; fi
rm -rf $O/trace
timeout 200 python tools/probes/vcycle_ab.py 1024 300 >> $O/run.txt 2>&1
tail -3 $O/run.txt; cat $O/vcycle_launches.txt
