mkdir -p gpurun_out/r04b
O=gpurun_out/r04b/vcycle_ab2.txt
: > $O
for rep in 1 2; do
for cfg in 4 16; do
  PLFX_MG_COARSEST_ELEMS=$cfg timeout 300 python tools/probes/vcycle_ab.py 1024 300 2>&1 | tail -1 >> $O
done
done
PLFX_MG_COARSEST_ELEMS=4 timeout 300 python tools/probes/vcycle_ab.py 2048 100 2>&1 | tail -1 >> $O
PLFX_MG_COARSEST_ELEMS=16 timeout 300 python tools/probes/vcycle_ab.py 2048 100 2>&1 | tail -1 >> $O
cat $O
timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_sharded.py -x -q -m gpu 2>&1 | tail -5
