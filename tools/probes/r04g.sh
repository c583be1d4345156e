mkdir -p gpurun_out/r04g
timeout 1700 python -m pytest tests/test_gpu_sharded.py -x -q -m gpu -k "bench_two_ranks" --durations=5 2>&1 | tail -15 > gpurun_out/r04g/bench2.txt
cat gpurun_out/r04g/bench2.txt
