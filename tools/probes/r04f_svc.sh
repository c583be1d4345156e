mkdir -p gpurun_out/r04f
timeout 900 python -m pytest tests/test_gpu_material.py tests/test_gpu_configs.py -x -q -m gpu -k "svc or config4" 2>&1 | tail -4 > gpurun_out/r04f/svc_tests.txt
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu --no-inclusion --no-2048 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline_svc']
print('svc sample: %.3f s, corrector %.1f ms (%d productive launches), streaming %.1f ms, us/element %.3f' % (r['seconds'], r['kernel_ms']['k_sweep_svc_wave<1> (50-sub-step corrector)'], r['launches']['corrector_productive'], r['kernel_ms']['k_sweep_svc_wave<0> (streaming phase)'], r['roofline']['us_per_element_update']))" > gpurun_out/r04f/svc_bench.txt 2>&1
cat gpurun_out/r04f/svc_tests.txt gpurun_out/r04f/svc_bench.txt
