#!/bin/bash
# heterogeneous (soft inclusion) variant of the bench under smoother variants of the V-cycle: ms per load step, PCG iterations
O=gpurun_out/${1:-knobs}
mkdir -p $O
for cfg in "2 0.65" "1 0.65" "1 0.8" "3 0.65"; do
  set -- $cfg
  MG_NU=$1 MG_OMEGA=$2 python bench.py --no-cpu --no-svc --steps 5 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); i=d['inclusion_variant']
print('nu=$1 omega=$2: homogeneous %.3f ms/step (%d its) | inclusion %.1f ms/step, %d its in %d solves (%.1f per computed solve)' % (d['ms_per_step'], d['pcg_iterations'], i['ms_per_step'], i['pcg_iterations'], i['solves'], i['pcg_iterations_per_computed_solve']))" | tee -a $O/knobs.txt
done
