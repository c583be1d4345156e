mkdir -p gpurun_out/r04f
O=gpurun_out/r04f/svc_ab.txt
: > $O
run() {
  PLFX_LIB=$1 timeout 300 python tools/svc_profile.py 128 2>/dev/null | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read())
print('$2: %.3f s, corrector %.1f ms, streaming %.1f ms' % (r['seconds'], r['kernel_ms']['k_sweep_svc_wave<1> (50-sub-step corrector)'], r['kernel_ms']['k_sweep_svc_wave<0> (streaming phase)']))" >> $O
}
for rep in 1 2 3; do
  for v in "$@"; do run $(pwd)/pylabfea_amd/libplfx$v.so "lib$v"; done
done
cat $O
