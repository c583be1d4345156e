#!/bin/bash
# rocprofv3 passes over the bounded config-4 (SVC) sample: tools/svc_profile_round.sh <tag> -> gpurun_out/<tag>/svc_summary.txt
# (kernel trace; SQ issue counters; the FP64 operation counters behind the true flop rate of roofline_svc)
set -u
TAG=${1:-svc}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o svc -- python tools/svc_profile.py 256 > $O/svc_sample.json 2> $O/trace.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc1 -o svc -- python tools/svc_profile.py 256 > /dev/null 2> $O/pmc1.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 --output-format csv -d $O/pmc2 -o svc -- python tools/svc_profile.py 256 > /dev/null 2> $O/pmc2.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY --output-format csv -d $O/pmc3 -o svc -- python tools/svc_profile.py 256 > /dev/null 2> $O/pmc3.err
python tools/svc_prof_summary.py $O 65536 $O/svc_summary.txt
rm -rf $O/trace $O/pmc1 $O/pmc2 $O/pmc3
