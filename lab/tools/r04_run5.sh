#!/bin/bash
# round 4, GPU call 5: small-site shapes (3 workgroups per CU, 4 waves x 2 slots), the complex64 test on real states, the fp32 tolerance table
mkdir -p gpurun_out
export QK_CACHE_DIR=/tmp/qkc
O=gpurun_out/exp4.txt
: > $O
run() { echo "== $*" >> $O; timeout -k 10 500 "$@" >> $O 2>&1 || { echo "FAILED rc $?" >> $O; tail -5 $O; exit 1; }; }
run python tools/ab_plan.py cfg3 5 QK_PLAN_TILE=8
for v in t3a t3b t2w4; do QK_AB_LIB=lab/libqkgram_$v.so run python tools/ab_plan.py cfg3 5 QK_PLAN_TILE=8; done
run python tools/ab_plan.py cfg4 3 QK_PLAN_TILE=8
for v in t3a t3b t2w4; do QK_AB_LIB=lab/libqkgram_$v.so run python tools/ab_plan.py cfg4 3 QK_PLAN_TILE=8; done
grep -E "^==|kernel " $O | cut -c1-250
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "complex64 or deterministic or communicator" > gpurun_out/suite_part.log 2>&1; echo "rc $?" >> gpurun_out/suite_part.log; tail -4 gpurun_out/suite_part.log
timeout -k 10 600 python tools/fp32_sweep.py > gpurun_out/fp32_tolerance.txt 2>&1 || tail -5 gpurun_out/fp32_tolerance.txt
tail -6 gpurun_out/fp32_tolerance.txt
