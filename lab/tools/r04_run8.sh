#!/bin/bash
# round 4, GPU call 8: the right-edge product cut into K chunks, against the build before (same box)
mkdir -p gpurun_out
export QK_CACHE_DIR=/tmp/qkc
O=gpurun_out/exp6.txt
: > $O
run() { echo "== $*" >> $O; timeout -k 10 500 "$@" >> $O 2>&1 || { echo "FAILED rc $?" >> $O; tail -5 $O; exit 1; }; }
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/suite_part.log 2>&1; echo "rc $?" >> gpurun_out/suite_part.log; tail -3 gpurun_out/suite_part.log
run python tools/ab_plan.py cfg3 8 QK_PLAN_TILE=8
QK_AB_LIB=lab/libqkgram_prev.so run python tools/ab_plan.py cfg3 8 QK_PLAN_TILE=8
run python tools/ab_plan.py cfg3 8 QK_PLAN_TILE=8
QK_AB_LIB=lab/libqkgram_prev.so run python tools/ab_plan.py cfg3 8 QK_PLAN_TILE=8
run python tools/ab_plan.py cfg4 4 QK_PLAN_TILE=8
QK_AB_LIB=lab/libqkgram_prev.so run python tools/ab_plan.py cfg4 4 QK_PLAN_TILE=8
run python tools/ab_plan.py cfg4 4 QK_PLAN_TILE=8
grep -E "^==|kernel " $O | cut -c1-200
