#!/bin/bash
# round 4, GPU call 3: parity suite with the DET kernels, DET against default, share tails
mkdir -p gpurun_out
export QK_CACHE_DIR=/tmp/qkc
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/suite.log 2>&1; rc=$?; echo "rc $rc" >> gpurun_out/suite.log
tail -15 gpurun_out/suite.log
[ $rc -eq 0 ] || exit $rc
O=gpurun_out/exp2.txt
: > $O
run() { echo "== $*" >> $O; timeout -k 10 500 "$@" >> $O 2>&1 || { echo "FAILED rc $?" >> $O; tail -5 $O; exit 1; }; }
run python tools/ab_plan.py cfg4 3 QK_PLAN_TILE=8 QK_DETERMINISTIC=1 QK_PLAN_TILE=8 QK_DETERMINISTIC=1
run python tools/ab_plan.py cfg3 5 QK_PLAN_TILE=8 QK_DETERMINISTIC=1 QK_PLAN_TILE=8 QK_DETERMINISTIC=1
run python tools/share_times.py cfg4 3 8
run python tools/share_times.py cfg4 3 8 QK_FUSED_SPLIT=0
grep -E "^==|kernel |world|rank " $O | cut -c1-250
