#!/bin/bash
# round 4, GPU call 24: GANG START (QK_GANG=1: the workgroups of an XCD begin their pairs together, qk_device.h: qk_gang_sync) against free-running workgroups
mkdir -p gpurun_out
export QK_CACHE_DIR=/tmp/qkc
O=gpurun_out/exp24.txt
: > $O
run() { echo "== $*" >> $O; timeout -k 10 500 "$@" >> $O 2>&1 || { echo "FAILED rc $?" >> $O; tail -20 $O; exit 1; }; }
QK_GANG=1 timeout -k 10 400 python lab/tools/fuzz_split.py 30 > gpurun_out/fuzz_gang.log 2>&1 || { echo "fuzz FAILED"; tail -30 gpurun_out/fuzz_gang.log; exit 1; }
echo "fuzz gang: $(tail -1 gpurun_out/fuzz_gang.log | cut -c1-60)" >> $O
run python tools/ab_plan.py cfg3 5 QK_PLAN_TILE=8 QK_PLAN_TILE=8,QK_GANG=1 QK_PLAN_TILE=8 QK_PLAN_TILE=8,QK_GANG=1
run python tools/ab_plan.py cfg4 3 QK_PLAN_TILE=8 QK_PLAN_TILE=8,QK_GANG=1 QK_PLAN_TILE=8 QK_PLAN_TILE=8,QK_GANG=1
grep -E "^fuzz|kernel " $O | cut -c1-160
