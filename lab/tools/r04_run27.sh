#!/bin/bash
# round 4, GPU call 27: where the plan cuts its two runs (QK_PLAN_SPLIT = share of a pair's padded work in small sites that sends it to the two-workgroup shape,
# QK_PLAN_FIT = the narrow site size) on the final kernels, cfg4
mkdir -p gpurun_out
export QK_CACHE_DIR=/tmp/qkc
O=gpurun_out/exp27.txt
: > $O
run() { echo "== $*" >> $O; timeout -k 10 800 "$@" >> $O 2>&1 || { echo "FAILED rc $?" >> $O; tail -20 $O; exit 1; }; }
run python tools/ab_plan.py cfg4 3 QK_PLAN_TILE=8 QK_PLAN_TILE=8,QK_PLAN_SPLIT=0.85 QK_PLAN_TILE=8,QK_PLAN_SPLIT=0.95 QK_PLAN_TILE=8,QK_PLAN_SPLIT=0.65 QK_PLAN_TILE=8,QK_PLAN_FIT=2304 QK_PLAN_TILE=8,QK_PLAN_FIT=4608 QK_PLAN_TILE=8,QK_PLAN_FIT=2304,QK_PLAN_SPLIT=0.9 QK_PLAN_TILE=8
grep -E "kernel " $O | cut -c1-190
