#!/bin/bash
# round 4, GPU call 18: ABLATIONS of the dual kernel (timing only, wrong results on purpose; -DQKF_ABL=bits, qk_fused.h): 1 = no operand sums in the matrix
# loops, 2 = no global loads in them, 4 = no s_barrier in the step loop, 8 = no LDS reads of X in phase 1's loops; abl15p4 = all of them and no phase-2 tails
mkdir -p gpurun_out
export QK_CACHE_DIR=/tmp/qkc
O=gpurun_out/exp18.txt
: > $O
run() { echo "== $*" >> $O; timeout -k 10 500 "$@" >> $O 2>&1 || { echo "FAILED rc $?" >> $O; tail -20 $O; exit 1; }; }
for v in tree abl1 abl2 abl4 abl8 abl11 abl15p4 tree; do
  if [ $v = tree ]; then unset QK_AB_LIB; else export QK_AB_LIB=lab/libqkgram_$v.so; fi
  run python tools/ab_plan.py cfg4 3 QK_PLAN_TILE=8
done
unset QK_AB_LIB
grep -E "kernel |library" $O | cut -c1-150 | sed 's/QK_PLAN_TILE=8 *//'
