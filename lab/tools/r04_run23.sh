#!/bin/bash
# round 4, GPU call 23: ABLATIONS of the one-tile kernel (qk_sweep_fused_kernel<8, 1, 4608, 4>) on cfg3 (timing only, wrong results on purpose; -DQKF_ABL=bits):
# 1 = no operand sums, 2 = no global loads in the loops, 4 = no s_barrier in the step loop, 8 = no LDS reads of X in phase 1's loops, 128 = no additions behind a
# block of phase 2, 384 = neither those nor the LDS adds, 399 = all of them
mkdir -p gpurun_out
export QK_CACHE_DIR=/tmp/qkc
O=gpurun_out/exp23.txt
: > $O
run() { echo "== $*" >> $O; timeout -k 10 500 "$@" >> $O 2>&1 || { echo "FAILED rc $?" >> $O; tail -20 $O; exit 1; }; }
for v in tree s1 s2 s4 s8 s128 s384 s399 tree; do
  if [ $v = tree ]; then unset QK_AB_LIB; else export QK_AB_LIB=lab/libqkgram_$v.so; fi
  run python tools/ab_plan.py cfg3 5 QK_PLAN_TILE=8
done
unset QK_AB_LIB
grep -E "kernel |library" $O | cut -c1-150 | sed 's/QK_PLAN_TILE=8 *//'
