#!/bin/bash
# round 4, GPU call 19: ABLATIONS of the one-wave sweep (qk_sweep_wave2_kernel<3, double>) on cfg5 (timing only, wrong results on purpose; -DQKF_ABL=bits):
# 1 = no operand sums, 16 = no LDS-DMA, 32 = no LDS reads of the fragments, 64 = no additions behind a T tile, 113 = all of them
mkdir -p gpurun_out
export QK_CACHE_DIR=/tmp/qkc
O=gpurun_out/exp19.txt
: > $O
run() { echo "== $*" >> $O; timeout -k 10 700 "$@" >> $O 2>&1 || { echo "FAILED rc $?" >> $O; tail -20 $O; exit 1; }; }
for v in tree w1 w16 w32 w64 w113 tree; do
  if [ $v = tree ]; then unset QK_AB_LIB; else export QK_AB_LIB=lab/libqkgram_$v.so; fi
  run python tools/ab_plan.py cfg5 3 QK_PLAN_TILE=8
done
unset QK_AB_LIB
grep -E "kernel |library" $O | cut -c1-150 | sed 's/QK_PLAN_TILE=8 *//'
