#!/bin/bash
mkdir -p gpurun_out
export QK_CACHE_DIR=/tmp/qkc
O=gpurun_out/exp14.txt
: > $O
run() { echo "== $*" >> $O; timeout -k 10 500 "$@" >> $O 2>&1 || { echo "FAILED rc $?" >> $O; tail -20 $O; exit 1; }; }
QK_AB_LIB=lab/libqkgram_gp1.so timeout -k 10 300 python lab/tools/fuzz_split.py 20 > gpurun_out/fuzz_gp1.log 2>&1 || { echo "fuzz gp1 FAILED"; tail -20 gpurun_out/fuzz_gp1.log; exit 1; }
echo "fuzz gp1: $(tail -1 gpurun_out/fuzz_gp1.log | cut -c1-40)" >> $O
for v in base gp gp1 base gp gp1; do
  export QK_AB_LIB=lab/libqkgram_$v.so
  run python tools/ab_plan.py cfg4 3 QK_PLAN_TILE=8
  run python tools/ab_plan.py cfg3 5 QK_PLAN_TILE=8
done
unset QK_AB_LIB
grep -E "^fuzz|kernel |library" $O | cut -c1-110 | sed 's/QK_PLAN_TILE=8 *//'
