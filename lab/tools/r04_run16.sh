#!/bin/bash
# round 4, GPU call 16: phase 2 of the dual kernel with the 3M product's last additions left to the LDS (lab/libqkgram_lds3m.so, -DQKF_LDS3M=1)
# against the shipped build
mkdir -p gpurun_out
export QK_CACHE_DIR=/tmp/qkc
O=gpurun_out/exp16.txt
: > $O
run() { echo "== $*" >> $O; timeout -k 10 500 "$@" >> $O 2>&1 || { echo "FAILED rc $?" >> $O; tail -20 $O; exit 1; }; }
QK_AB_LIB=lab/libqkgram_lds3m.so timeout -k 10 400 python lab/tools/fuzz_split.py 40 > gpurun_out/fuzz_lds3m.log 2>&1 || { echo "fuzz FAILED"; tail -20 gpurun_out/fuzz_lds3m.log; exit 1; }
echo "fuzz: $(tail -2 gpurun_out/fuzz_lds3m.log | head -1 | cut -c1-60)" >> $O
for v in tree lds3m tree lds3m; do
  if [ $v = tree ]; then unset QK_AB_LIB; else export QK_AB_LIB=lab/libqkgram_$v.so; fi
  run python tools/ab_plan.py cfg4 3 QK_PLAN_TILE=8
  run python tools/ab_plan.py cfg3 5 QK_PLAN_TILE=8
done
unset QK_AB_LIB
grep -E "^fuzz|^det|passed|failed|kernel |library" $O | cut -c1-150 | sed 's/QK_PLAN_TILE=8 *//'
