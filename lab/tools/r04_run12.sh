#!/bin/bash
# round 4, GPU call 12: variants of the leaner dual kernel against the build of HEAD, cfg4, alternating
mkdir -p gpurun_out
export QK_CACHE_DIR=/tmp/qkc
O=gpurun_out/exp12.txt
: > $O
run() { echo "== $*" >> $O; timeout -k 10 500 "$@" >> $O 2>&1 || { echo "FAILED rc $?" >> $O; tail -20 $O; exit 1; }; }
for v in gauss peel all8; do
  QK_AB_LIB=lab/libqkgram_$v.so timeout -k 10 300 python lab/tools/fuzz_split.py 20 > gpurun_out/fuzz_$v.log 2>&1 || { echo "fuzz $v FAILED"; tail -20 gpurun_out/fuzz_$v.log; exit 1; }
  echo "fuzz $v: $(tail -1 gpurun_out/fuzz_$v.log | cut -c1-40)" >> $O
done
for v in base tree gauss peel lean8 all8 base tree gauss peel lean8 all8; do
  if [ $v = tree ]; then unset QK_AB_LIB; else export QK_AB_LIB=lab/libqkgram_$v.so; fi
  run python tools/ab_plan.py cfg4 3 QK_PLAN_TILE=8
done
unset QK_AB_LIB
grep -E "^fuzz|kernel |library" $O | cut -c1-110 | sed 's/QK_PLAN_TILE=8 *//'
