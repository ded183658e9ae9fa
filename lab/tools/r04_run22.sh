#!/bin/bash
# round 4, GPU call 22: the quad kernel with its pieces dealt column pair by column pair (quad; quad0 = every quad as two column pairs) -- fuzz, then cfg4 and cfg3
mkdir -p gpurun_out
export QK_CACHE_DIR=/tmp/qkc
O=gpurun_out/exp22.txt
: > $O
run() { echo "== $*" >> $O; timeout -k 10 500 "$@" >> $O 2>&1 || { echo "FAILED rc $?" >> $O; tail -20 $O; exit 1; }; }
for v in quad quad0; do
QK_AB_LIB=lab/libqkgram_$v.so timeout -k 10 400 python lab/tools/fuzz_split.py 40 > gpurun_out/fuzz_$v.log 2>&1 || { echo "fuzz $v FAILED"; tail -30 gpurun_out/fuzz_$v.log; exit 1; }
echo "fuzz $v: $(tail -1 gpurun_out/fuzz_$v.log | cut -c1-60)" >> $O
done
for v in tree quad quad0 tree quad; do
  if [ $v = tree ]; then unset QK_AB_LIB; else export QK_AB_LIB=lab/libqkgram_$v.so; fi
  run python tools/ab_plan.py cfg4 3 QK_PLAN_TILE=8
done
unset QK_AB_LIB
grep -E "^fuzz|kernel |library" $O | cut -c1-220 | sed 's/QK_PLAN_TILE=8 *//'
