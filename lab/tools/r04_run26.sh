#!/bin/bash
# round 4, GPU call 26: one-wave sweep with the two tiles (p = 0 | 1) of a block walked together (working tree) against one after the other (lab/libqkgram_nopair.so, -DQKW_PAIR=0)
mkdir -p gpurun_out
export QK_CACHE_DIR=/tmp/qkc
O=gpurun_out/exp26.txt
: > $O
run() { echo "== $*" >> $O; timeout -k 10 700 "$@" >> $O 2>&1 || { echo "FAILED rc $?" >> $O; tail -20 $O; exit 1; }; }
timeout -k 10 400 python lab/tools/fuzz_split.py 40 > gpurun_out/fuzz_pair.log 2>&1 || { echo "fuzz FAILED"; tail -30 gpurun_out/fuzz_pair.log; exit 1; }
echo "fuzz: $(tail -1 gpurun_out/fuzz_pair.log | cut -c1-60)" >> $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "small_bond or cfg5 or complex64 or wave or mixed or randomised or f32" > gpurun_out/suite_pair.log 2>&1; rc=$?; tail -3 gpurun_out/suite_pair.log >> $O
[ $rc -eq 0 ] || { tail -30 gpurun_out/suite_pair.log; exit $rc; }
for v in nopair tree nopair tree; do
  if [ $v = tree ]; then unset QK_AB_LIB; else export QK_AB_LIB=lab/libqkgram_$v.so; fi
  run python tools/ab_plan.py cfg5 3 QK_PLAN_TILE=8
done
unset QK_AB_LIB
grep -E "^fuzz|passed|failed|kernel |library" $O | cut -c1-150 | sed 's/QK_PLAN_TILE=8 *//'
