#!/bin/bash
# round 4, GPU call 9: the pipelined phase 2 of the dual kernel (lab/libqkgram_pipe12.so, _pipe8.so) -- parity fuzz, then cfg4 / uniform A/B against the shipped build
mkdir -p gpurun_out
export QK_CACHE_DIR=/tmp/qkc
O=gpurun_out/exp9.txt
: > $O
run() { echo "== $*" >> $O; timeout -k 10 400 "$@" >> $O 2>&1 || { echo "FAILED rc $?" >> $O; tail -20 $O; exit 1; }; }
for v in pipe12 pipe8; do
  echo "== fuzz $v" >> $O
  QK_AB_LIB=lab/libqkgram_$v.so timeout -k 10 400 python lab/tools/fuzz_split.py 30 > gpurun_out/fuzz_$v.log 2>&1 || { echo "fuzz $v FAILED" >> $O; tail -20 gpurun_out/fuzz_$v.log; exit 1; }
  tail -2 gpurun_out/fuzz_$v.log >> $O
done
for v in shipped pipe12 pipe8 w8 shipped pipe8 pipe12; do
  if [ $v = shipped ]; then unset QK_AB_LIB; else export QK_AB_LIB=lab/libqkgram_$v.so; fi
  run python tools/ab_plan.py cfg4 3 QK_PLAN_TILE=8
done
grep -E "^==|kernel |worst|library" $O | cut -c1-250
