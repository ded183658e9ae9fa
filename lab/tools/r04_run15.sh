#!/bin/bash
# round 4, GPU call 15: X in panels of 16 columns (working tree) against the leaner kernels of HEAD (lab/libqkgram_lean.so) and the round's first build (lab/libqkgram_base.so)
mkdir -p gpurun_out
export QK_CACHE_DIR=/tmp/qkc
O=gpurun_out/exp15.txt
: > $O
run() { echo "== $*" >> $O; timeout -k 10 500 "$@" >> $O 2>&1 || { echo "FAILED rc $?" >> $O; tail -20 $O; exit 1; }; }
timeout -k 10 400 python lab/tools/fuzz_split.py 40 > gpurun_out/fuzz_panel.log 2>&1 || { echo "fuzz FAILED"; tail -20 gpurun_out/fuzz_panel.log; exit 1; }
echo "fuzz: $(tail -2 gpurun_out/fuzz_panel.log | head -1 | cut -c1-60)" >> $O
timeout -k 10 400 python lab/tools/fuzz_det.py 24 > gpurun_out/fuzz_det_panel.log 2>&1 || { echo "det fuzz FAILED"; tail -20 gpurun_out/fuzz_det_panel.log; exit 1; }
echo "det fuzz: $(tail -1 gpurun_out/fuzz_det_panel.log | cut -c1-40)" >> $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/suite.log 2>&1; rc=$?; tail -3 gpurun_out/suite.log >> $O
[ $rc -eq 0 ] || { tail -30 gpurun_out/suite.log; exit $rc; }
for v in base lean tree base lean tree; do
  if [ $v = tree ]; then unset QK_AB_LIB; else export QK_AB_LIB=lab/libqkgram_$v.so; fi
  run python tools/ab_plan.py cfg4 3 QK_PLAN_TILE=8 QK_DETERMINISTIC=1
  run python tools/ab_plan.py cfg3 5 QK_PLAN_TILE=8
done
unset QK_AB_LIB
grep -E "^fuzz|^det|passed|failed|kernel |library" $O | cut -c1-120 | sed 's/QK_PLAN_TILE=8 *//'
