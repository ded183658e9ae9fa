#!/bin/bash
# round 4, GPU call 11: the sweep kernels with fewer vector instructions (working tree) against the build of HEAD (lab/libqkgram_base.so): fuzz, A/B, instruction counts
mkdir -p gpurun_out
export QK_CACHE_DIR=/tmp/qkc
O=gpurun_out/exp11.txt
: > $O
run() { echo "== $*" >> $O; timeout -k 10 500 "$@" >> $O 2>&1 || { echo "FAILED rc $?" >> $O; tail -20 $O; exit 1; }; }
timeout -k 10 400 python lab/tools/fuzz_split.py 30 > gpurun_out/fuzz_lean.log 2>&1 || { echo "fuzz FAILED"; tail -20 gpurun_out/fuzz_lean.log; exit 1; }
tail -2 gpurun_out/fuzz_lean.log >> $O
timeout -k 10 400 python lab/tools/fuzz_det.py 16 > gpurun_out/fuzz_det_lean.log 2>&1 || { echo "det fuzz FAILED"; tail -20 gpurun_out/fuzz_det_lean.log; exit 1; }
tail -1 gpurun_out/fuzz_det_lean.log | cut -c1-60 >> $O
for v in base tree base tree; do
  if [ $v = tree ]; then unset QK_AB_LIB; else export QK_AB_LIB=lab/libqkgram_$v.so; fi
  run python tools/ab_plan.py cfg4 3 QK_PLAN_TILE=8 QK_DETERMINISTIC=1
  run python tools/ab_plan.py cfg3 5 QK_PLAN_TILE=8 QK_DETERMINISTIC=1
done
unset QK_AB_LIB
grep -E "^==|kernel |worst|library" $O | cut -c1-250
bash lab/tools/pmc_valu.sh > gpurun_out/pmc_valu.log 2>&1; tail -4 gpurun_out/pmc_valu.log | cut -c1-420
