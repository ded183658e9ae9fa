#!/bin/bash
# round 4, GPU call 25: edge products primed across the tiles of a wave (working tree) against every tile starting cold (lab/libqkgram_noprime.so, -DQKF_EDGE_PRIME=0)
mkdir -p gpurun_out
export QK_CACHE_DIR=/tmp/qkc
O=gpurun_out/exp25.txt
: > $O
run() { echo "== $*" >> $O; timeout -k 10 500 "$@" >> $O 2>&1 || { echo "FAILED rc $?" >> $O; tail -20 $O; exit 1; }; }
timeout -k 10 400 python lab/tools/fuzz_split.py 40 > gpurun_out/fuzz_prime.log 2>&1 || { echo "fuzz FAILED"; tail -30 gpurun_out/fuzz_prime.log; exit 1; }
echo "fuzz: $(tail -1 gpurun_out/fuzz_prime.log | cut -c1-60)" >> $O
timeout -k 10 400 python lab/tools/fuzz_det.py 16 > gpurun_out/fuzz_det_prime.log 2>&1 || { echo "det fuzz FAILED"; tail -20 gpurun_out/fuzz_det_prime.log; exit 1; }
echo "det fuzz: $(tail -1 gpurun_out/fuzz_det_prime.log | cut -c1-60)" >> $O
for v in noprime tree noprime tree; do
  if [ $v = tree ]; then unset QK_AB_LIB; else export QK_AB_LIB=lab/libqkgram_$v.so; fi
  run python tools/ab_plan.py cfg4 3 QK_PLAN_TILE=8 QK_DETERMINISTIC=1
  run python tools/ab_plan.py cfg3 5 QK_PLAN_TILE=8
done
unset QK_AB_LIB
grep -E "^fuzz|^det|kernel |library" $O | cut -c1-150 | sed 's/QK_PLAN_TILE=8 *//'
