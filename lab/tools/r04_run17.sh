#!/bin/bash
# round 4, GPU call 17: TIMING PROBES of phase 2 of the dual kernel (wrong results on purpose, -DQKF_P2_PROBE=n): 1 = the LDS adds at conflict-free
# addresses, 2 = plain stores, 3 = no LDS instruction, 4 = neither the tail additions nor the LDS instructions
mkdir -p gpurun_out
export QK_CACHE_DIR=/tmp/qkc
O=gpurun_out/exp17.txt
: > $O
run() { echo "== $*" >> $O; timeout -k 10 500 "$@" >> $O 2>&1 || { echo "FAILED rc $?" >> $O; tail -20 $O; exit 1; }; }
for v in tree p2probe1 p2probe2 p2probe3 p2probe4 tree p2probe1; do
  if [ $v = tree ]; then unset QK_AB_LIB; else export QK_AB_LIB=lab/libqkgram_$v.so; fi
  run python tools/ab_plan.py cfg4 3 QK_PLAN_TILE=8
done
unset QK_AB_LIB
grep -E "kernel |library" $O | cut -c1-150 | sed 's/QK_PLAN_TILE=8 *//'
