#!/bin/bash
# round 4, GPU call 21: the quad kernel with no whole quads (quad0: an 8-wave dual form on the quad kernel's skeleton) and with quads only from two full rounds on (quad2)
mkdir -p gpurun_out
export QK_CACHE_DIR=/tmp/qkc
O=gpurun_out/exp21.txt
: > $O
run() { echo "== $*" >> $O; timeout -k 10 500 "$@" >> $O 2>&1 || { echo "FAILED rc $?" >> $O; tail -20 $O; exit 1; }; }
for v in tree quad quad0 dual8 tree; do
  if [ $v = tree ]; then unset QK_AB_LIB; else export QK_AB_LIB=lab/libqkgram_$v.so; fi
  run python tools/ab_plan.py cfg4 3 QK_PLAN_TILE=8
done
unset QK_AB_LIB
grep -E "^fuzz|kernel |library" $O | cut -c1-220 | sed 's/QK_PLAN_TILE=8 *//'
