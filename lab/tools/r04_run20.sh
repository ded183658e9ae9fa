#!/bin/bash
# round 4, GPU call 20: the QUAD form of the site-fused sweep (qk_quad.h; lab/libqkgram_quad.so = -DQKF_QUAD=1: it stands in for the plain dual form) -- parity
# fuzz against the oracle, then cfg4 / cfg3 against the shipped build
mkdir -p gpurun_out
export QK_CACHE_DIR=/tmp/qkc
O=gpurun_out/exp20.txt
: > $O
run() { echo "== $*" >> $O; timeout -k 10 500 "$@" >> $O 2>&1 || { echo "FAILED rc $?" >> $O; tail -20 $O; exit 1; }; }
QK_AB_LIB=lab/libqkgram_quad.so timeout -k 10 400 python lab/tools/fuzz_split.py 40 > gpurun_out/fuzz_quad.log 2>&1 || { echo "fuzz FAILED"; tail -30 gpurun_out/fuzz_quad.log; exit 1; }
echo "fuzz: $(tail -2 gpurun_out/fuzz_quad.log | head -1 | cut -c1-200)" >> $O
echo "fuzz: $(tail -1 gpurun_out/fuzz_quad.log | cut -c1-60)" >> $O
for v in tree quad tree quad; do
  if [ $v = tree ]; then unset QK_AB_LIB; else export QK_AB_LIB=lab/libqkgram_$v.so; fi
  run python tools/ab_plan.py cfg4 3 QK_PLAN_TILE=8
done
unset QK_AB_LIB
grep -E "^fuzz|kernel |library" $O | cut -c1-220 | sed 's/QK_PLAN_TILE=8 *//'
