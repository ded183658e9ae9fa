#!/bin/bash
# round 4, experiments batch 1: plan tile shapes, non-temporal operand streams, 8-wave dual shape, section profile
mkdir -p gpurun_out
export QK_CACHE_DIR=/tmp/qkc
O=gpurun_out/exp1.txt
: > $O
B="QK_PLAN_TILE=8"
S32="QK_PLAN_TILE_X=32,QK_PLAN_TILE_Y=2,QK_PLAN_ORIENT_TILE=1"
run() { echo "== $*" >> $O; timeout -k 10 400 "$@" >> $O 2>&1 || { echo "FAILED rc $?" >> $O; tail -5 $O; exit 1; }; }
run python tools/ab_plan.py cfg4 3 $B $S32 QK_PLAN_TILE_X=32,QK_PLAN_TILE_Y=2 QK_PLAN_TILE_X=16,QK_PLAN_TILE_Y=4,QK_PLAN_ORIENT_TILE=1 QK_PLAN_TILE_X=2,QK_PLAN_TILE_Y=32,QK_PLAN_ORIENT_TILE=1 QK_PLAN_TILE_X=64,QK_PLAN_TILE_Y=1,QK_PLAN_ORIENT_TILE=1 QK_PLAN_TILE=8,QK_PLAN_ORIENT_TILE=1 $B
QK_AB_LIB=lab/libqkgram_ntA.so run python tools/ab_plan.py cfg4 3 $B $S32
QK_AB_LIB=lab/libqkgram_ntB.so run python tools/ab_plan.py cfg4 3 $B $S32
QK_AB_LIB=lab/libqkgram_w8.so run python tools/ab_plan.py cfg4 3 $B
QK_AB_LIB=lab/libqkgram_prof.so run python tools/ab_plan.py cfg4 1 $B
run python tools/ab_plan.py cfg3 5 $B $S32 $B
QK_AB_LIB=lab/libqkgram_ntA.so run python tools/ab_plan.py cfg3 5 $B
QK_AB_LIB=lab/libqkgram_ntB.so run python tools/ab_plan.py cfg3 5 $B
QK_AB_LIB=lab/libqkgram_prof.so run python tools/ab_plan.py cfg3 1 $B
grep -v "^states:\|^library" $O | cut -c1-230
