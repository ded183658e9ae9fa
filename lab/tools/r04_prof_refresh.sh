#!/bin/bash
# round 4: the profiles of the three configs again, after the sweep sources changed (digest-gated quoting in bench.py)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for c in cfg4 cfg3 cfg5; do bash tools/prof_one.sh r04 $c > gpurun_out/prof_one_$c.log 2>&1; tail -3 gpurun_out/prof_one_$c.log | cut -c1-200; done
ls gpurun_out/sum_r04_cfg4 gpurun_out/sum_r04_cfg3 gpurun_out/sum_r04_cfg5
