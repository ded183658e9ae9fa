#!/bin/bash
# one config through the profiling recipe and the summary:  bash tools/prof_one.sh r04 cfg4 [bench args]
set -o pipefail
cd $GRAFT_REPO_ROOT
RND=$1; CFG=$2; shift 2
BENCH_ARGS="--config $CFG $*" bash profiles/run_rocprof.sh ${RND}_$CFG > gpurun_out/prof_${RND}_$CFG.log 2>&1; echo "$CFG rc $?"
python tools/summarize_prof.py gpurun_out/prof_${RND}_$CFG gpurun_out/sum_${RND}_$CFG qk_sweep "$CFG (bench.py --config $CFG $*), 1 GPU" > gpurun_out/sum_${RND}_$CFG.log 2>&1; echo "sum $CFG rc $?"
cp gpurun_out/prof_${RND}_$CFG/trace_bench.json gpurun_out/sum_${RND}_$CFG/bench_during_trace.json 2>/dev/null
tail -40 gpurun_out/sum_${RND}_$CFG.log
