set -o pipefail
cd $GRAFT_REPO_ROOT
bash profiles/run_rocprof.sh r04_cfg4 > gpurun_out/prof_r04_cfg4.log 2>&1; echo "cfg4 rc $?"
BENCH_ARGS="--config cfg3" bash profiles/run_rocprof.sh r04_cfg3 > gpurun_out/prof_r04_cfg3.log 2>&1; echo "cfg3 rc $?"
BENCH_ARGS="--config cfg5" bash profiles/run_rocprof.sh r04_cfg5 > gpurun_out/prof_r04_cfg5.log 2>&1; echo "cfg5 rc $?"
for c in cfg4 cfg3 cfg5; do python tools/summarize_prof.py gpurun_out/prof_r04_$c gpurun_out/sum_r04_$c qk_sweep "$c (bench.py --config $c), 1 GPU" > gpurun_out/sum_r04_$c.log 2>&1; echo "sum $c rc $?"; done
ls gpurun_out/sum_r04_cfg4
