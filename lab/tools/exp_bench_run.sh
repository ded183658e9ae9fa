#!/bin/bash
# GPU box: cfg4 bench of every experiment build in gpurun_exp/ (states are built once and cached)
R=${GRAFT_REPO_ROOT:-$(pwd)}; export QK_CACHE_DIR=/tmp/qkc
for lib in $R/gpurun_exp/lib_*.so; do
  n=$(basename $lib .so)
  QK_LIB=$lib timeout -k 10 300 python3 $R/lab/tools/bench_lib.py --steps 2 --warmup 1 --cpu-seconds 0 ${BENCH_ARGS} > $R/gpurun_out/eb_$n.json 2> $R/gpurun_out/eb_$n.err || { echo "$n FAILED"; tail -3 $R/gpurun_out/eb_$n.err; continue; }
  python3 -c "
import json; d=json.load(open('$R/gpurun_out/eb_$n.json')); print('%-14s ms/step %.1f kernel %.1f frac %.4f diag %.1e' % ('$n', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['diag_err']))"
done
