#!/bin/bash
# rocprofv3 kernel-trace summaries of the supplementary workloads (which kernel ran, how long): cfg2 (wave kernel),
# cfg5 (small-bond kernel), cfg4 in fp32 (ring kernel, float).  Outputs under gpurun_out/prof_sup/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_sup; mkdir -p $OUT
export QK_CACHE_DIR=${QK_CACHE_DIR:-/tmp/qkc}
cd /tmp && export TMPDIR=/tmp
for job in "cfg2 f64" "cfg5 f64" "cfg4 f32"; do
  set -- $job
  python3 $R/bench.py --config $1 --precision $2 --steps 1 --warmup 0 --cpu-seconds 0 > /dev/null 2> $OUT/prime_$1_$2.err || { tail -3 $OUT/prime_$1_$2.err; exit 1; }
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$1_$2 -o trace -- python3 $R/bench.py --config $1 --precision $2 --steps 3 --warmup 1 --cpu-seconds 0 > $OUT/bench_$1_$2.json 2> $OUT/trace_$1_$2.err || { tail -3 $OUT/trace_$1_$2.err; exit 2; }
  head -4 $OUT/$1_$2/trace_kernel_stats.csv
done
