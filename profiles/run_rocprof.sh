#!/bin/bash
# Profiling recipe used for the summaries in this directory (run on the GPU box via gpurun):
#   bash profiles/run_rocprof.sh r04_cfg4                 (headline workload, cfg4)
#   BENCH_ARGS="--config cfg5" bash profiles/run_rocprof.sh r04_cfg5
# then: python tools/summarize_prof.py gpurun_out/prof_<tag> profiles/r04/<config>       (tools/prof_all.sh does all of it)
# 1) plain run fills the MPS cache so that nothing forks under the profiler
# 2) kernel trace + stats   3) PMC passes (separate runs, kernel-trace only)
set -o pipefail
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export QK_CACHE_DIR=${QK_CACHE_DIR:-/tmp/qkc}
export QK_BENCH_DEVICE_BUILD=0   # the profiled runs time the Gram sweep; the device builder has its own numbers
cd /tmp && export TMPDIR=/tmp
python3 "$R/bench.py" $BENCH_ARGS --steps 1 --warmup 0 --cpu-seconds 0 > "$OUT/prime.json" 2> "$OUT/prime.err" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 "$R/bench.py" $BENCH_ARGS --steps 3 --warmup 1 --cpu-seconds 0 > "$OUT/trace_bench.json" 2> "$OUT/trace.err" || exit 2
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o pmc -- python3 "$R/bench.py" $BENCH_ARGS --steps 1 --warmup 0 --cpu-seconds 0 > "$OUT/pmc_fetch.json" 2> "$OUT/pmc_fetch.err" || exit 3
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o pmc -- python3 "$R/bench.py" $BENCH_ARGS --steps 1 --warmup 0 --cpu-seconds 0 > "$OUT/pmc_write.json" 2> "$OUT/pmc_write.err" || exit 4
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq" -o pmc -- python3 "$R/bench.py" $BENCH_ARGS --steps 1 --warmup 0 --cpu-seconds 0 > "$OUT/pmc_sq.json" 2> "$OUT/pmc_sq.err" || echo "sq pass failed (non-fatal)"
# 3b) LDS pass (are the ds_add_f64 of phase 2 what the waves wait for?) and L2 hit rate (three TCC counters per pass: more hang the collector)
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d "$OUT/pmc_lds" -o pmc -- python3 "$R/bench.py" $BENCH_ARGS --steps 1 --warmup 0 --cpu-seconds 0 > "$OUT/pmc_lds.json" 2> "$OUT/pmc_lds.err" || echo "lds pass failed (non-fatal)"
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d "$OUT/pmc_tcc" -o pmc -- python3 "$R/bench.py" $BENCH_ARGS --steps 1 --warmup 0 --cpu-seconds 0 > "$OUT/pmc_tcc.json" 2> "$OUT/pmc_tcc.err" || echo "tcc pass failed (non-fatal)"
# 4) roctx ranges (qk:build / qk:upload / qk:sweep / qk:scatter / bench:step): the phase table of one run
rocprofv3 --kernel-trace --marker-trace --output-format csv -d "$OUT/marker" -o marker -- python3 "$R/bench.py" $BENCH_ARGS --steps 3 --warmup 1 --cpu-seconds 0 > "$OUT/marker_bench.json" 2> "$OUT/marker.err" || echo "marker pass failed (non-fatal)"
rocprofv3 -L > "$OUT/counters_list.txt" 2>&1 || true
find "$OUT" -name "*.csv" | head -50
