#!/bin/bash
# usage: [QK_LIB=...] lab/tools/pmc_bench.sh <tag> [bench args]  -- SQ / LDS / TCP counters of the sweep kernel on the bench workload (GPU box)
TAG=${1:-a}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
export QK_CACHE_DIR=${QK_CACHE_DIR:-/tmp/qkc}
OUT=$R/gpurun_out/pmcb_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/lab/tools/bench_lib.py --steps 1 --warmup 0 --cpu-seconds 0 "$@" > /dev/null 2> $OUT/prime.err || { tail -3 $OUT/prime.err; exit 1; }
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -o pmc -- python3 $R/lab/tools/bench_lib.py --steps 1 --warmup 0 --cpu-seconds 0 "$@" > $OUT/p$i.out 2> $OUT/p$i.err || { echo "pass $i failed: $grp"; tail -3 $OUT/p$i.err; }
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(float); n = collections.defaultdict(int)
for f in sorted(glob.glob("$OUT/p*/pmc_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "sweep" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(agg): print("%-36s %.6g  (per launch, %d launches)" % (k, agg[k] / n[k], n[k]))
PY
