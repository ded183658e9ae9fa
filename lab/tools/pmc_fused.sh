#!/bin/bash
# usage: lab/tools/pmc_fused.sh <tag> <chi> [n_states]  -- SQ / LDS counters of the sweep kernel on a uniform-bond set (GPU box)
TAG=${1:-a}; CHI=${2:-64}; NS=${3:-181}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmcf_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export QK_CHIS=$CHI
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_FLAT SQ_INSTS_SMEM" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -o pmc -- python3 $R/lab/tools/chi_scan.py 60 $NS > $OUT/p$i.out 2> $OUT/p$i.err || { echo "pass $i failed: $grp"; tail -3 $OUT/p$i.err; }
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
