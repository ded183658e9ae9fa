#!/bin/bash
# usage: lab/tools/pmc_stalls.sh <tag>   -- memory-pipeline stall/latency counters of the shipped sweep kernel (cfg4)
TAG=${1:-a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export QK_CACHE_DIR=${QK_CACHE_DIR:-/tmp/qkc}
OUT=$R/gpurun_out/pmcst_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 1 --warmup 0 --cpu-seconds 0 > /dev/null 2> $OUT/prime.err || { tail -3 $OUT/prime.err; exit 1; }
i=0
for grp in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum" "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_BUSY_CYCLES" "TCC_TAG_STALL_sum TCC_BUSY_avr" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -o pmc -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-seconds 0 > /dev/null 2> $OUT/p$i.err || echo "pass $i failed: $grp"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(float)
for f in sorted(glob.glob("$OUT/p*/pmc_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "sweep" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(agg): print("%-44s %.6g" % (k, agg[k]))
def ratio(a, b, name):
    if agg.get(b): print("%-44s %.4g" % (name, agg[a] / agg[b]))
ratio("TCP_TCC_READ_REQ_LATENCY_sum", "TCP_TCC_READ_REQ_sum", "avg L1->L2 read latency (cycles)")
ratio("TCP_TCC_WRITE_REQ_LATENCY_sum", "TCP_TCC_WRITE_REQ_sum", "avg L1->L2 write latency (cycles)")
ratio("TCP_UTCL1_TRANSLATION_MISS_sum", "TCP_UTCL1_REQUEST_sum", "UTCL1 (TLB) miss rate")
PY
