#!/bin/bash
# GPU box: how many vector-ALU instructions do the sweep kernels issue per matrix instruction, and how busy is the vector ALU? (cfg4 by default)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_valu
mkdir -p "$OUT"
export QK_CACHE_DIR=${QK_CACHE_DIR:-/tmp/qkc}
export QK_BENCH_DEVICE_BUILD=0
cd /tmp && export TMPDIR=/tmp
python3 "$R/bench.py" $BENCH_ARGS --steps 1 --warmup 0 --cpu-seconds 0 > "$OUT/prime.json" 2> "$OUT/prime.err" || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_INT32 SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d "$OUT/a" -o pmc -- python3 "$R/bench.py" $BENCH_ARGS --steps 1 --warmup 0 --cpu-seconds 0 > "$OUT/a.json" 2> "$OUT/a.err" || exit 2
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d "$OUT/b" -o pmc -- python3 "$R/bench.py" $BENCH_ARGS --steps 1 --warmup 0 --cpu-seconds 0 > "$OUT/b.json" 2> "$OUT/b.err" || echo "pass b failed"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for tag in ("a", "b"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for f in glob.glob(f"{out}/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "qk_sweep" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k, v in acc.items():
        d = max(1, len(n[k]))
        print(tag, k[:70], {c: round(x / d) for c, x in v.items()}, "dispatches", d)
PY
