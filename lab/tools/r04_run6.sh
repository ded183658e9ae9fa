#!/bin/bash
# round 4, GPU call 6: the 8-wave dual shape on uniform chains and on the capped cfg5 set; LDS / L2 counters of cfg4
mkdir -p gpurun_out
export QK_CACHE_DIR=/tmp/qkc
O=gpurun_out/exp5.txt
: > $O
run() { echo "== $*" >> $O; timeout -k 10 600 "$@" >> $O 2>&1 || { echo "FAILED rc $?" >> $O; tail -5 $O; exit 1; }; }
run python tools/uniform_ab.py 64,96,128,192 QK_FUSED_WGS=1
QK_AB_LIB=lab/libqkgram_w8.so run python tools/uniform_ab.py 64,96,128,192 QK_FUSED_WGS=1
grep -E "^==|cap |ms" $O | cut -c1-200
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_lds
mkdir -p $OUT
export QK_BENCH_DEVICE_BUILD=0
python3 $R/bench.py --steps 1 --warmup 0 --cpu-seconds 0 > $OUT/prime.json 2> $OUT/prime.err || exit 1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_lds -o pmc -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-seconds 0 > $OUT/pmc_lds.json 2> $OUT/pmc_lds.err || echo "lds pass failed"
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/pmc_tcc -o pmc -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-seconds 0 > $OUT/pmc_tcc.json 2> $OUT/pmc_tcc.err || echo "tcc pass failed"
cd $R
python - <<'PY'
import csv,glob,collections
for tag in ("lds","tcc"):
    for f in glob.glob(f"gpurun_out/prof_lds/pmc_{tag}/**/*counter_collection.csv", recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            if "qk_sweep" in r["Kernel_Name"]:
                k=r["Kernel_Name"][:60]; acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
        for k,v in acc.items():
            print(tag,k,len(n[k]),{c:"%.4g"%(x/len(n[k])) for c,x in v.items()})
PY
