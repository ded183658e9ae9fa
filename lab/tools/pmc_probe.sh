#!/bin/bash
# usage: lab/tools/pmc_probe.sh <tag> "<variant>:<wgs> ..."  -- FETCH/WRITE/TCC hit-miss of the sweep kernel per setting
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
export QK_CACHE_DIR=${QK_CACHE_DIR:-/tmp/qkc}
OUT=$R/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py $QK_BENCH_ARGS --steps 1 --warmup 0 --cpu-seconds 0 > /dev/null 2> $OUT/prime.err || { tail -3 $OUT/prime.err; exit 1; }
for vw in $1; do
  v=${vw%%:*}; w=${vw##*:}
  for ctr in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    name=$(echo $ctr | tr ' ' '_')
    QK_VARIANT=$v QK_WGS_PER_CU=$w rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/${v}_${w}_$name -o pmc -- python3 $R/bench.py $QK_BENCH_ARGS --steps 1 --warmup 0 --cpu-seconds 0 > /dev/null 2> $OUT/${v}_${w}_$name.err || echo "pass failed: $vw $ctr"
  done
  python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(float); dur = []
for f in glob.glob("$OUT/${v}_${w}_*/pmc_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "sweep" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
rd = agg.get("FETCH_SIZE", 0) * 1024 * 2; wr = agg.get("WRITE_SIZE", 0) * 1024
hit, miss = agg.get("TCC_HIT_sum", 0), agg.get("TCC_MISS_sum", 0)
print("variant $v wgs $w: read %.2f TB (FETCH_SIZE x2), write %.2f TB, kernel ms %s, L2 hit rate %.3f (hit %.3g miss %.3g)" % (rd / 1e12, wr / 1e12, ["%.0f" % d for d in dur], hit / max(1, hit + miss), hit, miss))
PY
done
