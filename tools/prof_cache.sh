#!/bin/bash
# L1 / L2 hit rates of the sweep kernels (two PMC passes, kernel trace only):  bash tools/prof_cache.sh <tag> [bench args]
#   pass 1: requests into the L1s and from the L1s to the L2s;  pass 2: L2 hits, misses, reads to the fabric
#   (at most three L2 counters in a pass: with six, rocprofiler_create_counter_config fails and the run hangs)
set -o pipefail
TAG=${1:-cache}
shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/cache_$TAG
mkdir -p "$OUT"
export QK_CACHE_DIR=${QK_CACHE_DIR:-/tmp/qkc}
export QK_BENCH_DEVICE_BUILD=0
cd /tmp && export TMPDIR=/tmp
python3 "$R/bench.py" "$@" --steps 1 --warmup 0 --cpu-seconds 0 > "$OUT/prime.json" 2> "$OUT/prime.err" || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TCC_READ_REQ_sum TA_FLAT_READ_WAVEFRONTS_sum --output-format csv -d "$OUT/l1" -o pmc -- python3 "$R/bench.py" "$@" --steps 1 --warmup 0 --cpu-seconds 0 > "$OUT/l1.json" 2> "$OUT/l1.err" || exit 2
timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d "$OUT/l2" -o pmc -- python3 "$R/bench.py" "$@" --steps 1 --warmup 0 --cpu-seconds 0 > "$OUT/l2.json" 2> "$OUT/l2.err" || exit 3
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for p in ("l1", "l2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for f in glob.glob(f"{out}/{p}/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "qk_sweep" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            seen.add((k, r["Dispatch_Id"]))
        for k, _ in seen: cnt[k] += 1
    for k in acc:
        print(p, k[:60], "launches", cnt[k], {c: f"{v / cnt[k]:.4g}" for c, v in acc[k].items()})
PY
