set -o pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03n/fetch5; mkdir -p $OUT
export QK_CACHE_DIR=/tmp/qkc QK_BENCH_DEVICE_BUILD=0
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --config cfg5 --steps 1 --warmup 0 --cpu-seconds 0 > $OUT/prime.json 2> $OUT/prime.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc -o pmc -- python3 $R/bench.py --config cfg5 --steps 1 --warmup 0 --cpu-seconds 0 > $OUT/pmc.json 2> $OUT/pmc.err || exit 2
python3 - $OUT <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/pmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'qk_sweep' in r['Kernel_Name']: print(r['Kernel_Name'][:50], r['Counter_Name'], r['Counter_Value'])
PY
