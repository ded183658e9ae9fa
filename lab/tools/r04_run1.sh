#!/bin/bash
# round 4, first GPU call: parity suite, then the bench lines with the cold step
mkdir -p gpurun_out
export QK_CACHE_DIR=/tmp/qkc
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/suite.log 2>&1; rc=$?; echo "rc $rc" >> gpurun_out/suite.log
tail -5 gpurun_out/suite.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --steps 5 --warmup 2 > gpurun_out/bench_cfg4.json 2> gpurun_out/bench_cfg4.err || { tail -20 gpurun_out/bench_cfg4.err; exit 1; }
grep "cold Gram" gpurun_out/bench_cfg4.err
timeout -k 10 200 python bench.py --config cfg3 --steps 10 --warmup 3 --cpu-seconds 0 > gpurun_out/bench_cfg3.json 2> gpurun_out/bench_cfg3.err || { tail -20 gpurun_out/bench_cfg3.err; exit 1; }
grep "cold Gram" gpurun_out/bench_cfg3.err
QK_BENCH_DEVICE_BUILD=0 timeout -k 10 300 python bench.py --config cfg5 --steps 5 --warmup 2 --cpu-seconds 0 > gpurun_out/bench_cfg5.json 2> gpurun_out/bench_cfg5.err || { tail -20 gpurun_out/bench_cfg5.err; exit 1; }
grep "cold Gram" gpurun_out/bench_cfg5.err
python - <<'PY'
import json
for c in ("cfg4","cfg3","cfg5"):
    d=json.load(open(f"gpurun_out/bench_{c}.json"))
    r=d["roofline"]
    print(c, "ms/step %.2f cold %.1f plan %.1f derive %.1f frac %.4f reuse_GB %.1f" % (d["ms_per_step"], d["cold_step_ms"], d["plan_ms"], d["derive_ms"], r["frac"], r["tile_reuse_gbytes"]), [ (l["kernel_ms"], l["bound"]) for l in r["launches"]], d["config"]["host_cpu"])
PY
