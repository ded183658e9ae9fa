#!/bin/bash
# usage: lab/tools/quick_bench.sh "<variant>:<wgs_per_cu> ..."   -- cfg4 bench only, per setting
export QK_CACHE_DIR=${QK_CACHE_DIR:-/tmp/qkc}
mkdir -p gpurun_out
for vw in $1; do
  v=${vw%%:*}; w=${vw##*:}
  b=${QK_PLAN_BLOCK:-16}
  QK_VARIANT=$v QK_WGS_PER_CU=$w timeout -k 10 300 python bench.py --cpu-seconds 0 --steps 2 > gpurun_out/qb_${v}_${w}.json 2> gpurun_out/qb_${v}_${w}.err || tail -5 gpurun_out/qb_${v}_${w}.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/qb_${v}_${w}.json").read().strip().splitlines()[-1])
    print("variant $v wgs/cu $w block $b prio ${QK_PRIO:-0}: ms_per_step %.1f  kernel_ms %.1f  frac %.4f  entries/s %.0f diag_err %.1e" % (d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["value"], d["config"]["diag_err"]))
except Exception as e:
    print("variant $v wgs $w: bench failed", e)
PY
done
