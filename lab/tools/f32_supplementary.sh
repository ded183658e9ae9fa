#!/bin/bash
export QK_CACHE_DIR=/tmp/qkc
mkdir -p gpurun_out
for c in cfg5 cfg3 cfg4; do timeout -k 10 900 python bench.py --config $c --cpu-seconds 0 --steps 2 --precision f32 > gpurun_out/b_${c}_32.json 2> gpurun_out/b_${c}_32.err || tail -5 gpurun_out/b_${c}_32.err; done
python - <<PY
import json
for c in ("cfg5","cfg3","cfg4"):
    d=json.loads(open("gpurun_out/b_%s_32.json"%c).read().strip().splitlines()[-1])
    print(c, d["dtype"], "ms %.2f kernel %s value %.0f"%(d["ms_per_step"], d["roofline"]["kernel"], d["value"]), d["config"]["f32_vs_f64_max_abs"])
PY
