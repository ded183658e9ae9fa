#!/bin/bash
export QK_CACHE_DIR=/tmp/qkc
mkdir -p gpurun_out
timeout -k 10 900 python bench.py --config cfg5 --cpu-seconds 0 --steps 2 --precision f32 > gpurun_out/b5_32.json 2> gpurun_out/b5_32.err || tail -5 gpurun_out/b5_32.err
timeout -k 10 600 python bench.py --config cfg3 --cpu-seconds 0 --steps 2 --precision f32 > gpurun_out/b3_32.json 2> gpurun_out/b3_32.err || tail -5 gpurun_out/b3_32.err
python - <<PY
import json
for f in ("b5_32","b3_32"):
    d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1])
    print(f, d["dtype"], "ms %.1f kernel %.1f frac %.4f value %.0f"%(d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["value"]), {k:v for k,v in d["config"].items() if k.startswith("f32") or k in ("diag_err","max_bond_max")})
PY
