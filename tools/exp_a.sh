#!/bin/bash
export QK_CACHE_DIR=/tmp/qkc
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "f32" 2>&1 | tail -3
timeout -k 10 300 python bench.py --cpu-seconds 0 --steps 2 --precision f32 > gpurun_out/b32.json 2> gpurun_out/b32.err || tail -5 gpurun_out/b32.err
python - <<PY
import json
for f in ("b32",):
    d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1])
    print(f, d["dtype"], "ms %.1f kernel %.1f frac %.4f value %.0f"%(d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["value"]), {k:v for k,v in d["config"].items() if k.startswith("f32") or k=="diag_err"})
PY
