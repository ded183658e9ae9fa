#!/bin/bash
# usage: lab/tools/variant_bench.sh "0 1 2"   -- chi scan + cfg4 bench per kernel variant (GPU box)
export QK_CACHE_DIR=${QK_CACHE_DIR:-/tmp/qkc}
mkdir -p gpurun_out
for v in $1; do
  echo "== variant $v: gpu tests"
  QK_VARIANT=$v timeout -k 10 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
  echo "== variant $v: chi scan"
  QK_VARIANT=$v timeout -k 10 200 python lab/tools/chi_scan.py 60 91 2>&1 | grep -v amdgpu.ids
  echo "== variant $v: cfg4 bench"
  QK_VARIANT=$v timeout -k 10 300 python bench.py --cpu-seconds 0 --steps 2 > gpurun_out/vb_$v.json 2> gpurun_out/vb_$v.err || tail -5 gpurun_out/vb_$v.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/vb_$v.json"))
    print("variant $v: ms_per_step %.1f  kernel_ms %.1f  frac %.4f  entries/s %.0f" % (d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["value"]))
except Exception as e:
    print("variant $v: bench failed", e)
PY
done
