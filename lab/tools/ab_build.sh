#!/bin/bash
# same-box A/B of compile-time switches: lab/tools/ab_build.sh "label|-DX=1 -DY=2" "label2|..." ...
# rebuilds libqkgram.so with the extra flags, runs the cfg4 bench (2 steps, no CPU baseline) per setting, then restores
# the default build.  QK_AB_ARGS adds bench arguments (e.g. "--config 5").
export QK_CACHE_DIR=${QK_CACHE_DIR:-/tmp/qkc}
mkdir -p gpurun_out
cs=qml-cutensornet_amd/csrc
build() { hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC $1 -o qml-cutensornet_amd/libqkgram.so $cs/qkgram.hip $cs/qk_lab.hip $cs/qk_build.hip 2> gpurun_out/ab_build.err || { tail -20 gpurun_out/ab_build.err; return 1; }; }
for spec in "$@"; do
  label=${spec%%|*}; flags=${spec#*|}
  build "$flags" || continue
  timeout -k 10 300 python bench.py --cpu-seconds 0 --steps 2 $QK_AB_ARGS > gpurun_out/ab_$label.json 2> gpurun_out/ab_$label.err || { tail -5 gpurun_out/ab_$label.err; exit 1; }
  python - <<PY
import json
d = json.loads(open("gpurun_out/ab_$label.json").read().strip().splitlines()[-1])
print("$label [$flags]: ms_per_step %.1f kernel_ms %.1f frac %.4f diag_err %.1e" % (d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["config"]["diag_err"]))
PY
done
build ""
