#!/bin/bash
# Round-end validation on the GPU box: full GPU suite, smoke, bench (with CPU baseline), rocprof recipe, supplementary
# configs and the uniform-chi scan of the shipped kernel.  Outputs under gpurun_out/final_<tag>/.
set -o pipefail
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/final_$TAG
mkdir -p "$OUT"
export QK_CACHE_DIR=/tmp/qkc
cd "$R"
timeout -k 10 600 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1 || { tail -20 "$OUT/pytest_gpu.log"; exit 1; }
tail -2 "$OUT/pytest_gpu.log"
timeout -k 10 300 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" > "$OUT/smoke.log" 2>&1 || { tail -20 "$OUT/smoke.log"; exit 2; }
tail -1 "$OUT/smoke.log"
timeout -k 10 600 python bench.py > "$OUT/bench_cfg4.json" 2> "$OUT/bench_cfg4.err" || { tail -20 "$OUT/bench_cfg4.err"; exit 3; }
cat "$OUT/bench_cfg4.json"
bash profiles/run_rocprof.sh $TAG > "$OUT/rocprof.log" 2>&1 || { tail -20 "$OUT/rocprof.log"; exit 4; }
cd "$R"
for c in cfg2 cfg3 cfg5; do
  timeout -k 10 600 python bench.py --config $c --cpu-seconds 0 > "$OUT/bench_$c.json" 2> "$OUT/bench_$c.err" || { tail -5 "$OUT/bench_$c.err"; exit 5; }
done
QK_CHIS=16,32,48,64,96,128,256 timeout -k 10 300 python lab/tools/chi_scan.py 60 181 2>&1 | grep -v amdgpu.ids > "$OUT/chi_scan.txt" || exit 6
cat "$OUT/chi_scan.txt"
