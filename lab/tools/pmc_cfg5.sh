#!/bin/bash
mkdir -p gpurun_out
QK_BENCH_ARGS="--config cfg5" bash lab/tools/pmc_probe.sh cfg5 "20:2" > gpurun_out/pmc_cfg5.log 2>&1
tail -2 gpurun_out/pmc_cfg5.log
python - <<PY
import json
d=json.loads(open("profiles/r01/bench_cfg5_supplementary.json").read().strip().splitlines()[-1])
print("algorithmic GB per launch", d["roofline"]["algorithmic_gbytes_per_launch"])
PY
