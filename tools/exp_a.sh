#!/bin/bash
mkdir -p gpurun_out
echo "== pairs"
bash tools/pmc_probe.sh pairs "20:2" 2>&1 | tail -2
echo "== quads"
QK_QUADS=1 bash tools/pmc_probe.sh quads "20:2" 2>&1 | tail -2
