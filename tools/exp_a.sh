#!/bin/bash
export QK_CACHE_DIR=/tmp/qkc
mkdir -p gpurun_out
echo "== default (ring) full gpu suite"
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for v in 24; do
echo "== variant $v parity"
QK_VARIANT=$v QK_WGS_PER_CU=4 timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
done
echo "== timings"
tools/quick_bench.sh "20:2 24:4 24:3"
