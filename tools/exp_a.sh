#!/bin/bash
export QK_CACHE_DIR=/tmp/qkc
mkdir -p gpurun_out
echo "== default parity"
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
echo "== timings"
tools/quick_bench.sh "20:2 26:2 17:2"
