#!/bin/bash
export QK_CACHE_DIR=/tmp/qkc
mkdir -p gpurun_out
echo "== ring kernel (variant 20) parity" 
QK_VARIANT=20 timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5
echo "== timings"
tools/quick_bench.sh "17:2 20:2"
