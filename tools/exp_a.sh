#!/bin/bash
export QK_CACHE_DIR=/tmp/qkc
mkdir -p gpurun_out
echo "== default full gpu suite"
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for v in 0 2 13 14 16 17 21 23; do
echo "== lab variant $v parity"
QK_VARIANT=$v timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not f32" 2>&1 | tail -1
done
QK_VARIANT=24 QK_WGS_PER_CU=4 timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not f32" 2>&1 | tail -1
echo "== timings"
tools/quick_bench.sh "20:2 17:2"
