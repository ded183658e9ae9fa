#!/bin/bash
export QK_CACHE_DIR=/tmp/qkc
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --cpu-seconds 0 --steps 1 --warmup 0 > /dev/null 2> gpurun_out/prime.err
timeout -k 10 400 python tools/rank_share_bench.py 1 16 64 512 2>&1 | grep -v amdgpu.ids
timeout -k 10 400 python tools/rank_share_bench.py 8 16 64 512 2>&1 | grep -v amdgpu.ids
