#!/bin/bash
# round 4, GPU call 7 (records): per-site cost on uniform small chains, uniform chains on the shipped shapes
mkdir -p gpurun_out
export QK_CACHE_DIR=/tmp/qkc
timeout -k 10 500 python tools/site_overhead.py > gpurun_out/site_overhead.txt 2>&1 || tail -5 gpurun_out/site_overhead.txt
grep -v "^+" gpurun_out/site_overhead.txt | tail -12
timeout -k 10 500 python tools/uniform_ab.py 40,48,56,64,96,128,192,256 QK_FUSED_WGS=0 > gpurun_out/uniform_shapes.txt 2>&1 || tail -5 gpurun_out/uniform_shapes.txt
grep "^cap " gpurun_out/uniform_shapes.txt | grep ms
