#!/bin/bash
export QK_CACHE_DIR=/tmp/qkc
mkdir -p gpurun_out
{ tools/quick_bench.sh "20:2 20:2"
timeout -k 10 300 python bench.py --cpu-seconds 0 --steps 2 --precision f32 > gpurun_out/b32.json 2> gpurun_out/b32.err; python -c "
import json; d=json.loads(open('gpurun_out/b32.json').read().strip().splitlines()[-1]); print('f32 ms %.1f'%d['ms_per_step'])"
QK_CHIS=48,64,128 timeout -k 10 300 python tools/chi_scan.py 60 181 2>&1 | grep -v amdgpu.ids
timeout -k 10 500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2; } > gpurun_out/prio.log 2>&1
