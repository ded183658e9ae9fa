#!/bin/bash
# uniform-chi scan (181 states = 16471 pairs = 32.2 rounds of 512 workgroups)
mkdir -p gpurun_out
export QK_CHIS=16,32,48,64,96,128,256
for v in 17 20; do echo "== variant $v"; QK_VARIANT=$v timeout -k 10 300 python lab/tools/chi_scan.py 60 181 2>&1 | grep -v amdgpu.ids; done
export QK_CHIS=32,64,128,256
for f in 4 2 1; do echo "== variant 13 flags $f"; QK_VARIANT=13 QK_DEBUG_FLAGS=$f timeout -k 10 300 python lab/tools/chi_scan.py 60 181 2>&1 | grep -v amdgpu.ids; done
