#!/bin/bash
export QK_CACHE_DIR=/tmp/qkc
mkdir -p gpurun_out
echo "== gpu suite"
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
echo "== chi scan small"
for sm in 1 0; do QK_SMALL=$sm QK_CHIS=2,8,16,24,32 timeout -k 10 300 python tools/chi_scan.py 60 181 2>&1 | grep -v amdgpu.ids; done
echo "== cfg5 / cfg2"
for sm in 1 0; do for c in cfg5 cfg2; do QK_SMALL=$sm timeout -k 10 600 python bench.py --config $c --cpu-seconds 0 --steps 2 > gpurun_out/bs_${c}_$sm.json 2> gpurun_out/bs_${c}_$sm.err || tail -3 gpurun_out/bs_${c}_$sm.err; python -c "
import json; d=json.loads(open('gpurun_out/bs_${c}_$sm.json').read().strip().splitlines()[-1]); print('$c small=$sm', 'ms %.2f value %.0f diag_err %.1e'%(d['ms_per_step'], d['value'], d['config']['diag_err']))"; done; done
