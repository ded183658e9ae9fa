#!/bin/bash
export QK_CACHE_DIR=/tmp/qkc
mkdir -p gpurun_out
echo "== gpu suite"
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 || exit 1
echo "== chi scan small"
QK_CHIS=2,8,16,24,32 timeout -k 10 300 python tools/chi_scan.py 60 181 2>&1 | grep -v amdgpu.ids || exit 2
echo "== cfg5 / cfg2"
for c in cfg5 cfg2; do timeout -k 10 600 python bench.py --config $c --cpu-seconds 0 --steps 2 > gpurun_out/bs_${c}_1.json 2> gpurun_out/bs_${c}_1.err || { tail -3 gpurun_out/bs_${c}_1.err; exit 3; }; python -c "
import json; d=json.loads(open('gpurun_out/bs_${c}_1.json').read().strip().splitlines()[-1]); print('$c small=1', 'ms %.2f value %.0f diag_err %.1e'%(d['ms_per_step'], d['value'], d['config']['diag_err']))"; done
timeout -k 10 600 python bench.py --config cfg5 --cpu-seconds 0 --steps 2 --precision f32 > gpurun_out/b5_32.json 2> gpurun_out/b5_32.err && python -c "
import json; d=json.loads(open('gpurun_out/b5_32.json').read().strip().splitlines()[-1]); print('cfg5 f32', 'ms %.2f value %.0f'%(d['ms_per_step'], d['value']), d['config']['f32_vs_f64_max_abs'])"
