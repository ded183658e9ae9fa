#!/bin/bash
export QK_CACHE_DIR=/tmp/qkc
mkdir -p gpurun_out
echo "start" > gpurun_out/wave_dbg.txt
QK_CHIS=2 timeout -k 5 40 python tools/chi_scan.py 6 3 >> gpurun_out/wave_dbg.txt 2>&1
rc=$?; echo "rc $rc" >> gpurun_out/wave_dbg.txt
[ $rc -eq 0 ] || exit 1
echo "== small/wave tests" >> gpurun_out/wave_dbg.txt
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q >> gpurun_out/wave_dbg.txt 2>&1 || exit 2
echo "== chi scan wave vs small" >> gpurun_out/wave_dbg.txt
for w in 1 0; do QK_WAVE=$w QK_CHIS=2,4,8,16 timeout -k 10 200 python tools/chi_scan.py 60 181 2>&1 | grep -v amdgpu.ids >> gpurun_out/wave_dbg.txt || exit 3; done
for w in 1 0; do QK_WAVE=$w timeout -k 10 300 python bench.py --config cfg2 --cpu-seconds 0 --steps 5 > gpurun_out/bs_cfg2_$w.json 2> gpurun_out/bs_cfg2_$w.err || { tail -3 gpurun_out/bs_cfg2_$w.err; exit 4; }; python -c "
import json; d=json.loads(open('gpurun_out/bs_cfg2_$w.json').read().strip().splitlines()[-1]); print('cfg2 wave=$w', 'ms %.3f kernel %.3f value %.0f diag_err %.1e'%(d['ms_per_step'], d['roofline']['kernel_ms'], d['value'], d['config']['diag_err']))" >> gpurun_out/wave_dbg.txt; done
