#!/bin/bash
# 20 back-to-back Grams of cfg4 (fp64) and of cfg5 (small-bond kernel): the invariants must hold on the last one
export QK_CACHE_DIR=${QK_CACHE_DIR:-/tmp/qkc}
mkdir -p gpurun_out
for c in cfg4 cfg5; do
  timeout -k 10 900 python bench.py --config $c --cpu-seconds 0 --steps 20 --warmup 1 > gpurun_out/soak_$c.json 2> gpurun_out/soak_$c.err || { tail -3 gpurun_out/soak_$c.err; exit 1; }
  python -c "
import json; d=json.loads(open('gpurun_out/soak_$c.json').read().strip().splitlines()[-1]); print('$c', 'steps', d['steps'], 'ms/step %.2f'%d['ms_per_step'], 'diag_err %.1e sym_err %.1e'%(d['config']['diag_err'], d['config']['sym_err']))"
done
