#!/bin/bash
# GPU box: chi scan of every experiment build in gpurun_exp/
R=${GRAFT_REPO_ROOT:-$(pwd)}
for lib in $R/gpurun_exp/lib_*.so; do
  echo "== $(basename $lib)"
  QK_FUSED=2 QK_LIB=$lib QK_CHIS=${QK_CHIS:-32,48,64,96,128} timeout -k 10 200 python3 $R/lab/tools/chi_scan.py 60 ${NS:-181} 2>&1 | grep -v amdgpu.ids
done
echo "== ring (QK_FUSED=0)"
QK_FUSED=0 QK_CHIS=${QK_CHIS:-32,48,64,96,128} timeout -k 10 200 python3 $R/lab/tools/chi_scan.py 60 ${NS:-181} 2>&1 | grep -v amdgpu.ids
