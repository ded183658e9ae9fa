#!/bin/bash
# usage: lab/tools/exp_fused.sh "name:-Dflag,-Dflag name2:..."  -- builds experiment variants of the LAB library HERE (no GPU) into gpurun_exp/
R=$(cd $(dirname $0)/.. && pwd); C=$R/qml-cutensornet_amd/csrc; rm -rf $R/gpurun_exp; mkdir -p $R/gpurun_exp
for spec in $1; do
  name=${spec%%:*}; flags=${spec#*:}; [ "$flags" = "$spec" ] && flags=""
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DQK_LAB ${flags//,/ } -o $R/gpurun_exp/lib_$name.so $C/qkgram.hip $C/qk_lab.hip $C/qk_build.hip &
done
wait; ls $R/gpurun_exp
