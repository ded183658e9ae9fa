#!/bin/bash
# same-box A/B of the device builder: lab/tools/ab_builder.sh "<bench spec>" "label|env assignments|-D flags" ...
mkdir -p gpurun_out
spec=$1; shift
build() { hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC $1 -o qml-cutensornet_amd/libqkgram.so qml-cutensornet_amd/csrc/qkgram.hip qml-cutensornet_amd/csrc/qk_lab.hip qml-cutensornet_amd/csrc/qk_build.hip 2> gpurun_out/abb_build.err || { tail -20 gpurun_out/abb_build.err; return 1; }; }
last="__none__"
for v in "$@"; do
  label=${v%%|*}; rest=${v#*|}; envs=${rest%%|*}; flags=${rest#*|}
  if [ "$flags" != "$last" ]; then build "$flags" || continue; last=$flags; fi
  echo "== $label [$envs] [$flags]"
  env $envs QK_BUILD_DEBUG=1 timeout -k 10 300 python lab/tools/dev_builder_bench.py $spec 2>&1 | grep -v amdgpu.ids || exit 1
done
build ""
