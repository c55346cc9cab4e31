#!/bin/bash
# tools/build_variant_multi.sh <name> <file.hip>=<flags,comma separated> ...: like build_variant.sh for SEVERAL sources with their own flags
set -e
cd "$(dirname "$0")/.."
name=$1; shift
out=build/variants/$name; mkdir -p $out
python -m transfusion_amd.build >/dev/null
declare -A built
for spec in "$@"; do
  src=${spec%%=*}; flags=${spec#*=}; flags=${flags//,/ }
  obj=$out/${src%.hip}.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -Wno-unused-result -DTF_EXPERIMENTS $flags -c transfusion_amd/csrc/$src -o $obj
  built[${src%.hip}]=$obj
done
objs=""
for f in gemm_bf16 wgrad_multi attn_bf16 attn_x3 rowops heads comm tf_api; do
  if [ -n "${built[$f]}" ]; then objs="$objs ${built[$f]}"; else objs="$objs transfusion_amd/csrc/_obj/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libtfusion_hip.so $objs -ldl
echo $out/libtfusion_hip.so
