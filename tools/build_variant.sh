#!/bin/bash
# tools/build_variant.sh <name> <file.hip> [-DFLAG ...]: rebuilds ONE source of the library with extra flags and links it with the
# in-tree objects of the others into build/variants/<name>/libtfusion_hip.so (select it with TFUSION_LIB=...; same-box A/B, ablations).
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
out=build/variants/$name; mkdir -p $out
python -m transfusion_amd.build >/dev/null
obj=$out/${src%.hip}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -Wno-unused-result -DTF_EXPERIMENTS "$@" -c transfusion_amd/csrc/$src -o $obj
objs=""
for f in gemm_bf16 wgrad_multi attn_bf16 attn_x3 rowops heads comm tf_api; do
  if [ "$f.hip" == "$src" ]; then objs="$objs $obj"; else objs="$objs transfusion_amd/csrc/_obj/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libtfusion_hip.so $objs -ldl
echo $out/libtfusion_hip.so
