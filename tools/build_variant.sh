#!/bin/bash
# Build an experimental variant of the library next to the product one:
#   tools/build_variant.sh NAME file.hip "-DMACRO=1 ..."   -> build_variants/libcra5_NAME.so
# Select it at run time with CRA5_LIB=build_variants/libcra5_NAME.so (A/B inside ONE gpurun call).
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; defs=$3
python -m cra5_amd.build >/dev/null
mkdir -p build_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -pragma-unroll-threshold=262144 -Wno-inline-asm $defs -c cra5_amd/csrc/$src -o build_variants/${name}_${src%.hip}.o
objs=""
for f in host_entropy gemm_f32 gemm_split_f16 attention_f32 attention_split_f16 elementwise hyper runtime; do
  if [ "$f.hip" == "$src" ]; then objs="$objs build_variants/${name}_${f}.o"; else objs="$objs cra5_amd/csrc/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_variants/libcra5_${name}.so $objs -lpthread
echo build_variants/libcra5_${name}.so
