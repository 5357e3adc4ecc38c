#!/bin/bash
# Tile-order experiment: libraries whose big-tile GEMM numbers its tiles in bands of G tile rows
# (production: 4), each with the in-kernel trace (-DCRA5_GEMM_TRACE: per-block timestamps + shader clock):
#   tools/build_gemm_group_variants.sh 1 2 8 16 64  -> build_variants/libcra5_grpG.so
# The production source is not touched: the constant is rewritten in a scratch copy.
set -e
cd "$(dirname "$0")/.."
python -m cra5_amd.build >/dev/null
mkdir -p build_variants
for G in "$@"; do
  sed "s/constexpr int GEMM_GROUP_M = 4;/constexpr int GEMM_GROUP_M = $G;/" cra5_amd/csrc/gemm_split_f16.hip > cra5_amd/csrc/_grp_tmp.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -pragma-unroll-threshold=262144 -Wno-inline-asm -DCRA5_GEMM_TRACE \
      -c cra5_amd/csrc/_grp_tmp.hip -o build_variants/grp${G}_gemm_split_f16.o
  rm -f cra5_amd/csrc/_grp_tmp.hip
  objs=""
  for f in host_entropy gemm_f32 gemm_split_f16 attention_f32 attention_split_f16 elementwise hyper runtime; do
    if [ "$f" == "gemm_split_f16" ]; then objs="$objs build_variants/grp${G}_gemm_split_f16.o"; else objs="$objs cra5_amd/csrc/$f.o"; fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_variants/libcra5_grp${G}.so $objs -lpthread
  echo build_variants/libcra5_grp${G}.so
done
