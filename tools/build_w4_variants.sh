#!/bin/bash
# Product-library builds that differ only in gemm_w4.hip's compile-time knobs, for same-box A/B (tools/lib_ab.sh):
#   tools/build_w4_variants.sh "GROUP=2" "GROUP=4" ...   ->  labelanything_amd/libla_w4_GROUP2.so ...
set -e
cd "$(dirname "$0")/../labelanything_amd/csrc"
make -j8 > /dev/null
for v in "$@"; do
  tag=$(echo "$v" | tr -d '= ' | tr ',' '_')
  defs=$(echo "$v" | tr ',' ' ' | sed 's/\([A-Z_0-9]*\)=\([0-9]*\)/-DLA_W4_\1=\2/g')
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $defs -c gemm_w4.hip -o build/gemm_w4_$tag.o
  objs=$(ls build/*.o | grep -v "gemm_w4")
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs build/gemm_w4_$tag.o -o ../libla_w4_$tag.so
  echo "built labelanything_amd/libla_w4_$tag.so ($defs)"
done
