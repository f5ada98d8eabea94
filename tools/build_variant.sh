#!/bin/bash
# A product-library build that differs in ONE source file's compile-time defines, for same-box A/B (tools/lib_ab.sh, tools/attn_ab.sh):
#   tools/build_variant.sh attn_enc lsum -DLA_ATTN_LSUM_MFMA=1   ->  labelanything_amd/libla_attn_enc_lsum.so
set -e
SRC=$1; TAG=$2; shift 2
cd "$(dirname "$0")/../labelanything_amd/csrc"
make -j8 > /dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function "$@" -c $SRC.hip -o build/${SRC}__$TAG.o
objs=$(ls build/*.o | grep -v "__" | grep -v "build/$SRC.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs build/${SRC}__$TAG.o -o ../libla_${SRC}_$TAG.so
echo "built labelanything_amd/libla_${SRC}_$TAG.so ($*)"
