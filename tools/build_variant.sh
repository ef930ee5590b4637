#!/bin/bash
# tools/build_variant.sh <tag> <file.hip> <extra hipcc flags...> — build variants/libotter_<tag>.so: the in-tree objects with ONE
# source recompiled under extra flags (ablation / A-B switches).  Run here (CPU container); the .so travels to the GPU box.
set -e
TAG=$1; SRC=$2; shift 2
cd "$(dirname "$0")/.."
mkdir -p variants/_obj
O=variants/_obj/${TAG}_$(basename $SRC .hip).o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -pthread "$@" -c otter_amd/csrc/$SRC -o $O
OBJS=""
for f in common index encoder gemm_bf16; do
  if [ "$f.hip" == "$SRC" ]; then OBJS="$OBJS $O"; else OBJS="$OBJS otter_amd/csrc/_obj/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o variants/libotter_$TAG.so $OBJS
echo variants/libotter_$TAG.so
