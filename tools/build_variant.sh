#!/bin/bash
# A/B helper: build libml3d_hip.so with extra -D flags for ONE source into ml3d/lib/ab/<name>.so
# usage: tools/build_variant.sh <name> <source.hip> <extra flags...>
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
NAME=$1; SRC=$2; shift 2
OUT=$ROOT/open3d-ml_amd/ml3d/lib
mkdir -p $OUT/ab
make -s -C $ROOT/open3d-ml_amd/csrc >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -mllvm -amdgpu-mfma-vgpr-form -I$ROOT/include \
  -I$ROOT/open3d-ml_amd/csrc -Wno-unused-function "$@" -c $ROOT/open3d-ml_amd/csrc/$SRC -o $OUT/ab/$NAME.o
OBJS=$(ls $OUT/obj/*.o | grep -v "/${SRC%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $OUT/ab/$NAME.so $OBJS $OUT/ab/$NAME.o
echo $OUT/ab/$NAME.so
