#!/bin/bash
# Build the HOST emulation of libml3d_hip.so (tests only; see include/hip/hip_runtime.h).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
CXX=${HIPEMU_CXX:-/opt/rocm/lib/llvm/bin/clang++}
OUT="$HERE/build"
mkdir -p "$OUT"
SRCS=$(ls "$ROOT"/open3d-ml_amd/csrc/*.hip)
FLAGS="-std=c++17 -O1 -g -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-value -Wno-deprecated-declarations ${HIPEMU_EXTRA}"
OBJS=""
for s in $SRCS "$HERE/hipemu.cpp"; do
  o="$OUT/$(basename "$s").o"
  if [ ! -f "$o" ] || [ "$s" -nt "$o" ] || [ "$HERE/include/hip/hip_runtime.h" -nt "$o" ] || \
     [ -n "$(find "$ROOT/open3d-ml_amd/csrc" "$ROOT/include" -name '*.h' -newer "$o" 2>/dev/null)" ]; then
    $CXX $FLAGS -I"$HERE/include" -I"$ROOT/include" -I"$ROOT/open3d-ml_amd/csrc" -x c++ -c "$s" -o "$o"
  fi
  OBJS="$OBJS $o"
done
$CXX -shared -rdynamic -o "$OUT/libml3d_emu.so" $OBJS -lpthread
echo "$OUT/libml3d_emu.so"
