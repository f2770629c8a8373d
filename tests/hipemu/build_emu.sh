#!/bin/bash
# Build the HOST emulation of libml3d_hip.so (tests only; see include/hip/hip_runtime.h).
# Safe under concurrent callers (pytest-xdist workers, the subprocesses of test_emulated_api.py): one builder at a time
# (flock), the library is relinked only when an object changed, and it appears under its name by an atomic rename.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
CXX=${HIPEMU_CXX:-/opt/rocm/lib/llvm/bin/clang++}
OUT="$HERE/build"
mkdir -p "$OUT"
exec 9>"$OUT/.lock"
flock 9
SRCS=$(ls "$ROOT"/open3d-ml_amd/csrc/*.hip)
FLAGS="-std=c++17 -O1 -g -fPIC -DML3D_TEST_HOOKS -ffp-contract=off -fno-fast-math -Wno-unused-value -Wno-deprecated-declarations ${HIPEMU_EXTRA}"
OBJS=""
RELINK=0
[ -f "$OUT/libml3d_emu.so" ] || RELINK=1
for s in $SRCS "$HERE/hipemu.cpp"; do
  o="$OUT/$(basename "$s").o"
  if [ ! -f "$o" ] || [ "$s" -nt "$o" ] || [ "$HERE/include/hip/hip_runtime.h" -nt "$o" ] || \
     [ -n "$(find "$ROOT/open3d-ml_amd/csrc" "$ROOT/include" -name '*.h' -newer "$o" 2>/dev/null)" ]; then
    $CXX $FLAGS -I"$HERE/include" -I"$ROOT/include" -I"$ROOT/open3d-ml_amd/csrc" -x c++ -c "$s" -o "$o"
    RELINK=1
  fi
  [ "$o" -nt "$OUT/libml3d_emu.so" ] && RELINK=1
  OBJS="$OBJS $o"
done
if [ $RELINK = 1 ]; then
  $CXX -shared -rdynamic -o "$OUT/libml3d_emu.so.tmp.$$" $OBJS -lpthread
  mv -f "$OUT/libml3d_emu.so.tmp.$$" "$OUT/libml3d_emu.so"
fi
echo "$OUT/libml3d_emu.so"
