#!/bin/bash
# Run the HIP sources through the host emulator under AddressSanitizer: global / LDS / workspace indexing bugs show up
# here without a GPU (the ucontext warning ASan prints at start-up is expected).
#   tools/emu_asan.sh [python-script]     default: tools/emu_asan_check.py
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT=${EMU_ASAN_DIR:-/tmp/ml3d_emu_asan}
mkdir -p $OUT && cp -r $ROOT/tests/hipemu/. $OUT/ && rm -rf $OUT/build
sed -i "s#ROOT=\"\$(cd \"\$HERE/../..\" \&\& pwd)\"#ROOT=$ROOT#" $OUT/build_emu.sh
HIPEMU_EXTRA="-fsanitize=address -fno-omit-frame-pointer" bash $OUT/build_emu.sh > /dev/null
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
ML3D_EMU_LIB=$OUT/build/libml3d_emu.so LD_PRELOAD=$RT \
  ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:verify_asan_link_order=0 \
  python ${1:-$ROOT/tools/emu_asan_check.py}
