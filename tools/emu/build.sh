#!/bin/sh
# Builds tools/emu/build/libsemseg_emu.so: the UNCHANGED kernel sources of semantic-segmentation_amd/csrc compiled
# for the host against the CPU emulation shim (tools/emu/include).  Test infrastructure only.
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(cd "$HERE/../.." && pwd)
SRC="$ROOT/semantic-segmentation_amd/csrc"
# sh tools/emu/build.sh f16 : the fp16-storage build (-DSSA_ELEM_F16) -> build_f16/libsemseg_emu.so
if [ "$1" = "f16" ]; then OUT="$HERE/build_f16"; EXTRA="-DSSA_ELEM_F16"; else OUT="$HERE/build"; EXTRA=""; fi
CXX=${EMU_CXX:-/opt/rocm/lib/llvm/bin/clang++}
FLAGS="-x c++ -std=c++17 -O2 -g0 -fPIC -DSSA_EMU $EXTRA -I$HERE/include -Wno-unused-function -Wno-unused-value -Wno-unknown-pragmas -Wno-pass-failed"
mkdir -p "$OUT"
pids=""
objs=""
for f in "$SRC"/*.hip "$HERE/emu_runtime.cpp"; do
  o="$OUT/$(basename "$f" | sed 's/\.[a-z]*$//').o"
  objs="$objs $o"
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ "$SRC/common.h" -nt "$o" ] || [ "$SRC/group.h" -nt "$o" ] || \
     [ "$HERE/include/hip/hip_runtime.h" -nt "$o" ] || [ "$ROOT/include/semseg_hip.h" -nt "$o" ]; then
    $CXX $FLAGS -c "$f" -o "$o" &
    pids="$pids $!"
  fi
done
for p in $pids; do wait "$p"; done
$CXX -shared -fPIC -o "$OUT/libsemseg_emu.so" $objs -lpthread
echo "built $OUT/libsemseg_emu.so"
