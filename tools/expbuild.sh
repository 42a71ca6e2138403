#!/bin/sh
# Experiment build: lib/libsemseg_hip_<name>.so = the product objects with ONE source recompiled under extra flags.
#   sh tools/expbuild.sh <name> <file.hip> [-DFLAG=..]...      (tools/headbench.py --lib <name>, tools/tilebench.py --timing)
set -e
cd "$(dirname "$0")/../semantic-segmentation_amd/csrc"
name=$1; src=$2; shift 2
make -s -j8
mkdir -p build_$name
base=$(basename $src .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c $src -o build_$name/$base.o
objs=""
for o in build/*.o; do
  if [ "$(basename $o)" = "$base.o" ]; then objs="$objs build_$name/$base.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libsemseg_hip_$name.so $objs
echo ../lib/libsemseg_hip_$name.so
