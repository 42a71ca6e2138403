#!/bin/sh
# builds tools/bin/convbench (links the in-tree libsemseg_hip.so; run it on the GPU box)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/convbench.hip -o tools/bin/convbench \
  -Lsemantic-segmentation_amd/lib -lsemseg_hip -Wl,-rpath,'$ORIGIN/../../semantic-segmentation_amd/lib'
