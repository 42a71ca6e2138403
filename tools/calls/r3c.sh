#!/bin/bash
# Round 3, call C: phase timing of the persistent trunk conv (tools/tilebench.py --timing, -DSSA_TILE_TIMING build).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
timeout 600 python tools/tilebench.py --timing > gpurun_out/r3c_timing.log 2>&1
echo "rc=$?" >> gpurun_out/r3c_timing.log
cat gpurun_out/r3c_timing.log
