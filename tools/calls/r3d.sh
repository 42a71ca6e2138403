#!/bin/bash
# Round 3, call D: phase timing + micro-benchmark of the persistent trunk conv after a kernel change.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
timeout 300 python tools/tilebench.py --timing > gpurun_out/r3d_timing.log 2>&1
timeout 300 python tools/tilebench.py 20 > gpurun_out/r3d_tilebench.log 2>&1
timeout 300 python -m pytest tests/test_group_gpu.py -q -m gpu -x -k "basic_block or folded or conv_group" > gpurun_out/r3d_tests.log 2>&1
grep -v "Warning\|super()\|amdgpu.ids" gpurun_out/r3d_timing.log | head -40
grep -v "Warning\|super()\|amdgpu.ids" gpurun_out/r3d_tilebench.log
tail -3 gpurun_out/r3d_tests.log
