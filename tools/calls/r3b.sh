#!/bin/bash
# Round 3, call B: micro-benchmark of the trunk conv kernels per problem / per level (tools/tilebench.py).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
timeout 600 python tools/tilebench.py 20 > gpurun_out/r3b_tilebench.log 2>&1
echo "rc=$?" >> gpurun_out/r3b_tilebench.log
cat gpurun_out/r3b_tilebench.log
