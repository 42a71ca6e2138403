#!/bin/bash
# Round 6 call K: tools/wgradbench.py -- the trunk's weight-gradient kernels per channel class, all-taps form on / off.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== all-taps on"; timeout 600 python tools/wgradbench.py 10 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6k_wgradbench_new.txt
echo "== all-taps off"; SSA_WGRAD_ALL=0 timeout 600 python tools/wgradbench.py 10 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6k_wgradbench_old.txt
