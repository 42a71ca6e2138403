#!/bin/bash
# Round 2, call K: fresh kernel trace of the current build (per-dispatch timestamps kept).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
mkdir -p gpurun_out/r2k_prof
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2k_prof -o r2k -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/r2k_bench.log 2>&1
echo "prof rc=$?"
ls -la gpurun_out/r2k_prof | head; find gpurun_out/r2k_prof -name '*.csv' | head
grep -h '^{' gpurun_out/r2k_bench.log | cut -c1-300
