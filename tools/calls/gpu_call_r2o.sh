#!/bin/bash
# Round 2, call O: fresh kernel trace of the current build (per-dispatch timestamps kept).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
mkdir -p gpurun_out/r2o_prof
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2o_prof -o r2o -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/r2o_bench.log 2>&1
echo "prof rc=$?"
ls -la gpurun_out/r2o_prof | head; find gpurun_out/r2o_prof -name '*.csv' | head
grep -h '^{' gpurun_out/r2o_bench.log | cut -c1-300
