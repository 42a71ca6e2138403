#!/bin/bash
# The full `-m gpu` suite alone (no smoke), bounded for a short GPU budget.  bash tools/calls/suite2.sh [tag] [seconds]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=${1:-suite2}
mkdir -p gpurun_out
timeout ${2:-1500} python -m pytest tests -q -m gpu --durations=30 > gpurun_out/${T}_gpu_suite.log 2>&1
echo "suite rc=$?"
tail -45 gpurun_out/${T}_gpu_suite.log
