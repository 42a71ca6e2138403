#!/bin/bash
# smoke + the full `-m gpu` suite with per-test durations.  bash tools/calls/suite.sh [tag]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=${1:-suite}
mkdir -p gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${T}_smoke.log 2>&1
echo "smoke rc=$?: $(grep '^smoke' gpurun_out/${T}_smoke.log | tail -1)"
timeout 1500 python -m pytest tests -q -m gpu --durations=40 > gpurun_out/${T}_gpu_suite.log 2>&1
echo "suite rc=$?"
tail -60 gpurun_out/${T}_gpu_suite.log
