#!/bin/bash
# smoke + the full `-m gpu` suite on the current tree.  bash tools/calls/suite.sh [tag]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
T=${1:-suite}
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${T}_smoke.log 2>&1; echo "smoke rc=$?"
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/${T}_gpu_suite.log 2>&1; echo "suite rc=$?"
tail -4 gpurun_out/${T}_gpu_suite.log; grep "smoke" gpurun_out/${T}_smoke.log | tail -2
