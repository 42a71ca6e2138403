#!/bin/bash
# Round 2, call AB: bench.py after the capture-error bookkeeping; conftest deselection of the two-GPU tests.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_multi_gpu_rccl.py tests/test_ddp_graph_gpu.py tests/test_rccl_direct_gpu.py -q -m gpu > gpurun_out/r2ab_tests.log 2>&1
echo "rc=$?"; tail -4 gpurun_out/r2ab_tests.log
timeout 120 python bench.py --no-cpu-baseline --no-roofline 2>&1 | grep '^{' | cut -c1-420
