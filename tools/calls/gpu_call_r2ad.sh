#!/bin/bash
# Round 2, call AD: end-to-end tests (incl. smoke) on the final kernel sources.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
mkdir -p gpurun_out
timeout 170 python -m pytest tests/test_e2e_gpu.py tests/test_fuse_bwd_gpu.py -q -m gpu > gpurun_out/r2ad_tests.log 2>&1
echo "rc=$?"; tail -3 gpurun_out/r2ad_tests.log
