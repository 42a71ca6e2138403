#!/bin/bash
# Round 2, call AA: the GPU tests that have not run since the single weight-gradient flush became the default.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
mkdir -p gpurun_out
timeout 540 python -m pytest tests/test_siblings_gpu.py tests/test_deepv3_gpu.py tests/test_ddp_gpu.py tests/test_ddp_graph_gpu.py tests/test_rccl_direct_gpu.py tests/test_optim_gpu.py tests/test_data_gpu.py tests/test_kernels_gpu.py -q -m gpu --durations=8 > gpurun_out/r2aa_tests.log 2>&1
echo "rc=$?"; tail -14 gpurun_out/r2aa_tests.log
