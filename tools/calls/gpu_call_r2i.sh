#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD/semantic-segmentation_amd:$PWD
timeout 600 python -m pytest tests/test_attnscale_gpu.py tests/test_parity_1024_gpu.py -q -s -m gpu > gpurun_out/r2i_tests.log 2>&1; echo "tests rc=$?"
tail -3 gpurun_out/r2i_tests.log
