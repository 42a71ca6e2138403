#!/bin/bash
# Round-end evidence on the FINAL sources (after the halo-staging fix of conv_tile_p.hip and the opt-in conv_tile_q.hip):
# tools/calls/final.sh, then the two heaviest parity files on the new trunk-conv code.   bash tools/calls/final2.sh
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
bash tools/calls/final.sh r4fin2
timeout 500 python -m pytest tests/test_parity_1024_gpu.py tests/test_e2e_gpu.py -q -m gpu --durations=5 > gpurun_out/r4fin2_parity_tests.log 2>&1
echo "parity_1024 + e2e: $(tail -1 gpurun_out/r4fin2_parity_tests.log)"
grep -E "^FAILED|^ERROR" gpurun_out/r4fin2_parity_tests.log | head
