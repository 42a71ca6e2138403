#!/bin/bash
# Round 6 call T: the OCR block's small chains as parallel branches (SSA_FORK tag ocr: SpatialGather passes, key / value
# stacks): parity tests, the step with the tag on / off.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=${1:-r6t}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_e2e_gpu.py tests/test_parity_1024_gpu.py tests/test_graphed_step_gpu.py tests/test_siblings_gpu.py -q -x -m gpu > gpurun_out/${T}_parity.log 2>&1
echo "parity rc=$?"; tail -4 gpurun_out/${T}_parity.log
for cfg in "fuse,loss,ocr" "fuse,loss" "fuse,loss,ocr" "fuse,loss" "fuse,loss,ocr"; do
  SSA_FORK=$cfg timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --eager-steps 0 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${T}_bench.json").read().strip().splitlines()[-1])
    print("fork=$cfg: %.3f ms  %.2f img/s" % (d["ms_per_step"], d["value"]))
except Exception as e:
    print("fork=$cfg failed:", e); print(open("gpurun_out/${T}_bench.err").read()[-1500:])
PY
done
