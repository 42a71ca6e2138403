#!/bin/bash
# Round 6 call S: more parallel branches of the captured step -- the OCR block's auxiliary head (SSA_FORK tag aux), the
# RMI loss term (loss), the step's filter re-pack next to the stem (SSA_PACK_EARLY): tests with everything on, then the
# step with each off.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=${1:-r6s}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py::test_batched_filter_repack tests/test_group_gpu.py tests/test_e2e_gpu.py tests/test_parity_1024_gpu.py tests/test_graphed_step_gpu.py tests/test_amp_fp16_gpu.py tests/test_siblings_gpu.py tests/test_ddp_gpu.py tests/test_ddp_graph_gpu.py -q -x -m gpu > gpurun_out/${T}_parity.log 2>&1
echo "parity rc=$?"; tail -4 gpurun_out/${T}_parity.log
for cfg in "fuse,aux,loss 20" "fuse 0" "fuse,aux 0" "fuse,loss 0" "fuse,aux,loss 0" "fuse,aux,loss 20" "x 0" "fuse,aux,loss 20"; do
  set -- $cfg
  SSA_FORK=$1 SSA_PACK_EARLY=$2 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --eager-steps 0 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${T}_bench.json").read().strip().splitlines()[-1])
    print("fork=$1 pack_early=$2: %.3f ms  %.2f img/s" % (d["ms_per_step"], d["value"]))
except Exception as e:
    print("fork=$1 pack_early=$2 failed:", e); print(open("gpurun_out/${T}_bench.err").read()[-1500:])
PY
done
