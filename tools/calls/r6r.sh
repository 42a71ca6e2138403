#!/bin/bash
# Round 6 call R: two host-side changes of the captured step -- the upsampling half of every HRNet fuse level on a
# forked stream (SSA_FUSE_STREAM), conv bias gradients on the weight-gradient stream (SSA_BIAS_GRAD_STREAM): parity
# tests with both on, then the step with each on / off.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=${1:-r6r}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_group_gpu.py tests/test_e2e_gpu.py tests/test_parity_1024_gpu.py tests/test_graphed_step_gpu.py tests/test_amp_fp16_gpu.py -q -x -m gpu > gpurun_out/${T}_parity.log 2>&1
echo "parity rc=$?"; tail -4 gpurun_out/${T}_parity.log
for cfg in "0 0" "1 0" "0 1" "1 1" "0 0" "1 1"; do
  set -- $cfg
  SSA_FUSE_STREAM=$1 SSA_BIAS_GRAD_STREAM=$2 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --eager-steps 0 > gpurun_out/${T}_bench_f$1_b$2.json 2> gpurun_out/${T}_bench.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${T}_bench_f$1_b$2.json").read().strip().splitlines()[-1])
    print("fork=$1 bias=$2: %.3f ms  %.2f img/s" % (d["ms_per_step"], d["value"]))
except Exception as e:
    print("fork=$1 bias=$2 failed:", e); print(open("gpurun_out/${T}_bench.err").read()[-1500:])
PY
done
