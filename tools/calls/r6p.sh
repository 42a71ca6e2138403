#!/bin/bash
# Round 6 call P: the one-launch BatchNorm backward (reduce, grid-wide rendezvous, apply): its test, bnbench with the
# fused form beside the two launches, the step with it on / off.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=${1:-r6p}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_group_gpu.py tests/test_fuse_bwd_gpu.py -q -x -m gpu > gpurun_out/${T}_kernel_tests.log 2>&1
echo "kernel tests rc=$?"; tail -4 gpurun_out/${T}_kernel_tests.log
echo "== bnbench"; timeout 300 python tools/bnbench.py 30 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${T}_bnbench.txt
for f in 1 0 1 0; do
  SSA_BN_FUSED_BWD=$f timeout 400 python bench.py --dtype bf16 --steps 30 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/${T}_bench_fused$f.json 2> gpurun_out/${T}_bench.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${T}_bench_fused$f.json").read().strip().splitlines()[-1])
    f = d["roofline"]["families"]
    print("fused=$f: %.2f ms | launches %s | " % (d["ms_per_step"], d["config"]["library_launches_per_step"]) + "  ".join("%s %.3f" % (k, v["ms_per_step"]) for k, v in f.items() if k.startswith("Bn")))
except Exception as e:
    print("fused=$f failed:", e); print(open("gpurun_out/${T}_bench.err").read()[-1500:])
PY
done
timeout 300 python -m pytest tests/test_e2e_gpu.py tests/test_parity_1024_gpu.py -q -x -m gpu > gpurun_out/${T}_parity.log 2>&1
echo "parity rc=$?"; tail -3 gpurun_out/${T}_parity.log
